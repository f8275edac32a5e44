// dev harness: accuracy of the table-driven exp / log1p / log of bayesian-coresets_amd/csrc/proj_math.h against long double.
//   g++ -O2 -std=c++17 -o /tmp/series_check tools/series_check.cpp && /tmp/series_check
// (host build of the same header the device kernel compiles; fma() is the correctly rounded libm / hardware one)
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../bayesian-coresets_amd/csrc/proj_math.h"

static double ulp_of(long double v) {
  int e; frexpl(v < 0 ? -v : v, &e);
  return ldexp(1.0, e - 53);
}

int main(int argc, char** argv) {
  const long n = argc > 1 ? atol(argv[1]) : 10000000;
  static double tab[PJT_DOUBLES];
  pjm_fill_tables(tab);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  double worst_exp = 0, worst_l1p = 0, worst_log = 0, worst_log_abs = 0, worst_sp = 0;
  double at_exp = 0, at_l1p = 0, at_log = 0;
  for (long k = 0; k < n; ++k) {
    // exp on [-800, 0]: a third uniform, a third near zero, a third log-uniform
    double x;
    const int c = k % 3;
    if (c == 0) x = -760.0 * U(g);
    else if (c == 1) x = -40.0 * U(g);
    else x = -exp(-40.0 * U(g));
    {
      const double got = pjm_exp_nonpos(x, tab);
      const long double want = expl((long double)x);
      if (want > 1e-300L) {
        const double err = fabs((double)((long double)got - want)) / ulp_of(want);
        if (err > worst_exp) { worst_exp = err; at_exp = x; }
      }
    }
    // log1p on (0, 1]: uniform and log-uniform
    double u = (k & 1) ? U(g) : exp(-700.0 * U(g));
    if (k % 1000 == 0) u = (double)(k % 65) / 64.0;                     // the table's own nodes
    if (k % 1000 == 1) u = nextafter((double)(k % 64 + 0.5) / 64.0, k & 2 ? 1.0 : 0.0);   // the interval edges
    {
      const double got = pjm_log1p01(u, tab);
      const long double want = log1pl((long double)u);
      const double err = want == 0 ? 0 : fabs((double)((long double)got - want)) / ulp_of(want);
      if (err > worst_l1p) { worst_l1p = err; at_l1p = u; }
    }
    // softplus(t) = max(t, 0) + log1p(exp(-|t|)) on [-120, 120]
    {
      const double t = 240.0 * U(g) - 120.0;
      const double got = pjm_relu(t) + pjm_log1p01(pjm_exp_nonpos(-fabs(t), tab), tab);
      const long double want = (t > 0 ? (long double)t : 0.0L) + log1pl(expl(-(long double)fabs(t)));
      const double err = fabs((double)((long double)got - want)) / ulp_of(want);
      if (err > worst_sp) worst_sp = err;
    }
    // log on positive normals: rates in [3.7e-44, 1e6], and a band around 1
    double y = (k % 4 == 0) ? 1.0 + (U(g) - 0.5) * 1e-3 : exp(-100.0 + 114.0 * U(g));
    if (k % 1000 == 2) y = nextafter(1.0, k & 8 ? 2.0 : 0.0);
    {
      const double got = pjm_log_pos(y, tab);
      const long double want = logl((long double)y);
      const double aerr = fabs((double)((long double)got - want));
      if (fabs(y - 1.0) < 0.3) { if (aerr > worst_log_abs) worst_log_abs = aerr; }
      else {
        const double err = aerr / ulp_of(want);
        if (err > worst_log) { worst_log = err; at_log = y; }
      }
    }
  }
  // NaN in, NaN out; -inf and arguments below the clamp give 0
  const double qnan = std::nan("");
  const bool special = std::isnan(pjm_exp_nonpos(qnan, tab)) && std::isnan(pjm_exp_nonpos(-qnan, tab)) && std::isnan(pjm_log1p01(qnan, tab)) &&
                       std::isnan(pjm_relu(qnan) + pjm_log1p01(pjm_exp_nonpos(-fabs(qnan), tab), tab)) &&
                       pjm_exp_nonpos(-INFINITY, tab) == 0.0 && pjm_exp_nonpos(-1e300, tab) == 0.0 && pjm_exp_nonpos(-0.0, tab) == 1.0 &&
                       pjm_relu(-0.0) == 0.0 && pjm_relu(3.5) == 3.5 && pjm_relu(-2.0) == 0.0;
  printf("special values: %s\n", special ? "ok" : "WRONG");
  printf("%ld arguments each\n", n);
  printf("exp(x), x in [-800, 0]            : worst %.2f ulp (x = %.17g)\n", worst_exp, at_exp);
  printf("log1p(u), u in (0, 1]             : worst %.2f ulp (u = %.17g)\n", worst_l1p, at_l1p);
  printf("softplus(t), t in [-120, 120]     : worst %.2f ulp\n", worst_sp);
  printf("log(x), |x - 1| >= 0.3            : worst %.2f ulp (x = %.17g)\n", worst_log, at_log);
  printf("log(x), |x - 1| <  0.3            : worst absolute error %.3g\n", worst_log_abs);
  // bounds the parity tests rely on (tests/test_series_math.py)
  const bool ok = special && worst_exp <= 3.0 && worst_l1p <= 4.0 && worst_sp <= 5.0 && worst_log <= 2.0 && worst_log_abs <= 1.2e-16;
  printf("%s\n", ok ? "WITHIN BOUNDS" : "OUT OF BOUNDS");
  return ok ? 0 : 1;
}
