// laplace.hip -- the sampler of the reference's logistic / Poisson regression experiment from weights that live on the device
// (examples/logistic_poisson_regression/main.py:15-41 get_laplace + :155-162 sampler_w): the Laplace approximation of the
// WEIGHTED posterior of k points under a standard-normal prior,
//     mu  = argmax_theta  sum_j w_j log p(z_j | theta) - |theta|^2 / 2,        Sigma = (I - sum_j w_j hess_j)^-1  at mu,
// and S draws  theta = mu + R W  with  W = L^-1,  L L^T = Sigma^-1  (Sigma = W^T W).  SparseVI asks for it once per ADAM step
// (sparsevi.py:25): on the host that is a SciPy minimisation + a Cholesky + an upload per step; here it is ONE launch of one
// workgroup that reads the k weights where the ADAM kernel left them, so the whole loop can be enqueued (csrc/svi.hip).
//
// The reference finds the mode with SciPy's BFGS (gtol 1e-5); this is damped Newton -- the same iteration as
// examples/common/model_lr.py / model_poiss.py laplace_fit of this package, which tests/ pin to the reference's outputs (F15):
// gradient and Hessian sums over the points (k x D^2 multiply-adds spread over the workgroup, the points resident in LDS),
// the D x D Newton system by the in-register Cholesky of csrc/chol32.h (D <= 32: one wave, the inverse of the factor comes with
// it), step halving until the objective does not decrease, stop at |step| < tol.  The Poisson log-likelihood is not concave
// in the linear predictor everywhere: a Newton matrix that is not positive definite gets a multiple of the identity added
// (ten-fold until the factorisation succeeds).  The mode of the previous call is kept on the device and may seed the next
// one (warm: the weights of consecutive ADAM steps differ little: 2-3 Newton steps instead of ~10).
#include <atomic>
#include <string>
#include "bcx_internal.h"
#include "dev_util.h"
#include "chol32.h"

#define LAP_DMAX 32
#define LAP_THREADS 256
enum { LAP_LOGISTIC = 0, LAP_POISSON = 1 };

struct LapArgs {
  const double* w;      // k weights (negative ones count as zero)
  const double* pts;    // k x ldp: logistic rows z = y x (D values); Poisson rows [x (D values), y]
  double* mu;           // D: in (warm != 0) the start of the iteration, out the mode
  const double* R;      // S x ld standard normal numbers
  const double* Rbar;   // ld: their column means
  double* theta;        // S x ld draws
  double* tbar;         // D: mean of the draws
  int* status;          // [0] 0 ok / 1 iteration limit / 2 no positive definite Newton matrix; [1] Newton steps taken
  double tol;
  int family, k, D, ldp, S, ld, max_iter, warm;
};

// d/ds and d^2/ds^2 of the log-likelihood in the linear predictor s, and the log-likelihood itself (constants in theta dropped)
static __device__ __forceinline__ void lap_point(int family, double s, double y, double& ll, double& g, double& h) {
  if (family == LAP_LOGISTIC) {
    // log p = -log(1 + exp(-s)), linear tail beyond -s >= 100 (model_lr.py:29-31)
    const double arg = -s;
    if (arg < 100.0) {
      const double e = exp(arg);
      ll = -log1p(e);
      g = e / (1.0 + e);
      h = -e / ((1.0 + e) * (1.0 + e));
    } else { ll = -arg; g = 1.0; h = 0.0; }
  } else {
    // rate = log(1 + e^s); log p = y log rate - rate (- log y!); log rate = s where rate = e^s to every bit (model_poiss.py:25-38)
    const double e = exp(-fabs(s));
    const double rate = fmax(s, 0.0) + log1p(e);
    const double lr = s > -100.0 ? log(rate > 0.0 ? rate : 1.0) : s;
    ll = y * lr - rate;
    const double sig = (s >= 0.0 ? 1.0 : e) / (1.0 + e);           // rate'
    const double dsig = e / ((1.0 + e) * (1.0 + e));                // rate''
    const double safe = rate > 0.0 ? rate : 1.0;
    const double r1 = rate > 0.0 ? sig / safe : 1.0;                // rate' / rate
    const double r2 = rate > 0.0 ? (dsig * safe - sig * sig) / (safe * safe) : 0.0;
    g = y * r1 - sig;
    h = y * r2 - dsig;
  }
}

__global__ __launch_bounds__(LAP_THREADS) void laplace_sampler_kernel(LapArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lap_dyn[];
  __shared__ double s_dg[32 * 33], s_W[32 * 33];
  __shared__ double s_th[32], s_cand[32], s_grad[32], s_step[32], s_y[32];
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ int s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = a.k, D = a.D, Dp = D + 1;           // (rows of D + 1 doubles in LDS: consecutive points on different banks)
  double* sX = lap_dyn;                              // k x Dp features
  double* sw = sX + (size_t)k * Dp;                  // k weights
  double* sy = sw + k;                               // k responses (Poisson)
  double* sg = sy + k;                               // k: w_j g_j
  double* sh = sg + k;                               // k: w_j h_j
  for (int e = tid; e < k * D; e += LAP_THREADS) { const int j = e / D, c = e - j * D; sX[j * Dp + c] = a.pts[(size_t)j * a.ldp + c]; }
  for (int j = tid; j < k; j += LAP_THREADS) {
    sw[j] = fmax(a.w[j], 0.0);
    sy[j] = a.family == LAP_POISSON ? a.pts[(size_t)j * a.ldp + D] : 0.0;
  }
  if (tid < 32) s_th[tid] = (tid < D && a.warm) ? a.mu[tid] : 0.0;
  __syncthreads();

  // objective at `at` (LDS, D values); with_derivs: also w_j g_j, w_j h_j per point
  auto objective = [&](const double* at, bool with_derivs) -> double {
    double part[1] = {0.0};
    for (int j = tid; j < k; j += LAP_THREADS) {
      double s = 0.0;
      for (int c = 0; c < D; ++c) s = fma(sX[j * Dp + c], at[c], s);
      double ll, g, h;
      lap_point(a.family, s, sy[j], ll, g, h);
      part[0] += sw[j] * ll;
      if (with_derivs) { sg[j] = sw[j] * g; sh[j] = sw[j] * h; }
    }
    block_allsum<1>(part, scratch);
    double q = 0.0;
    for (int c = 0; c < D; ++c) q = fma(at[c], at[c], q);
    return part[0] - 0.5 * q;
  };
  // gradient and Newton matrix I - sum_j w_j h_j x_j x_j^T from sg / sh, the matrix padded to 32 x 32 with the identity
  auto assemble = [&]() {
    for (int e = tid; e < 32 * 32 + 32; e += LAP_THREADS) {
      if (e < 32 * 32) {
        const int r = e >> 5, c = e & 31;
        double v = r == c ? 1.0 : 0.0;
        if (r < D && c < D) {
          double t = 0.0;
          for (int j = 0; j < k; ++j) t = fma(sh[j] * sX[j * Dp + r], sX[j * Dp + c], t);
          v -= t;
        }
        s_dg[r * 33 + c] = v;
      } else {
        const int c = e - 32 * 32;
        double t = 0.0;
        if (c < D) {
          for (int j = 0; j < k; ++j) t = fma(sg[j], sX[j * Dp + c], t);
          t -= s_th[c];
        }
        s_grad[c] = t;
      }
    }
    __syncthreads();
  };
  // W = L^-1 of the Newton matrix in s_dg (row-major in s_W); a matrix that is not positive definite gets lambda I added
  auto factor = [&]() -> bool {
    double lambda = 0.0;
    for (int attempt = 0; attempt < 24; ++attempt) {
      if (tid == 0) s_bad = 0;
      __syncthreads();
      if (wave == 0) {
        double r[32];
        const int row = lane & 31;
        const bool top = lane < 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) r[c] = top ? (c <= row ? s_dg[row * 33 + c] + (c == row ? lambda : 0.0) : 0.0) : (c == row ? 1.0 : 0.0);
        double dmin;
        chol32_factor(r, dmin);
        if (!(dmin > 0.0)) { if (lane == 0) s_bad = 1; }
        else if (!top) {
#pragma unroll
          for (int c = 0; c < 32; ++c) s_W[c * 33 + row] = r[c];     // lane 32 + kk holds X[kk][c] = W[c][kk]
        }
      }
      __syncthreads();
      if (!s_bad) return true;
      double dmax = 1.0;
      for (int c = 0; c < D; ++c) dmax = fmax(dmax, fabs(s_dg[c * 33 + c]));
      lambda = lambda == 0.0 ? 1e-8 * dmax : lambda * 10.0;
      __syncthreads();
    }
    return false;
  };

  int status = 1, steps = 0;
  double f = objective(s_th, true);
  for (int it = 0; it < a.max_iter; ++it) {
    assemble();
    if (!factor()) { status = 2; break; }
    // step = W^T (W grad)
    if (tid < 32) {
      double t = 0.0;
      for (int c = 0; c <= tid; ++c) t = fma(s_W[tid * 33 + c], s_grad[c], t);
      s_y[tid] = t;
    }
    __syncthreads();
    if (tid < 32) {
      double t = 0.0;
      for (int c = tid; c < 32; ++c) t = fma(s_W[c * 33 + tid], s_y[c], t);
      s_step[tid] = tid < D ? t : 0.0;
    }
    __syncthreads();
    double smax = 0.0;
    for (int c = 0; c < D; ++c) smax = fmax(smax, fabs(s_step[c]));
    // damped: the objective must not decrease
    double t = 1.0, fc;
    for (;;) {
      if (tid < 32) s_cand[tid] = s_th[tid] + t * s_step[tid];
      __syncthreads();
      fc = objective(s_cand, true);
      if (fc >= f || t < 1e-10) break;
      t *= 0.5;
      __syncthreads();
    }
    __syncthreads();
    if (tid < 32) s_th[tid] = s_cand[tid];
    f = fc;
    ++steps;
    __syncthreads();
    if (smax * t < a.tol) { status = 0; break; }
  }
  // the covariance factor at the mode (sg / sh are the mode's: every accepted candidate was evaluated with derivatives)
  bool have_W = false;
  if (status != 2) { assemble(); have_W = factor(); if (!have_W) status = 2; }
  if (tid < D) a.mu[tid] = s_th[tid];
  if (tid == 0) { a.status[0] = status; a.status[1] = steps; }
  if (!have_W) {                                     // (no factor: the draws are the mode -- the caller raises on the status)
    for (int e = tid; e < 32 * 33; e += LAP_THREADS) s_W[e] = 0.0;
    __syncthreads();
  }
  // theta = mu + [R; Rbar] W   (W lower triangular: column a takes rows i >= a)
  for (int e = tid; e < (a.S + 1) * a.ld; e += LAP_THREADS) {
    const int s = e / a.ld, c = e - s * a.ld;
    double v = 0.0;
    if (c < D) {
      const double* rr = s < a.S ? a.R + (size_t)s * a.ld : a.Rbar;
      v = s_th[c];
      for (int i = c; i < D; ++i) v = fma(rr[i], s_W[i * 33 + c], v);
    }
    if (s < a.S) a.theta[(size_t)s * a.ld + c] = v;
    else if (c < D) a.tbar[c] = v;
  }
}

void bcx_project_set_error(const std::string& msg);   // proj.hip
extern "C" int64_t bcx_laplace_sampler_lds_bytes(int32_t k, int32_t D) {
  if (k < 0 || D < 1 || D > LAP_DMAX) return -1;
  return ((int64_t)k * (D + 1) + 4 * (int64_t)k) * (int64_t)sizeof(double);
}
// 1 when one workgroup can hold the k points (D <= 32 parameters, the points and four doubles each in 96 KiB of LDS)
extern "C" int bcx_laplace_sampler_ok(int32_t k, int32_t D) {
  const int64_t b = bcx_laplace_sampler_lds_bytes(k, D);
  return b >= 0 && b <= 96 * 1024;
}
extern "C" int bcx_laplace_sampler(void* stream, int32_t family, int32_t k, int32_t D, const void* w_dev, const void* pts_dev, int64_t ldp,
                                   void* mu_dev, int32_t warm, double tol, int32_t max_iter, const void* R_dev, const void* Rbar_dev,
                                   int32_t S, int32_t ld, void* theta_dev, void* tbar_dev, void* status_dev) {
  if ((family != LAP_LOGISTIC && family != LAP_POISSON) || !bcx_laplace_sampler_ok(k, D) || S < 1 || ld < D || max_iter < 1 || !(tol > 0.0) ||
      !mu_dev || !R_dev || !Rbar_dev || !theta_dev || !tbar_dev || !status_dev ||
      (k > 0 && (!w_dev || !pts_dev || ldp < D + (family == LAP_POISSON ? 1 : 0)))) {
    bcx_project_set_error("bcx_laplace_sampler: bad arguments (family 0 logistic / 1 Poisson, D <= 32 parameters, the points within "
                          "bcx_laplace_sampler_ok)");
    return BCX_ERR_ARG;
  }
  LapArgs a;
  a.w = (const double*)w_dev; a.pts = (const double*)pts_dev; a.mu = (double*)mu_dev; a.R = (const double*)R_dev;
  a.Rbar = (const double*)Rbar_dev; a.theta = (double*)theta_dev; a.tbar = (double*)tbar_dev; a.status = (int*)status_dev;
  a.tol = tol; a.family = family; a.k = k; a.D = D; a.ldp = (int)ldp; a.S = S; a.ld = ld; a.max_iter = max_iter; a.warm = warm;
  const size_t lds = (size_t)bcx_laplace_sampler_lds_bytes(k, D);
  if (lds > 32 * 1024) {
    static std::atomic<size_t> lds_max[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > lds_max[dev].load(std::memory_order_acquire)) {
      if (hipFuncSetAttribute((const void*)laplace_sampler_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        bcx_project_set_error("bcx_laplace_sampler: hipFuncSetAttribute failed");
        return BCX_ERR_HIP;
      }
      lds_max[dev].store(lds, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(laplace_sampler_kernel, dim3(1), dim3(LAP_THREADS), lds, (hipStream_t)stream, a);
  if (hipGetLastError() != hipSuccess) { bcx_project_set_error("bcx_laplace_sampler: launch failed"); return BCX_ERR_HIP; }
  return BCX_OK;
}
