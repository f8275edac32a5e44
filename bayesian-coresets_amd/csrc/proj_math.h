// proj_math.h -- the transcendental pieces of the likelihood epilogues of proj.hip, table driven.
//
// On this chip an fp64 VALU instruction and an fp64 MFMA use the same datapath: two waves on one SIMD, one issuing
// v_mfma_f64_4x4x4_4b_f64 and the other v_fma_f64, take the SUM of their separate times (tools/probe/mfma_valu_overlap.hip:
// 11.9 + 11.5 -> 23.4 ms), and 32-bit moves / integer instructions beside them cost nothing measurable.  An epilogue can
// therefore not hide under the MFMAs of the other resident workgroup; what it costs is its number of fp64 instructions.
// The series forms of round 2/3 (exp: Cody-Waite + degree 13, log1p / log: 2 atanh with a division and 11-16 terms) spend
// ~52 fp64 instructions per logistic value and ~95 per Poisson value.  With three small tables in LDS the same functions
// take 11 + 10 and 12:
//   exp(x), x <= 0 : x = n ln2/64 + r, n = 64 k + j  ->  2^k * T[j] * (1 + r + ... + r^5/120),   |r| <= ln2/128
//   log1p(u), 0<=u<=1 : c = i/64 nearest to u  ->  L1[i] + log1p(w), w = (u - c) / (1 + c), series to w^7/7, |w| <= 1/128
//   log(x), x > 0 : x = 2^e m, c = 1 + i/128 nearest to m  ->  e ln2 + L2[i] + log1p(w), w = m R[i] - 1 (one fma; L2[i] is
//                   -log of the ROUNDED R[i], so the identity is exact), |w| <= 1/256
// n, i and e come out of the operands' bit patterns (integer instructions), the table entries by ds_read gathers.
// Accuracy against long double (tools/series_check.cpp, 10^7 arguments each): exp <= 1.5 ulp, log1p <= 1.5 ulp, log <= 1 ulp
// away from x = 1 and an absolute 6e-17 next to it (the likelihood uses y log(rate) beside the rate itself).
//
// The header compiles for the device (proj.hip) and for the host (the accuracy harness and the table builder).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define PJM_HD __device__ __forceinline__
#else
#define PJM_HD inline
#endif

// table layout, in doubles
#define PJT_EXP 0                      // 64:  2^(j/64)
#define PJT_L1P 64                     // 65 x {c, R = 1/(1+c) rounded, L = log1p(c), 0}
#define PJT_LOG (64 + 65 * 4)          // 129 x {R = 1/c rounded, L = -log(R)},  c = 1 + i/128
#define PJT_LFACT (PJT_LOG + 129 * 2)  // 256: log(y!) = gammaln(y + 1) for the integer responses y = 0 .. 255 of a Poisson model
#define PJT_NFACT 256
#define PJT_DOUBLES (PJT_LFACT + PJT_NFACT)
#define PJT_DOUBLES_LOGISTIC PJT_LFACT   // (the logistic family has no gammaln; its column sums take ONE log per column and tile: proj.hip)
#define PJT_BYTES(n) (((n) * 8 + 15) / 16 * 16)

PJM_HD int pjm_hi(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __double2hiint(x);
#else
  int64_t b; std::memcpy(&b, &x, 8); return (int)(b >> 32);
#endif
}
PJM_HD int pjm_lo(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __double2loint(x);
#else
  int64_t b; std::memcpy(&b, &x, 8); return (int)(uint32_t)b;
#endif
}
PJM_HD double pjm_make(int hi, int lo) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __hiloint2double(hi, lo);
#else
  const int64_t b = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
  double x; std::memcpy(&x, &b, 8); return x;
#endif
}

// a * b + C and a * K + b with the constant in a scalar register pair (device): hipcc materialises an fp64 literal with two
// v_mov_b32 into a VGPR pair and tends to keep the pairs of a whole polynomial live -- registers the 128-column tile
// does not have; as an SGPR operand of v_fma_f64 the constant costs two s_mov_b32 on the scalar unit.
PJM_HD double pjm_fma_c(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c));
  return r;
#else
  return fma(a, b, c);
#endif
}
PJM_HD double pjm_fma_k(double a, double k, double b) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(b));
  return r;
#else
  return fma(a, k, b);
#endif
}

// max(t, 0) by the sign bit (NaNs with a clear sign bit pass, the others meet a NaN term anyway)
PJM_HD double pjm_relu(double t) { return pjm_hi(t) < 0 ? 0.0 : t; }

// exp(x) for x <= 0.  11 fp64 instructions.
template <class TP> PJM_HD double pjm_exp_nonpos(double x, TP tab) {
  const double magic = 6755399441055744.0;                  // 1.5 * 2^52: the sum's low mantissa bits hold rint(x * 64/ln2)
  // clamp at -800 by the bit pattern (for x <= 0 the high word grows with |x|) so that a NaN passes and comes out as a
  // NaN, as it does in the reference: fmax(x, -800) returns -800 for it and the likelihood of a point with a NaN feature
  // would be a finite number.  (Measured against the fmax forms: 1-2 % of the transcendental kernels.)
  const unsigned hx = (unsigned)pjm_hi(x);
  x = (hx > 0xC0890000u && hx <= 0xFFF00000u) ? -800.0 : x;
  const double t = pjm_fma_k(x, 92.332482616893656877, magic);    // 64 / ln 2
  const int n = pjm_lo(t);                                  // <= 0, two's complement
  const double kf = t - magic;
  double r = pjm_fma_k(kf, -6.93147180369123816490e-01 / 64.0, x);    // ln2/64 in two pieces (the high one has 32 significant bits)
  r = pjm_fma_k(kf, -1.90821492927058770002e-10 / 64.0, r);
  double q = pjm_fma_k(r, 1.0 / 120.0, 1.0 / 24.0);
  q = pjm_fma_c(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  q = fma(q, r, 1.0);
  q = fma(q, r, 1.0);
  return ldexp(tab[PJT_EXP + (n & 63)] * q, n >> 6);
}

// log1p(u) for 0 <= u <= 1.  10 fp64 instructions.
template <class TP> PJM_HD double pjm_log1p01(double u, TP tab) {
  const double v = 1.0 + u;                                 // only its leading bits are used: the index
  int i = (pjm_hi(v) - 0x3FF00000 + (1 << 13)) >> 14;       // rint((v - 1) * 64): 0 .. 64
  i = i < 0 ? 0 : (i > 64 ? 64 : i);                        // (a NaN stays a NaN below; it must not pick the address)
  const double c = tab[PJT_L1P + 4 * i], R = tab[PJT_L1P + 4 * i + 1], L = tab[PJT_L1P + 4 * i + 2];
  const double w = (u - c) * R;                             // u - c is exact
  double p = pjm_fma_k(w, 1.0 / 7.0, -1.0 / 6.0);
  p = pjm_fma_c(p, w, 0.2);
  p = pjm_fma_c(p, w, -0.25);
  p = pjm_fma_c(p, w, 1.0 / 3.0);
  p = fma(p, w, -0.5);
  p = fma(p, w, 1.0);
  return fma(w, p, L);
}

// log(x) for a positive normal x.  12 fp64 instructions.
template <class TP> PJM_HD double pjm_log_pos(double x, TP tab) {
  const int hi = pjm_hi(x);
  const int e = (hi >> 20) - 1023, mh = hi & 0xFFFFF;
  const int i = (mh + (1 << 12)) >> 13;                     // rint((m - 1) * 128): 0 .. 128
  const double m = pjm_make(mh | 0x3FF00000, pjm_lo(x));    // [1, 2)
  const double R = tab[PJT_LOG + 2 * i], L = tab[PJT_LOG + 2 * i + 1];
  const double w = fma(m, R, -1.0);
  double p = pjm_fma_k(w, 1.0 / 7.0, -1.0 / 6.0);
  p = pjm_fma_c(p, w, 0.2);
  p = pjm_fma_c(p, w, -0.25);
  p = pjm_fma_c(p, w, 1.0 / 3.0);
  p = fma(p, w, -0.5);
  p = fma(p, w, 1.0);
  const double ed = (double)e;
  return fma(ed, 6.93147180369123816490e-01, L) + fma(w, p, ed * 1.90821492927058770002e-10);
}

// The tables, rounded from long double (host).
inline void pjm_fill_tables(double* tab) {
  for (int j = 0; j < 64; ++j) tab[PJT_EXP + j] = (double)exp2l((long double)j / 64.0L);
  for (int i = 0; i <= 64; ++i) {
    const long double c = (long double)i / 64.0L;
    tab[PJT_L1P + 4 * i] = (double)c;
    tab[PJT_L1P + 4 * i + 1] = (double)(1.0L / (1.0L + c));
    tab[PJT_L1P + 4 * i + 2] = (double)log1pl(c);
    tab[PJT_L1P + 4 * i + 3] = 0.0;
  }
  for (int i = 0; i <= 128; ++i) {
    const double R = (double)(1.0L / (1.0L + (long double)i / 128.0L));
    tab[PJT_LOG + 2 * i] = R;
    tab[PJT_LOG + 2 * i + 1] = (double)(-logl((long double)R));
  }
  for (int y = 0; y < PJT_NFACT; ++y) tab[PJT_LFACT + y] = (double)lgammal((long double)y + 1.0L);
  // (i = 0: R = 1, L = 0 exactly, so log(x) keeps full relative accuracy just above 1; just below 1 -- e = -1, i = 128,
  // L = double(ln 2) -- the two pieces of e ln2 cancel L up to its rounding: an absolute 2e-17)
}
