// chol32.h -- Cholesky of a 32 x 32 tile and the inverse of its factor in ONE wave, in registers (csrc/lrpost.hip: the diagonal
// tiles of the cooperative D x D factorisation; csrc/laplace.hip: the Newton systems of the weighted Laplace posterior).
#pragma once
#include <hip/hip_runtime.h>

static __device__ __forceinline__ double lp_readlane(double v, int l) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
static __device__ __forceinline__ double lp_rsqrt(double d) {
  double r = __builtin_amdgcn_rsq(d);
  double e = fma(-d * r, r, 1.0);                   // two Newton steps for 1 / sqrt(d)
  r = fma(0.5 * r, e, r);
  e = fma(-d * r, r, 1.0);
  return fma(0.5 * r, e, r);
}

// Lane l < 32 holds row l of the symmetric positive definite tile in a[0 .. 31] (entries right of the diagonal: zero), lane
// 32 + l row l of the identity; the column operations that produce L in the first turn the second into
// X = L^-T.  Per column c the multipliers l_jc (j > c) reach all lanes by v_readlane; they are fetched EIGHT at a time into
// distinct scalar registers before their multiply-adds, and the 30 - c updates of columns j >= c + 2 are issued beside the next
// column's dependent chain (update of column c + 1 -> pivot -> rsqrt with two Newton steps -> scale).  Measured on this chip
// (tools/probe/f64_chain_probe.hip, cycles per wave instruction): independent v_fma_f64 5, dependent 6.5, v_rsq_f64 18, a
// v_readlane pair + fma 11 when the scalar pair is not reused back to back (29 when it is: the compiler's own schedule of the
// plain loop, 6 us per tile), a readlane -> fma hop 25, a BROADCAST ds_read_b128 36 per wave (so multipliers through LDS are
// slower than through v_readlane: 8 us per tile), a dependent v_mfma_f64_16x16x4 64.  The loop body is software-pipelined by
// hand and pinned with scheduling barriers.  On return a[] holds L (lanes 0 .. 31) and X = L^-T (lanes 32 .. 63);
// dmin: the smallest pivot met (not positive: the tile was not positive definite, a[] holds NaNs).
static __device__ __forceinline__ void chol32_factor(double (&a)[32], double& dmin) {
  double d = lp_readlane(a[0], 0);
  dmin = d;
  double r = lp_rsqrt(d);
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    a[c] *= r;
    if (c + 1 < 32) {
      const double l1 = lp_readlane(a[c], c + 1);
      a[c + 1] = fma(-a[c], l1, a[c + 1]);
      __builtin_amdgcn_sched_barrier(0);
      d = lp_readlane(a[c + 1], c + 1);
      dmin = fmin(dmin, d);                         // (a non-positive pivot leaves NaNs behind; reported once, by the caller)
      r = lp_rsqrt(d);
#pragma unroll
      for (int j0 = c + 2; j0 < 32; j0 += 8) {
        double m[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) if (j0 + b < 32) m[b] = lp_readlane(a[c], j0 + b);
#pragma unroll
        for (int b = 0; b < 8; ++b) if (j0 + b < 32) a[j0 + b] = fma(-a[c], m[b], a[j0 + b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
