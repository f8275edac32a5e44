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

// ---- the same through a 16 + 16 split ---------------------------------------------------------------------------------------
// Columns 0 .. 15 as above (updates kept inside the block), then the update of columns 16 .. 31 of ALL 64 rows by the first
// sixteen in one go on the matrix cores,  A[:, 16:32] -= Y L21^T  (Y = the finished columns 0 .. 15, L21 = its rows 16 .. 31),
// then columns 16 .. 31.  That replaces 256 of the 496 readlane-pair updates (11 cycles each) by 8 v_mfma_f64_16x16x4 and two
// trips through LDS (row-per-lane -> operand layout, accumulator layout -> row-per-lane): ~1100 cycles for ~2800.
// sc: 1088 doubles of LDS private to the wave.
typedef double c32_4d __attribute__((ext_vector_type(4)));
typedef double c32_2d __attribute__((ext_vector_type(2)));
template <int C0, int C1>
static __device__ __forceinline__ void chol32_block(double (&a)[32], double& r, double& dmin) {
#pragma unroll
  for (int c = C0; c < C1; ++c) {
    a[c] *= r;
    if (c + 1 < C1) {
      const double l1 = lp_readlane(a[c], c + 1);
      a[c + 1] = fma(-a[c], l1, a[c + 1]);
      __builtin_amdgcn_sched_barrier(0);
      const double d = lp_readlane(a[c + 1], c + 1);
      dmin = fmin(dmin, d);
      r = lp_rsqrt(d);
#pragma unroll
      for (int j0 = c + 2; j0 < C1; j0 += 8) {
        double m[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) if (j0 + b < C1) m[b] = lp_readlane(a[c], j0 + b);
#pragma unroll
        for (int b = 0; b < 8; ++b) if (j0 + b < C1) a[j0 + b] = fma(-a[c], m[b], a[j0 + b]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}
static __device__ __forceinline__ void chol32_factor_split(double (&a)[32], double& dmin, double* sc) {
  const int lane = threadIdx.x & 63;
  double d = lp_readlane(a[0], 0);
  dmin = d;
  double r = lp_rsqrt(d);
  chol32_block<0, 16>(a, r, dmin);
  // Y, element (row, c) at (c / 4) * 256 + row * 4 + c % 4: the operand slices (row l % 16 of a block of 16 rows, c = 4 t + l / 16)
  // are 64 consecutive doubles
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *(c32_2d*)(sc + g * 256 + lane * 4) = (c32_2d){a[4 * g], a[4 * g + 1]};
    *(c32_2d*)(sc + g * 256 + lane * 4 + 2) = (c32_2d){a[4 * g + 2], a[4 * g + 3]};
  }
  __builtin_amdgcn_sched_barrier(0);
  // Only two of the four blocks of 16 rows need the update: rows 16 .. 31 of the tile (lanes 16 .. 31) and rows 0 .. 15 of
  // X = L^-T (lanes 32 .. 47).  Lanes 0 .. 15 hold finished rows of L (their entries right of the diagonal are not read by
  // anyone), lanes 48 .. 63 rows of X that are still zero in columns 0 .. 15.
  double y[2][4];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int t = 0; t < 4; ++t) y[b][t] = sc[t * 256 + (16 * (b + 1) + (lane & 15)) * 4 + (lane >> 4)];
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  c32_4d acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b) acc[b] = (c32_4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[b][t], y[0][t], acc[b], 0, 0, 0);
  // accumulator register q of block b: row 16 (b + 1) + lane / 16 + 4 q, column lane % 16 -> row-major, stride 17
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int q = 0; q < 4; ++q) sc[(16 * (b + 1) + (lane >> 4) + 4 * q) * 17 + (lane & 15)] = acc[b][q];
  __builtin_amdgcn_sched_barrier(0);
  double pr[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) pr[c] = sc[lane * 17 + c];
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  const bool upd = lane >= 16 && lane < 48;
#pragma unroll
  for (int c = 0; c < 16; ++c) a[16 + c] -= upd ? pr[c] : 0.0;
  d = lp_readlane(a[16], 16);
  dmin = fmin(dmin, d);
  r = lp_rsqrt(d);
  chol32_block<16, 32>(a, r, dmin);
}
