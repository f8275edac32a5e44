// Device-side helpers: deterministic wave / workgroup reductions for gfx950 (wave64).
// Cross-lane movement stays out of LDS: DPP inside a row of 16 lanes, v_permlane16_swap /
// v_permlane32_swap across rows (lane mappings verified with tools/probe/lane_probe.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

// Development / test switches (kernel A/B selection, forced code paths) are read ONLY when the process exports
// BCX_DEV=1 (tests/conftest.py and the tools/ scripts do): a production process never changes behaviour because of
// a stray variable.  Every switch in csrc/*.hip goes through this one function.
inline const char* bcx_dev_env(const char* name) {
  static const bool on = [] { const char* e = getenv("BCX_DEV"); return e && e[0] == '1'; }();
  return on ? getenv(name) : nullptr;
}

#define BCX_WAVE 64
#define BCX_SCRATCH 64   // doubles of LDS scratch for block reductions (NV <= 4, <= 16 waves)

typedef unsigned bcx_v2u __attribute__((ext_vector_type(2)));

template <int CTRL> __device__ __forceinline__ unsigned bcx_dpp_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}
template <int CTRL> __device__ __forceinline__ double bcx_dpp_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = bcx_dpp_u32<CTRL>((unsigned)b), hi = bcx_dpp_u32<CTRL>((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// value held by the lane 16 / 32 positions away (xor 16 / xor 32)
__device__ __forceinline__ unsigned bcx_xor16_u32(unsigned v) {
  const bcx_v2u r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  const unsigned x = r.x, y = r.y;   // (r[0]/r[1] through a bit_cast is miscompiled by ROCm 7.2)
  // rows after the swap: x = (r0, r0, r2, r2), y = (r1, r1, r3, r3); pick the other row's copy
  return ((__lane_id() >> 4) & 1) ? x : y;
}
__device__ __forceinline__ unsigned bcx_xor32_u32(unsigned v) {
  const bcx_v2u r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  const unsigned x = r.x, y = r.y;   // x = (lo, lo), y = (hi, hi)
  return (__lane_id() >> 5) ? x : y;
}
__device__ __forceinline__ double bcx_xor16_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = bcx_xor16_u32((unsigned)b), hi = bcx_xor16_u32((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double bcx_xor32_f64(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = bcx_xor32_u32((unsigned)b), hi = bcx_xor32_u32((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// Butterfly all-reduce: every lane ends with the same value (fixed association order).
__device__ __forceinline__ double wave_allsum(double v) {
  v += bcx_dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]   (xor 1)
  v += bcx_dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]   (xor 2)
  v += bcx_dpp_f64<0x141>(v);   // row_half_mirror       (other quad of the 8)
  v += bcx_dpp_f64<0x140>(v);   // row_mirror            (other half of the 16)
  v += bcx_xor16_f64(v);
  v += bcx_xor32_f64(v);
  return v;
}
__device__ __forceinline__ double wave_allmax(double v) {
  v = fmax(v, bcx_dpp_f64<0xB1>(v));
  v = fmax(v, bcx_dpp_f64<0x4E>(v));
  v = fmax(v, bcx_dpp_f64<0x141>(v));
  v = fmax(v, bcx_dpp_f64<0x140>(v));
  v = fmax(v, bcx_xor16_f64(v));
  v = fmax(v, bcx_xor32_f64(v));
  return v;
}

// Sum NV doubles per thread over the whole workgroup; every thread receives the totals.
// Fixed order (butterfly inside a wave, then waves 0..nw-1), so results are reproducible
// and identical on every shard.  scratch needs NV * (blockDim.x/64) doubles.
template <int NV>
__device__ __forceinline__ void block_allsum(double (&v)[NV], double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = wave_allsum(v[i]);
  if (nw == 1) return;
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) scratch[wave * NV + i] = v[i];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double t = scratch[i];
    for (int w = 1; w < nw; ++w) t += scratch[w * NV + i];
    v[i] = t;
  }
  __syncthreads();
}

__device__ __forceinline__ double block_allmax(double v, double* scratch) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v = wave_allmax(v);
  if (nw == 1) return v;
  if (lane == 0) scratch[wave] = v;
  __syncthreads();
  double t = scratch[0];
  for (int w = 1; w < nw; ++w) t = fmax(t, scratch[w]);
  __syncthreads();
  return t;
}
