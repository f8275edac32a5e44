// Device-side helpers: deterministic wave / workgroup reductions for gfx950 (wave64).
#pragma once
#include <hip/hip_runtime.h>

#define BCX_WAVE 64
#define BCX_SCRATCH 64   // doubles of LDS scratch for block reductions (NV <= 4, <= 16 waves)

__device__ __forceinline__ double wave_allsum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, BCX_WAVE);
  return v;
}
__device__ __forceinline__ double wave_allmax(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off, BCX_WAVE));
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
