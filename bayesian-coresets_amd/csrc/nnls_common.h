// Shared by nnls.hip (OMP step, single-workgroup solves) and nnls_grid.hip (multi-workgroup optimize()).
#pragma once
#include "bcx_internal.h"
#include "dev_util.h"
#include "apply_common.h"

#define NN_THREADS 1024
#define FLAG_INS 1
#define FLAG_REJ 2
#define FLAG_RM 4

struct NnlsArgs {
  ApplyArgs a;
  double* gram;
  double* hinv;
  int64_t ldg;
  double* cvec;
  int32_t* plist;
  int32_t* ppos;
  double* x;       // per slot
  double* z;       // per position
  double* t0;      // per position scratch
  double* t1;
  double* t2;
  double* t3;      // per slot: Gram-row candidate
  int32_t* flag;   // per slot
  double* wbak;    // per slot
};

// ---- small workgroup-wide helpers ---------------------------------------------------------------
// arg-max of (val, idx): larger val wins, ties -> smaller idx.  Entries with idx < 0 are ignored.
struct ArgBest { double v; int i; };
static __device__ ArgBest block_argbest(double v, int i, double* scratch) {
  __shared__ double sv[16];
  __shared__ int si[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (i < 0) v = -INFINITY;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(v, off, BCX_WAVE);
    const int oi = __shfl_xor(i, off, BCX_WAVE);
    if (oi >= 0 && (i < 0 || ov > v || (ov == v && oi < i))) { v = ov; i = oi; }
  }
  if (lane == 0) { sv[wave] = v; si[wave] = i; }
  __syncthreads();
  ArgBest r; r.v = sv[0]; r.i = si[0];
  for (int w = 1; w < nw; ++w)
    if (si[w] >= 0 && (r.i < 0 || sv[w] > r.v || (sv[w] == r.v && si[w] < r.i))) { r.v = sv[w]; r.i = si[w]; }
  __syncthreads();
  return r;
}

// ---- grid barrier: an arrival counter in device memory ---------------------------------------------------
// Barrier `index` (1, 2, ...) of a launch is "counter >= base + index * gridDim.x"; every workgroup arrives
// exactly once per barrier (or the same number of times by grid_arrive when it skips some).  All workgroups
// of the launch must be co-resident (grids of <= 64 workgroups on 256 CUs, nothing else on the stream).
struct GridSync {
  unsigned long long* counter;
  unsigned long long base;
  long long timeout_ticks;
};

static __device__ __forceinline__ void grid_arrive(const GridSync& g, int times) {
  __threadfence();
  __hip_atomic_fetch_add(g.counter, (unsigned long long)times, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// all threads; false after a timeout (the build is then stopped instead of hanging the GPU)
static __device__ bool grid_barrier(const GridSync& g, int index, int* s_flag) {
  __syncthreads();
  if (threadIdx.x == 0) {
    grid_arrive(g, 1);
    const unsigned long long target = g.base + (unsigned long long)index * gridDim.x;
    const long long t0 = wall_clock64();
    int ok = 1;
    while (__hip_atomic_load(g.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t0 > g.timeout_ticks) { ok = 0; break; }
    }
    __threadfence();
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}


void fill_nnls_args(bcx_solver* s, NnlsArgs& n, const double* recs);   // nnls.hip
int bcx_launch_optimize_grid(bcx_solver* s, double tol, int k);         // nnls_grid.hip
