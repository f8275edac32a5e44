// Shared by nnls.hip (OMP step, single-workgroup solves) and nnls_grid.hip (multi-workgroup optimize()).
#pragma once
#include "bcx_internal.h"
#include "dev_util.h"
#include "apply_common.h"

#define NN_THREADS 1024
// words behind bcx_solver::grid_counter: [0] arrivals, [1] barrier base of the next OMP step, [16] (its own 128-byte line) the
// barrier index word of the split form (GridSync::gen)
#define BCX_GRID_WORDS 32
#define FLAG_INS 1
#define FLAG_REJ 2
#define FLAG_RM 4

struct NnlsArgs {
  ApplyArgs a;
  double* gram;
  double* hinv;
  double* hlo;     // low words of hinv (double-double), same layout; only omp_lh.hip reads / writes it
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
  double* xr;      // 2 * ldg doubles: row-phase exchange of the OMP step (row_j . x_f, row_j . residual), by slot
};

// ---- small workgroup-wide helpers ---------------------------------------------------------------
// arg-max of (val, idx): larger val wins, ties -> smaller idx.  Entries with idx < 0 are ignored.
struct ArgBest { double v; int i; };
static __device__ __forceinline__ ArgBest block_argbest(double v, int i, double* scratch) {
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
// of the launch must be co-resident (grids of <= 128 workgroups, one per CU, on 256 CUs; nothing else on the stream).
struct GridSync {
  unsigned long long* counter;
  unsigned long long base;
  long long timeout_ticks;
  unsigned long long* gen;   // null: every workgroup polls the arrival counter itself.  Else (optimize_lh_kernel): the word the
                    // LAST arriver of a barrier writes the barrier's index into and everybody else polls -- 128 pollers on the
                    // word the arrivals' atomics go to keep that word saturated (one word serves ~90 operations per us)
  int fences;       // 1 (default): agent-scope release before every arrival and acquire after every wait; 0 (dev switch of
                    // omp_lh.hip): none -- valid where all cross-workgroup data is written write-through and read with sc1 loads
};

// write-through (sc1) store / sc1 load of one double: the pair that is coherent across XCDs without relying on
// what the reader's L2 happens to hold
static __device__ __forceinline__ double coh_load(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
static __device__ __forceinline__ void coh_store(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The hand-off recipe of the CDNA programming guide (section 6, guideline 16), counter form:
//   producer  every wave drains its stores (s_waitcnt vmcnt(0)) -> __syncthreads() -> lane 0: agent RELEASE fence
//             (writes this XCD's dirty L2 lines back) -> explicit s_waitcnt again (ROCm 7.2 may drop the wait that
//             belongs to the fence, and the counter would overtake the write-back) -> relaxed agent fetch_add
//   consumer  lane 0 polls with RELAXED loads -> ONE agent ACQUIRE fence -> __syncthreads() -> loads
// A workgroup-scope barrier alone publishes nothing to other CUs, and a fence by lane 0 does not wait for the
// stores of the other waves -- both showed up as rare wrong results of optimize() (tests/race_hunt.py).
static __device__ __forceinline__ void grid_publish() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}
static __device__ __forceinline__ void grid_signal(const GridSync& g, unsigned long long times) {   // lane 0 only
  if (g.fences) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __hip_atomic_fetch_add(g.counter, times, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// arrive `times` times without waiting (paths that skip barriers).  All threads.
static __device__ __forceinline__ void grid_arrive(const GridSync& g, int times) {
  grid_publish();
  if (threadIdx.x == 0) grid_signal(g, (unsigned long long)times);
}
// the arrival that completes barrier `index` publishes the index (split form, g.gen != null).  Lane 0 only.
static __device__ __forceinline__ bool grid_signal_split(const GridSync& g, int index) {
  if (g.fences) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long target = g.base + (unsigned long long)index * gridDim.x;
  const unsigned long long old = __hip_atomic_fetch_add(g.counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool last = old + 1ull == target;
  if (last) __hip_atomic_store(g.gen, (unsigned long long)index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return last;
}
// arrive at barrier `index` without waiting (a workgroup that leaves the kernel before the others' last barrier).  All threads.
static __device__ __forceinline__ void grid_arrive_at(const GridSync& g, int index) {
  grid_publish();
  if (threadIdx.x == 0) {
    if (g.gen) grid_signal_split(g, index); else grid_signal(g, 1ull);
  }
}
// all threads; false after a timeout (the build is then stopped instead of hanging the GPU)
static __device__ bool grid_barrier(const GridSync& g, int index, int* s_flag) {
  grid_publish();
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int ok = 1;
    if (g.gen) {
      if (!grid_signal_split(g, index)) {
        while (__hip_atomic_load(g.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)index) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t0 > g.timeout_ticks) { ok = 0; break; }
        }
      }
    } else {
      grid_signal(g, 1ull);
      const unsigned long long target = g.base + (unsigned long long)index * gridDim.x;
      while (__hip_atomic_load(g.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > g.timeout_ticks) { ok = 0; break; }
      }
    }
    if (g.fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// ---- shared by the OMP step kernels (nnls.hip legacy forms, omp_lh.hip) -------------------------------------------------
// winner over the shards' records: (score desc, global index asc); -1 if none is valid.  One thread.
static __device__ __forceinline__ int omp_pick_record(const ApplyArgs& a, int* overflow) {
  const int recw = a.d + BCX_REC_HDR;
  int win = -1, ovf = 0;
  for (int r = 0; r < a.world; ++r) {
    const double* rec = a.recs + (size_t)r * recw;
    if (rec[3] == BCX_REC_OVERFLOW) ovf = 1;
    if (rec[3] != BCX_REC_VALID) continue;
    if (win < 0) { win = r; continue; }
    const double* best = a.recs + (size_t)win * recw;
    if (rec[0] > best[0] || (rec[0] == best[0] && rec[1] < best[1])) win = r;
  }
  *overflow = ovf;
  return win;
}

// (value desc, global index asc) arg-max over the workgroup; carries the slot of the winner.  idx < 0: no entry.
struct NegBest { double v; long long idx; int slot; };
static __device__ __forceinline__ bool negbest_better(double v, long long i, double ov, long long oi) {
  return oi >= 0 && (i < 0 || ov > v || (ov == v && oi < i));
}


void fill_nnls_args(bcx_solver* s, NnlsArgs& n, const double* recs);   // nnls.hip
struct ResolveArgs;
int bcx_launch_omp_lh(bcx_solver* s, NnlsArgs& n, const ResolveArgs* fused);   // omp_lh.hip; 1 = not applicable (LDS budget)
int bcx_launch_optimize_grid(bcx_solver* s, double tol, int k);         // nnls_grid.hip
int bcx_launch_optimize_lh(bcx_solver* s, double tol, int k, const int32_t* warm_p);   // omp_lh.hip; 1 = not applicable; warm_p: csrc/warm.hip ran
size_t bcx_warm_bytes(int k);                                            // warm.hip
int bcx_warm_start(bcx_solver* s, int k, void* buf, double* gram_work, const int32_t** p_dev);   // 1 = not applicable
