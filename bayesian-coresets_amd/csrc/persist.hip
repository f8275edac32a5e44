// persist.hip -- several greedy iterations of GIGA / Frank-Wolfe in ONE launch (single shard, rows of a few GB).
//
// One launch per kernel makes an iteration scan -> boundary -> tail -> boundary -> scan: while the single workgroup of
// the tail walks its chain of memory round trips (8-11 us, resolve.hip) the memory system idles, and the next scan pays
// its own ramp (dispatch of a few hundred workgroups, the first round trip of every wave) after that.  On a shard of 1 GB
// (configs[1]: 144 us of scan) the two are 7 % of the iteration.  Here the workgroups of both stay resident:
//
//   workgroup 0         the tail of every iteration (resolve_core + apply_core, the code of tail_kernel): while the scan
//                       runs it fetches the state the step needs (weight list, scalars, xw / b / bn into LDS), then waits
//                       for the scan workgroups' STAMPS (one word per workgroup: the number of the iteration whose
//                       partials it holds), resolves, re-weights, writes the next query and publishes GO = that number;
//   workgroups 1..grid  scan_body: issue the loads of their first trip of rows (the rows do not depend on the query),
//                       wait for GO, fetch the query, scan, store partials, then stamp.
//
// So per iteration the chip idles only for the dependent part of the tail (partials -> winner row -> sums -> query) and
// the two hand-offs; launch boundaries, the tail's prefetch and the scan's first round trip leave the critical path.
// Workgroups sit on different XCDs with separate L2s.  The tail's workgroup (four waves) fences: release (write-back) before
// GO, acquire (invalidate) after the stamps.  The scan's workgroups do NOT -- a fence per wave of a launch of 2048 waves
// cost 39 us per iteration, measured: their partials and stamps are write-through (sc1) stores in program order behind
// one wait, and they fetch the query, its scale and the state machine's switch with sc1 loads.  All workgroups must be resident
// together: the launcher checks the occupancy and takes the one-launch-per-kernel path (api.hip) otherwise, every wait
// has a time-out that stops the state machine (HALT_GRID_TIMEOUT -> an error from bcx_build_poll, never a wrong result),
// and workgroup 0 -- dispatched first -- is the one everybody waits for.  Sequence numbers never restart during the
// life of a solver, so no word needs resetting between launches.
//
// Same partials, same winner, same re-weight arithmetic as scan_kernel + tail_kernel: traces are bit-identical
// (tests/test_persist.py).  Reference loop: snnls.py:41-74 (one pass = _select + _reweight).
#include <stdlib.h>
#include <math.h>
#include <hip/hip_fp16.h>
#include "bcx_internal.h"
#include "dev_util.h"
#include "scan_core.h"
#include "tail_core.h"

static_assert(BCX_SCAN_THREADS == BCX_APPLY_THREADS, "one workgroup shape for both roles");

struct PersistArgs {
  int iters;                 // iterations of this launch
  unsigned seq0;             // iterations of earlier launches (GO and every stamp are <= seq0 when the launch starts)
  unsigned* go;              // tail -> scan workgroups: iterations whose re-weight is done and whose next query is written
  unsigned* stamp;           // scan workgroup b -> tail: iteration (1-based, + seq0) whose partials slot b holds
  long long timeout_ticks;   // wall_clock64 ticks (100 MHz)
  long long* dbg;            // dev (BCX_PERSIST_DBG): 8 time stamps per iteration of the launch, tools/persist_timeline.py
};
#define PDBG(slot) do { if (p.dbg && tid == 0) p.dbg[it * 8 + (slot)] = wall_clock64(); } while (0)

static __device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
static __device__ __forceinline__ void st_agent(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every thread: its own stores are performed (written back beyond this XCD's L2) before anything that follows
static __device__ __forceinline__ void release_mine() {
  __threadfence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (explicit: ROCm 7.2 can drop the wait that belongs to a fence)
}
static __device__ __forceinline__ void acquire_all() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); }

// The tail of every iteration of a launch: workgroup 0.  (Inlined into the kernel: as a function of its own its arguments
// live on the stack, and every field read after the acquire below is a trip to memory -- measured: resolve 7.1 us, where
// the one-launch-per-kernel tail takes 8 us for everything.  Inlined, the two roles together make the compiler take ~245
// VGPRs: two workgroups per CU, which is what the scan's launch width asks for anyway; the launcher trims the width by one
// workgroup where it has to.)
template <int ALG>
static __device__ __forceinline__ void persist_tail(const ResolveArgs& r, const ApplyArgs& ap, const PersistArgs& p) {
  DevState* st = ap.st;
  const int tid = threadIdx.x;
  __shared__ int s_flag;
  extern __shared__ double dyn[];
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ Winner win;
  const StateVecs v = carve(dyn, ap.d);
  for (int it = 0; it < p.iters; ++it) {
    if (!st->active) break;
    BCX_STAMP(st, 0);
    PDBG(0);
    const SlotPre pre = slot_prefetch(ap);
    stage_state(ap, v);
    if (tid == 0) s_flag = 0;
    __syncthreads();
    {
      const unsigned want = p.seq0 + (unsigned)it + 1u;
      const long long t0 = wall_clock64();
      unsigned polls = 0;
      bool late = false;
#pragma unroll 1
      for (int t = 0; t < PP_MAX && !late; ++t) {
        const int q = tid + t * (BCX_MAX_PARTIALS / PP_MAX);
        if (q >= r.n_partials) break;
        while (ld_agent(p.stamp + q) != want) {
          if (polls < 4096) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(32);
          if ((++polls & 63u) == 0 && wall_clock64() - t0 > p.timeout_ticks) { late = true; break; }
        }
      }
      if (late) s_flag = 1;
    }
    __syncthreads();
    if (s_flag) {
      if (tid == 0) { st->active = 0; st->halt = HALT_GRID_TIMEOUT; }
      break;
    }
    PDBG(1);
    acquire_all();
    resolve_core(r, &win, v.xf, scratch);
    PDBG(2);
    if (win.flags == BCX_REC_OVERFLOW) { if (tid == 0) { st->active = 0; st->halt = HALT_NEED_EXACT; } }
    else if (win.flags != BCX_REC_VALID) { if (tid == 0) { st->active = 0; st->halt = HALT_DONE; } }
    else apply_core<ALG>(ap, v, win.gidx, win.norm, scratch, pre);
    // the next query and the state are written: publish
    PDBG(3);
    release_mine();
    __syncthreads();
    if (tid == 0) st_agent(p.go, p.seq0 + (unsigned)it + 1u);
    PDBG(4);
  }
  // nobody may be left waiting, whatever stopped the loop (a scan workgroup looks at st->active after GO)
  release_mine();
  __syncthreads();
  if (tid == 0) st_agent(p.go, p.seq0 + (unsigned)p.iters);
}

template <typename ST, bool DUAL, int G, int CH, int UR>
__global__ __launch_bounds__(BCX_SCAN_THREADS) void scan_persist_kernel(ScanArgs a, ResolveArgs r, ApplyArgs ap, PersistArgs p) {
  DevState* st = ap.st;
  const int tid = threadIdx.x;
  __shared__ int s_flag;
  if (blockIdx.x == 0) {
    persist_tail<DUAL ? BCX_ALG_GIGA : BCX_ALG_FW>(r, ap, p);
    return;
  }
  // ---- the scan of every iteration ----
  const unsigned blk = blockIdx.x - 1, nblk = gridDim.x - 1;
  for (int it = 0; it < p.iters; ++it) {
    auto gate = [&]() -> bool {
      if (it > 0) {   // (the first query of a launch was written by an earlier launch)
        if (tid == 0) {
          const unsigned want = p.seq0 + (unsigned)it;
          const long long t0 = wall_clock64();
          unsigned polls = 0;
          int bad = 0;
          while ((int)(ld_agent(p.go) - want) < 0) {
            if (polls < 4096) __builtin_amdgcn_s_sleep(2); else __builtin_amdgcn_s_sleep(32);
            if ((++polls & 63u) == 0 && wall_clock64() - t0 > p.timeout_ticks) { bad = 1; break; }
          }
          s_flag = bad;
        }
        __syncthreads();
        if (s_flag) {
          if (tid == 0) { st->active = 0; st->halt = HALT_GRID_TIMEOUT; }
          return false;
        }
      }
      if (blk == 0) PDBG(5);
      if (p.dbg && tid == 0 && it == p.iters - 1) p.dbg[1024 + 2 * blk] = wall_clock64();      // last iteration: every workgroup's pass
      return true;     // (scan_body reads the query and the state machine's switch with sc1 loads: no fence here)
    };
    if (!scan_body<ST, DUAL, G, CH, UR>(a, blk, nblk, gate)) return;
    if (blk == 0) PDBG(6);
    if (p.dbg && tid == 0 && it == p.iters - 1) p.dbg[1024 + 2 * blk + 1] = wall_clock64();
    if (tid == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the partials (sc1 stores of this thread) are performed
      st_agent(p.stamp + blk, p.seq0 + (unsigned)it + 1u);
    }
    __syncthreads();   // (the workgroup's LDS exchange of scan_body is reused by the next pass)
  }
}

// ---- host side ----------------------------------------------------------------------------
template <bool DUAL>
static int launch_p(bcx_solver* s, const ScanArgs& a, const ResolveArgs& r, const ApplyArgs& ap, const PersistArgs& p,
                    const ScanPlan& pl, size_t lds, int cus, bool launch, int64_t* resident) {
#define L(GG, CC, UU)                                                                                                    \
  if (pl.G == GG && pl.CH == CC && pl.UR == UU) {                                                                        \
    auto kfn = scan_persist_kernel<float, DUAL, GG, CC, UU>;                                                             \
    int per_cu = 0;                                                                                                      \
    BCX_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kfn, BCX_SCAN_THREADS, lds));             \
    if (per_cu < 1) return 1;                                                                                            \
    *resident = (int64_t)per_cu * cus;                                                                                   \
    if (!launch) return BCX_OK;                                                                                          \
    hipLaunchKernelGGL(kfn, dim3(pl.grid + 1), dim3(BCX_SCAN_THREADS), lds, s->stream, a, r, ap, p);                     \
    return BCX_OK;                                                                                                       \
  }
  L(16, 1, 4) L(32, 1, 4) L(64, 1, 4) L(64, 2, 4) L(64, 4, 2) L(64, 8, 1)
  L(16, 1, 8) L(32, 1, 8) L(64, 1, 8) L(64, 2, 8) L(64, 4, 4) L(64, 8, 2)
#undef L
  return 1;
}

// Opt-in: BCX_PERSIST=1 in the environment.  Measured on one MI355X against one launch per kernel, interleaved
// (tools/persist_ab.sh, profiles/r06_persist_ab.txt): +3-6 % at 200 MB per pass, +0.3-1 % at 1-5 GB, -4 .. +2 % below
// 100 MB -- the launch boundaries this form removes cost ~1.3 us each and the one-launch tail already fetches its state in
// the shadow of its first round trip, so what is left to win is the scan's ramp.  Not the default: a launch whose
// workgroups wait for each other must have the GPU to itself (two such processes on one GPU can starve each other into the
// time-out), which one launch per kernel never needs.  Read at every call: a process may switch it between builds.
static bool persist_allowed() {
  const char* e = getenv("BCX_PERSIST");
  return e && e[0] == '1';
}

static int persist_one(bcx_solver* s, int64_t iters, int64_t* covered);

// ONE launch of up to `iters` greedy iterations (scan + tail each; at most ~100 GB of scanning).  Returns 1 when this form
// does not apply (the caller enqueues one launch per kernel), BCX_OK with *covered = the iterations this launch holds.
int bcx_launch_persist(bcx_solver* s, int64_t iters, int64_t* covered) {
  *covered = 0;
  if (!persist_allowed() || iters < 2) return 1;
  const int rc0 = persist_one(s, iters, covered);
  // tests: BCX_PERSIST_REQUIRE=1 turns "does not apply" into an error, so a parity test cannot pass on the other path
  if (rc0 == 1 && bcx_dev_env("BCX_PERSIST_REQUIRE")) { s->err = "persist: the batched form does not apply to this solver"; return BCX_ERR_STATE; }
  return rc0;
}

static int persist_one(bcx_solver* s, int64_t iters, int64_t* covered) {
  if (s->cfg.world_size != 1 || s->cfg.store_dtype != BCX_F32) return 1;
  if (s->cfg.alg != BCX_ALG_GIGA && s->cfg.alg != BCX_ALG_FW) return 1;
  const size_t lds = 5 * (size_t)s->cfg.d * sizeof(double);
  if (lds > 32 * 1024) return 1;
  const double bytes = (double)s->cfg.n_local * s->ld * s->elem;
  double max_gb = 6.0;         // beyond this the tail is < 0.5 % of an iteration: one launch per kernel, as measured and profiled
  if (const char* e = bcx_dev_env("BCX_PERSIST_MAX_GB")) max_gb = atof(e);
  if (bytes > max_gb * 1e9 || s->cfg.n_local < 1) return 1;
  ScanArgs a;
  ScanPlan pl;
  int rc = bcx_scan_plan(s, 0, &a, &pl);
  if (rc != BCX_OK) return rc;
  if (pl.f64 || pl.f16 || pl.long_rows) return 1;
  if (!s->pflags) {
    unsigned* f = nullptr;
    BCX_HIP(hipMalloc((void**)&f, (1 + BCX_MAX_PARTIALS) * sizeof(unsigned)));
    BCX_HIP(hipMemset(f, 0, (1 + BCX_MAX_PARTIALS) * sizeof(unsigned)));
    s->pflags = f;
    s->pseq = 0;
  }
  int cus = 0;
  BCX_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, s->cfg.device));
  int64_t chunk = (int64_t)(100e9 / (bytes > 1.0 ? bytes : 1.0));
  if (const char* e = bcx_dev_env("BCX_PERSIST_CHUNK")) chunk = atol(e);
  chunk = chunk < 2 ? 2 : (chunk > 128 ? 128 : chunk);
  const int64_t now = iters < chunk ? iters : chunk;
  ResolveArgs r;
  const int np_before = s->n_partials;
  s->n_partials = pl.grid;
  bcx_fill_resolve_args(s, r, nullptr, 0);
  r.need_score = 0;
  ApplyArgs ap;
  fill_apply_args(s, ap, nullptr);
  PersistArgs p;
  p.iters = (int)now;
  p.seq0 = (unsigned)s->pseq;
  p.go = s->pflags;
  p.stamp = s->pflags + 1;
  p.timeout_ticks = (long long)(5.0 * 1e8);
  p.dbg = nullptr;
  if (bcx_dev_env("BCX_PERSIST_DBG")) {
    if (!s->pdbg) { BCX_HIP(hipMalloc((void**)&s->pdbg, (1024 + 2 * BCX_MAX_PARTIALS) * sizeof(long long))); }
    BCX_HIP(hipMemsetAsync(s->pdbg, 0, (1024 + 2 * BCX_MAX_PARTIALS) * sizeof(long long), s->stream));
    p.dbg = s->pdbg;
  }
  // all workgroups of the launch resident together: the scan's width is trimmed to what fits beside the tail's workgroup
  // (another split of the rows over workgroups: other partials, the same winner -- the arg-max is exact whatever the split)
  int64_t resident = 0;
  rc = pl.dual ? launch_p<true>(s, a, r, ap, p, pl, lds, cus, false, &resident) : launch_p<false>(s, a, r, ap, p, pl, lds, cus, false, &resident);
  if (rc != BCX_OK) { s->n_partials = np_before; return rc; }
  if (resident < 2) { s->n_partials = np_before; return 1; }
  if (pl.grid > resident - 1) {
    if (resident - 1 < pl.grid / 2) { s->n_partials = np_before; return 1; }    // (far from the measured launch width: the other path)
    pl.grid = (int)(resident - 1);
    a.out = partial_view(s->partials, pl.grid);
    s->n_partials = pl.grid;
    bcx_fill_resolve_args(s, r, nullptr, 0);
    r.need_score = 0;
  }
  if ((rc = bcx_prof_begin(s, (int)now, true))) return rc;
  rc = pl.dual ? launch_p<true>(s, a, r, ap, p, pl, lds, cus, true, &resident) : launch_p<false>(s, a, r, ap, p, pl, lds, cus, true, &resident);
  if (rc != BCX_OK) return rc;
  BCX_HIP(hipGetLastError());
  if ((rc = bcx_prof_end(s))) return rc;
  s->pseq += (uint64_t)now;
  *covered = now;
  return BCX_OK;
}

// dev: the time stamps of the last launch (8 per iteration, 128 iterations; then begin / end of every scan workgroup's pass in the
// launch's last iteration; zero = not written); not in include/bcx.h
extern "C" int bcx_debug_persist(bcx_solver* s, long long* out1024) {   // (1024 + 2 x 2048 words)
  if (!s || !out1024 || !s->pdbg) return BCX_ERR_ARG;
  BCX_HIP(hipStreamSynchronize(s->stream));
  BCX_HIP(hipMemcpy(out1024, s->pdbg, (1024 + 2 * BCX_MAX_PARTIALS) * sizeof(long long), hipMemcpyDeviceToHost));
  return BCX_OK;
}
