// resolve.hip -- the O(d) fp64 tail of a greedy iteration: single-workgroup, latency-bound kernels.
//   resolve_kernel      : scan partials -> candidate rows -> exact fp64 re-score -> this shard's record
//   apply_kernel<ALG>   : winner over all shards' records, reweight, monotone check / revert / retry /
//                         latch (snnls.py:41-74), next query vector
//   tail_kernel<ALG>    : both of the above in one launch (single shard)
//   begin_kernel        : start of a build() call (snnls.py:31-40) + first query
// All state is replicated: every shard runs the same code on the same gathered records, so xw,
// weights and the trace stay bit-identical across shards.
//
// Latency design: the replicated vectors (xw, b, bn) are staged into LDS at kernel entry (they do not
// depend on the scan), every phase issues all of its independent global loads before the first use,
// and reductions stay in registers (DPP / permlane butterflies) with one LDS exchange per workgroup sum.
#include <hip/hip_fp16.h>
#include "bcx_internal.h"
#include "dev_util.h"
#include "apply_common.h"

#include "resolve_core.h"
#include "tail_core.h"


// ---- kernels --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resolve_kernel(ResolveArgs a) {
  if (!a.st->active) return;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ Winner win;
  resolve_core(a, &win, nullptr, scratch);
}

template <int ALG, bool BIG>
__global__ __launch_bounds__(BCX_APPLY_THREADS) void apply_kernel(ApplyArgs a) {
  DevState* st = a.st;
  if (!st->active) return;
  extern __shared__ double dyn[];
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ int s_win, s_overflow;
  const StateVecs v = carve(vec_base<BIG>(a, dyn), a.d);
  const SlotPre pre = slot_prefetch(a);
  stage_state(a, v);
  const int tid = threadIdx.x, d = a.d;
  const int recw = d + BCX_REC_HDR;
  if (tid == 0) {
    int win = -1, ovf = 0;
    for (int r = 0; r < a.world; ++r) {
      const double* rec = a.recs + (size_t)r * recw;
      if (rec[3] == BCX_REC_OVERFLOW) ovf = 1;
      if (rec[3] != BCX_REC_VALID) continue;
      if (win < 0) { win = r; continue; }
      const double* best = a.recs + (size_t)win * recw;
      if (rec[0] > best[0] || (rec[0] == best[0] && rec[1] < best[1])) win = r;
    }
    s_win = win; s_overflow = ovf;
  }
  __syncthreads();
  if (s_overflow) { if (tid == 0) { st->active = 0; st->halt = HALT_NEED_EXACT; } return; }
  if (s_win < 0) { if (tid == 0) { st->active = 0; st->halt = HALT_DONE; } return; }
  const double* rec = a.recs + (size_t)s_win * recw;
  for (int j = tid; j < d; j += blockDim.x) v.xf[j] = rec[BCX_REC_HDR + j];
  const int64_t f = (int64_t)rec[1];
  const double nf = rec[2];
  __syncthreads();
  apply_core<ALG>(a, v, f, nf, scratch, pre);
}

// single shard: resolve + apply in one launch
template <int ALG, bool BIG>
__global__ __launch_bounds__(BCX_APPLY_THREADS) void tail_kernel(ResolveArgs r, ApplyArgs a) {
  DevState* st = a.st;
  if (!st->active) return;
  extern __shared__ double dyn[];
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ Winner win;
  BCX_STAMP(st, 0);
  const StateVecs v = carve(vec_base<BIG>(a, dyn), a.d);
  const SlotPre pre = slot_prefetch(a);
  stage_state(a, v);
  resolve_core(r, &win, v.xf, scratch);
  if (win.flags == BCX_REC_OVERFLOW) { if (threadIdx.x == 0) { st->active = 0; st->halt = HALT_NEED_EXACT; } return; }
  if (win.flags != BCX_REC_VALID) { if (threadIdx.x == 0) { st->active = 0; st->halt = HALT_DONE; } return; }
  apply_core<ALG>(a, v, win.gidx, win.norm, scratch, pre);
}

// ---- peer mailbox exchange (row-sharded builds, no host-side collective) --------------------------
// System-scope accesses: the mailboxes are fine-grained memory written by other GPUs over xGMI while
// this kernel runs, so neither the stores nor the loads may linger in this GPU's caches.
static __device__ __forceinline__ void st_sys(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
}
static __device__ __forceinline__ double ld_sys(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_SYSTEM));
}

// Publish this shard's record (hdr[4] + row[recw-4], both in LDS) to every mailbox and wait for the
// records of all shards.  Returns this shard's slots of the exchange (world records; read them with
// ld_sys) or nullptr after a timeout.  All threads of the workgroup; *s_flag is LDS scratch.
static __device__ const double* mailbox_exchange(const Mailbox& m, const double* hdr, const double* row, int* s_flag) {
  const int tid = threadIdx.x;
  const unsigned long long seq = *m.seq + 1;
  const int par = (int)(seq & 1ull);
  const long long t_in = tid == 0 ? wall_clock64() : 0;
  const size_t my_slot = m.off_slots + ((size_t)par * m.world + m.rank) * m.recw * sizeof(double);
  for (int w = 0; w < m.world; ++w) {
    double* dst = (double*)((char*)m.peers[w] + my_slot);
    for (int j = tid; j < m.recw; j += blockDim.x) st_sys(dst + j, j < BCX_REC_HDR ? hdr[j] : row[j - BCX_REC_HDR]);
  }
  __threadfence_system();          // every thread's record stores are performed before the flags below
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (explicit: ROCm 7.2 can drop the wait that belongs to a fence)
  if (tid == 0) *s_flag = 0;
  __syncthreads();
  const long long t_posted = tid == 0 ? wall_clock64() : 0;
  if (tid < m.world) {
    unsigned long long* out = (unsigned long long*)m.peers[tid] + (par * m.world + m.rank);
    __hip_atomic_store(out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long* in = (const unsigned long long*)m.peers[m.rank] + (par * m.world + tid);
    // Spin, then back off: a record that is a few microseconds away (the normal case: the peers' scans end within a few
    // percent of each other) is caught by the short sleeps; a peer that is late by milliseconds -- its kernel not
    // co-resident (ranks sharing a GPU, a profiler or a second job on its queue) -- is polled every ~50 us, which leaves
    // the memory system and this CU's other waves alone.  A flag that never arrives: the rank is recorded for the host's
    // error message (stat[5]: bit per rank, stat[6]: the exchange number).
    const long long t0 = wall_clock64();
    unsigned polls = 0;
    while (__hip_atomic_load(in, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < seq) {
      ++polls;
      if (polls < 64) __builtin_amdgcn_s_sleep(2);                                  // ~0.1 us each
      else if (polls < 256) __builtin_amdgcn_s_sleep(32);                           // ~1 us each
      else { for (int z = 0; z < 40; ++z) __builtin_amdgcn_s_sleep(127); }          // ~50 us each
      if (wall_clock64() - t0 > m.timeout_ticks) {
        atomicOr(s_flag, 1);
        atomicOr(&m.stat[5], 1ull << (tid & 63));
        m.stat[6] = seq;
        break;
      }
    }
  }
  __syncthreads();
  __threadfence_system();
  if (*s_flag) return nullptr;
  if (tid == 0) {
    *m.seq = seq;
    // device-side timing of the exchange step (bench.py: config.exchange_wait_us): how long this shard waited for
    // the slowest peer's record after posting its own, and the whole step including its own G stores
    const unsigned long long now = (unsigned long long)wall_clock64();
    const unsigned long long wait = now - (unsigned long long)t_posted, all = now - (unsigned long long)t_in;
    m.stat[0] += 1ull;
    m.stat[1] += wait; if (wait > m.stat[2]) m.stat[2] = wait;
    m.stat[3] += all;  if (all > m.stat[4]) m.stat[4] = all;
  }
  return (const double*)((const char*)m.peers[m.rank] + m.off_slots + (size_t)par * m.world * m.recw * sizeof(double));
}

// (max score, lowest global index) over the valid records; -1 if none.  One thread.
static __device__ int pick_record(const double* recs, int world, int recw, int* overflow) {
  int win = -1;
  double bs = 0.0, bi = 0.0;
  *overflow = 0;
  for (int r = 0; r < world; ++r) {
    const double* rec = recs + (size_t)r * recw;
    const double fl = ld_sys(rec + 3);
    if (fl == BCX_REC_OVERFLOW) *overflow = 1;
    if (fl != BCX_REC_VALID) continue;
    const double sc = ld_sys(rec), gi = ld_sys(rec + 1);
    if (win < 0 || sc > bs || (sc == bs && gi < bi)) { win = r; bs = sc; bi = gi; }
  }
  return win;
}

// row-sharded GIGA / FW iteration tail in one launch: resolve, exchange records with the peers, apply
template <int ALG, bool BIG>
__global__ __launch_bounds__(BCX_APPLY_THREADS) void tail_exchange_kernel(ResolveArgs r, ApplyArgs a, Mailbox m) {
  DevState* st = a.st;
  if (!st->active) return;
  extern __shared__ double dyn[];
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ Winner win;
  __shared__ double s_hdr[BCX_REC_HDR];
  __shared__ int s_flag, s_win, s_overflow;
  const int tid = threadIdx.x;
  const StateVecs v = carve(vec_base<BIG>(a, dyn), a.d);
  const SlotPre pre = slot_prefetch(a);
  stage_state(a, v);
  resolve_core(r, &win, v.xf, scratch);
  if (tid == 0) { s_hdr[0] = win.score; s_hdr[1] = (double)win.gidx; s_hdr[2] = win.norm; s_hdr[3] = win.flags; }
  __syncthreads();
  const double* recs = mailbox_exchange(m, s_hdr, v.xf, &s_flag);
  if (!recs) { if (tid == 0) { st->active = 0; st->halt = HALT_EXCHANGE_TIMEOUT; } return; }
  if (tid == 0) { int ovf; s_win = pick_record(recs, m.world, m.recw, &ovf); s_overflow = ovf; }
  __syncthreads();
  if (s_overflow) { if (tid == 0) { st->active = 0; st->halt = HALT_NEED_EXACT; } return; }
  if (s_win < 0) { if (tid == 0) { st->active = 0; st->halt = HALT_DONE; } return; }
  const double* rec = recs + (size_t)s_win * m.recw;
  if (s_win != m.rank)
    for (int j = tid; j < a.d; j += blockDim.x) v.xf[j] = ld_sys(rec + BCX_REC_HDR + j);
  const int64_t f = (int64_t)ld_sys(rec + 1);
  const double nf = ld_sys(rec + 2);
  __syncthreads();
  apply_core<ALG>(a, v, f, nf, scratch, pre);
}

// OMP: resolve + exchange; the gathered records go to `gather` for the apply kernels of nnls.hip
__global__ __launch_bounds__(BCX_APPLY_THREADS) void resolve_exchange_kernel(ResolveArgs r, Mailbox m, double* gather) {
  DevState* st = r.st;
  if (!st->active) return;
  extern __shared__ double dyn[];   // d doubles: the local winner's row
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ Winner win;
  __shared__ double s_hdr[BCX_REC_HDR];
  __shared__ int s_flag;
  const int tid = threadIdx.x;
  resolve_core(r, &win, dyn, scratch);
  if (tid == 0) { s_hdr[0] = win.score; s_hdr[1] = (double)win.gidx; s_hdr[2] = win.norm; s_hdr[3] = win.flags; }
  __syncthreads();
  const double* recs = mailbox_exchange(m, s_hdr, dyn, &s_flag);
  if (!recs) { if (tid == 0) { st->active = 0; st->halt = HALT_EXCHANGE_TIMEOUT; } return; }
  for (int j = tid; j < m.world * m.recw; j += blockDim.x) gather[j] = ld_sys(recs + j);
}

// one exchange with a known payload: checks mapping, ordering and visibility between all shards
__global__ __launch_bounds__(BCX_APPLY_THREADS) void exchange_probe_kernel(Mailbox m) {
  extern __shared__ double dyn[];   // recw - 4 payload doubles
  __shared__ double s_hdr[BCX_REC_HDR];
  __shared__ int s_flag, s_bad;
  const int tid = threadIdx.x, nrow = m.recw - BCX_REC_HDR;
  const double seq = (double)(*m.seq + 1);
  for (int j = tid; j < nrow; j += blockDim.x) dyn[j] = seq * 4096.0 + m.rank * 8192.0 * 4096.0 + j;
  if (tid == 0) { s_hdr[0] = seq; s_hdr[1] = (double)m.rank; s_hdr[2] = 0.5; s_hdr[3] = -3.0; s_bad = 0; }
  __syncthreads();
  const double* recs = mailbox_exchange(m, s_hdr, dyn, &s_flag);
  if (!recs) { if (tid == 0) *m.probe = -1; return; }
  for (int w = 0; w < m.world; ++w) {
    const double* rec = recs + (size_t)w * m.recw;
    if (tid == 0 && (ld_sys(rec) != seq || ld_sys(rec + 1) != (double)w || ld_sys(rec + 3) != -3.0)) s_bad = 1;
    for (int j = tid; j < nrow; j += blockDim.x)
      if (ld_sys(rec + BCX_REC_HDR + j) != seq * 4096.0 + w * 8192.0 * 4096.0 + j) s_bad = 1;
  }
  __syncthreads();
  if (tid == 0) *m.probe = s_bad ? -2 : 1;
}

__global__ __launch_bounds__(BCX_APPLY_THREADS) void begin_kernel(ApplyArgs a, int64_t itrs, double tol) {
  __shared__ double scratch[BCX_SCRATCH];
  DevState* st = a.st;
  if (threadIdx.x == 0) {
    st->itrs = itrs; st->it = 0; st->retried = 0; st->tol = tol;
    st->active = 1; st->halt = HALT_NONE; st->exact_mode = 0;
  }
  __syncthreads();
  refresh_state(a, scratch, st->k > 0);
  prepare_next(a, scratch);
}

__global__ __launch_bounds__(64) void resume_exact_kernel(DevState* st) {
  if (threadIdx.x == 0 && st->halt == HALT_NEED_EXACT) { st->active = 1; st->halt = HALT_NONE; st->exact_mode = 1; st->n_exact += 1; }
}

// error() outside a build: refresh xw from the slots and recompute err (snnls.py:28-29)
__global__ __launch_bounds__(BCX_APPLY_THREADS) void error_refresh_kernel(ApplyArgs a) {
  __shared__ double scratch[BCX_SCRATCH];
  refresh_state(a, scratch, a.st->k > 0);
}

// ---- launchers -------------------------------------------------------------------------------
void bcx_fill_resolve_args(bcx_solver* s, ResolveArgs& a, double* send_dev, int exact) {
  a.pv = partial_view(s->partials, s->n_partials);
  a.n_partials = s->n_partials;
  a.st = s->st;
  a.An = s->An;
  a.store_f64 = s->cfg.store_dtype == BCX_F64;
  a.store_f16 = s->cfg.store_dtype == BCX_F16;
  a.ld = s->ld;
  a.A64 = s->A64;
  a.ld64 = s->ld64;
  a.norms = s->norms;
  a.q64 = s->q64;
  a.d = s->cfg.d;
  a.alg = s->cfg.alg;
  a.n_local = s->cfg.n_local;
  a.row_offset = s->cfg.row_offset;
  // the raw fp64 scan already produced exact scores; without raw rows the fp32 arg-max is taken as is
  a.exact = exact || a.store_f64;
  a.need_score = 1;
  a.rec = send_dev;
}

#define BCX_LDS_VEC_MAX_D 3584    // 5 d doubles = 140 KiB of the CU's 160 KiB
static bool vec_big(const bcx_solver* s) { return s->cfg.d > BCX_LDS_VEC_MAX_D; }
static size_t vec_lds_bytes(const bcx_solver* s) { return vec_big(s) ? 0 : 5 * (size_t)s->cfg.d * sizeof(double); }

template <typename K> static int allow_lds(bcx_solver* s, K kfn, size_t bytes) {
  if (bytes > 48 * 1024) BCX_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return BCX_OK;
}

int bcx_launch_resolve(bcx_solver* s, double* send_dev, int exact) {
  ResolveArgs a;
  bcx_fill_resolve_args(s, a, send_dev, exact);
  hipLaunchKernelGGL(resolve_kernel, dim3(1), dim3(256), 0, s->stream, a);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

int bcx_launch_begin(bcx_solver* s, int64_t itrs, double tol) {
  ApplyArgs a;
  fill_apply_args(s, a, nullptr);
  hipLaunchKernelGGL(begin_kernel, dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, a, itrs, tol);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

int bcx_launch_resume_exact(bcx_solver* s) {
  hipLaunchKernelGGL(resume_exact_kernel, dim3(1), dim3(64), 0, s->stream, s->st);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

int bcx_launch_apply_omp(bcx_solver* s, const double* recv_dev);  // nnls.hip

int bcx_launch_apply(bcx_solver* s, const double* recv_dev) {
  if (s->cfg.alg == BCX_ALG_OMP) return bcx_launch_apply_omp(s, recv_dev);
  ApplyArgs a;
  fill_apply_args(s, a, recv_dev);
  const size_t lds = vec_lds_bytes(s);
  int rc;
  if (s->cfg.alg == BCX_ALG_GIGA) {
    if (vec_big(s)) hipLaunchKernelGGL((apply_kernel<BCX_ALG_GIGA, true>), dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, a);
    else {
      if ((rc = allow_lds(s, apply_kernel<BCX_ALG_GIGA, false>, lds))) return rc;
      hipLaunchKernelGGL((apply_kernel<BCX_ALG_GIGA, false>), dim3(1), dim3(BCX_APPLY_THREADS), lds, s->stream, a);
    }
  } else {
    if (vec_big(s)) hipLaunchKernelGGL((apply_kernel<BCX_ALG_FW, true>), dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, a);
    else {
      if ((rc = allow_lds(s, apply_kernel<BCX_ALG_FW, false>, lds))) return rc;
      hipLaunchKernelGGL((apply_kernel<BCX_ALG_FW, false>), dim3(1), dim3(BCX_APPLY_THREADS), lds, s->stream, a);
    }
  }
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

// single shard, GIGA / FW: resolve + apply fused
int bcx_launch_tail(bcx_solver* s, int exact) {
  ResolveArgs r;
  bcx_fill_resolve_args(s, r, nullptr, exact);
  r.need_score = 0;
  ApplyArgs a;
  fill_apply_args(s, a, nullptr);
  const size_t lds = vec_lds_bytes(s);
  int rc;
  if (s->cfg.alg == BCX_ALG_GIGA) {
    if (vec_big(s)) hipLaunchKernelGGL((tail_kernel<BCX_ALG_GIGA, true>), dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, r, a);
    else {
      if ((rc = allow_lds(s, tail_kernel<BCX_ALG_GIGA, false>, lds))) return rc;
      hipLaunchKernelGGL((tail_kernel<BCX_ALG_GIGA, false>), dim3(1), dim3(BCX_APPLY_THREADS), lds, s->stream, r, a);
    }
  } else {
    if (vec_big(s)) hipLaunchKernelGGL((tail_kernel<BCX_ALG_FW, true>), dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, r, a);
    else {
      if ((rc = allow_lds(s, tail_kernel<BCX_ALG_FW, false>, lds))) return rc;
      hipLaunchKernelGGL((tail_kernel<BCX_ALG_FW, false>), dim3(1), dim3(BCX_APPLY_THREADS), lds, s->stream, r, a);
    }
  }
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

Mailbox bcx_mailbox(const bcx_solver* s) {
  Mailbox m;
  m.peers = s->peer_tab;
  m.seq = s->xseq;
  m.stat = s->xseq + 1;
  m.probe = s->xprobe;
  m.world = s->cfg.world_size;
  m.rank = s->cfg.rank;
  m.recw = s->cfg.d + BCX_REC_HDR;
  m.off_slots = bcx_mailbox_slot_offset(s->cfg.world_size);
  m.timeout_ticks = (long long)(s->exchange_timeout_s * 1e8);
  return m;
}

// row shards with a peer mailbox: resolve + exchange + apply (GIGA / FW: one launch; OMP: + the NNLS kernels)
int bcx_launch_tail_exchange(bcx_solver* s, int exact) {
  ResolveArgs r;
  bcx_fill_resolve_args(s, r, nullptr, exact);
  const Mailbox m = bcx_mailbox(s);
  int rc;
  if (s->cfg.alg == BCX_ALG_OMP) {
    const size_t lds = (size_t)s->cfg.d * sizeof(double);
    if ((rc = allow_lds(s, resolve_exchange_kernel, lds))) return rc;
    hipLaunchKernelGGL(resolve_exchange_kernel, dim3(1), dim3(BCX_APPLY_THREADS), lds, s->stream, r, m, s->rec_gather);
    BCX_HIP(hipGetLastError());
    return bcx_launch_apply_omp(s, s->rec_gather);
  }
  ApplyArgs a;
  fill_apply_args(s, a, nullptr);
  const size_t lds = vec_lds_bytes(s);
  if (s->cfg.alg == BCX_ALG_GIGA) {
    if (vec_big(s)) hipLaunchKernelGGL((tail_exchange_kernel<BCX_ALG_GIGA, true>), dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, r, a, m);
    else {
      if ((rc = allow_lds(s, tail_exchange_kernel<BCX_ALG_GIGA, false>, lds))) return rc;
      hipLaunchKernelGGL((tail_exchange_kernel<BCX_ALG_GIGA, false>), dim3(1), dim3(BCX_APPLY_THREADS), lds, s->stream, r, a, m);
    }
  } else {
    if (vec_big(s)) hipLaunchKernelGGL((tail_exchange_kernel<BCX_ALG_FW, true>), dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, r, a, m);
    else {
      if ((rc = allow_lds(s, tail_exchange_kernel<BCX_ALG_FW, false>, lds))) return rc;
      hipLaunchKernelGGL((tail_exchange_kernel<BCX_ALG_FW, false>), dim3(1), dim3(BCX_APPLY_THREADS), lds, s->stream, r, a, m);
    }
  }
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

int bcx_launch_exchange_probe(bcx_solver* s) {
  const Mailbox m = bcx_mailbox(s);
  int rc;
  if ((rc = allow_lds(s, exchange_probe_kernel, (size_t)s->cfg.d * sizeof(double)))) return rc;
  hipLaunchKernelGGL(exchange_probe_kernel, dim3(1), dim3(BCX_APPLY_THREADS), (size_t)s->cfg.d * sizeof(double), s->stream, m);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

int bcx_launch_error_refresh(bcx_solver* s) {
  ApplyArgs a;
  fill_apply_args(s, a, nullptr);
  hipLaunchKernelGGL(error_refresh_kernel, dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, a);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}
