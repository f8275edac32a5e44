// Internal definitions shared by the gfx950 kernels and the C-ABI host layer.
// Not part of the public boundary (that is include/bcx.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>
#include "../../include/bcx.h"

#define BCX_CHUNK_ROWS 1024      // rows per column-sum chunk (fixed summation tree, shard-count independent)
#define BCX_MAX_CAND 64          // candidate rows re-scored in fp64 per shard per iteration
#define BCX_REC_HDR 4            // record header doubles: score, global index, norm, flags
#define BCX_REC_VALID 1.0
#define BCX_REC_OVERFLOW 2.0
#define BCX_MAX_D BCX_MAX_ROW_LENGTH   // include/bcx.h
#define BCX_SCAN_THREADS 256
#define BCX_APPLY_THREADS 256

enum { HALT_NONE = 0, HALT_DONE = 1, HALT_LIMIT = 2, HALT_NEED_EXACT = 3, HALT_EXCHANGE_TIMEOUT = 4, HALT_GRID_TIMEOUT = 5 };
enum { OMP_IDLE = 0, OMP_DONE = 1, OMP_FAST_TRY = 2, OMP_FAST_ACCEPT = 3, OMP_GENERAL = 4,
       OMP_OPT_FALLBACK = 5 };   // optimize_lh_kernel asks for the refined solve of nnls_grid.hip (its Newton check failed)

// Per-workgroup result of the correlation scan, stored as separate arrays (coalesced reads in the
// resolve step): the two best upper bounds with their local row indices, a bound on everything else
// the workgroup saw, and the best lower bound.  (For fp64 storage U == L == the score.)
struct PartialView {
  double *U1, *U2, *U3, *L;
  int32_t *i1, *i2;
};
#define BCX_PARTIAL_BYTES 40   // per workgroup: 4 doubles + 2 int32
#define BCX_MAX_PARTIALS 2048
static inline __host__ __device__ PartialView partial_view(void* base, int n) {
  PartialView v;
  v.U1 = (double*)base; v.U2 = v.U1 + n; v.U3 = v.U2 + n; v.L = v.U3 + n;
  v.i1 = (int32_t*)(v.L + n); v.i2 = v.i1 + n;
  return v;
}

// Peer mailbox: the per-iteration record exchange of a row-sharded build without a host-side collective.
// Every shard owns one fine-grained device allocation, mapped by all peers (hipIpc):
//   flags [2][world] u64  -- sequence number of the record in slot [parity][source shard]
//   slots [2][world][d+4] doubles, starting at off_slots
// Exchange number `seq` uses parity seq & 1: a shard stores its record into slot [parity][rank] of every
// mailbox, fences, stores seq into the matching flags, then waits until its own flags [parity][*] reach
// seq.  A peer can be at most one exchange ahead (it needs this shard's next record to go further), so two
// parities suffice.  Sequence numbers never restart during the life of the solver.
struct Mailbox {
  void* const* peers;          // world mailbox base addresses (device array; [rank] is this shard's own)
  unsigned long long* seq;     // exchanges completed so far (device word; identical on every shard)
  int32_t* probe;              // result of bcx_exchange_probe: 1 ok, -1 timeout, -2 payload mismatch
  unsigned long long* stat;    // [5] ranks whose record did not arrive before the timeout (bit per rank & 63), [6] that exchange's number;
                               // [0] exchanges timed, [1] sum / [2] max of the wait for the peers' records, [3] sum / [4] max
                               // of the whole exchange (own stores + wait), wall_clock64 ticks (100 MHz); bcx_exchange_stats
  int world, rank, recw;
  unsigned off_slots;
  long long timeout_ticks;     // wall_clock64 ticks (100 MHz) before a wait gives up
};

static inline unsigned bcx_mailbox_slot_offset(int world) { return (unsigned)((2u * world * 8u + 255u) / 256u * 256u); }

// Replicated solver state (device resident; every shard holds an identical copy).
struct DevState {
  int64_t itrs;        // loop iterations requested by the current build() call
  int64_t it;          // loop iterations consumed so far in this call
  int32_t active;      // kernels of the current call do work only while this is 1
  int32_t halt;        // HALT_*
  int32_t retried;     // retried_already (snnls.py:40)
  int32_t limit;       // reached_numeric_limit
  int32_t k;           // slots in the sparse weight list
  int32_t since_refresh;
  int32_t exact_mode;  // current iteration's scan was exact (no candidate window)
  int32_t zero_row;    // first zero-norm local row + 1 (0 = none)
  int32_t no_monotone; // check_error_monotone=False (snnls.py:9,45,56): no error comparison, no revert, retry flag never cleared
  int32_t np;          // size of the passive set P (OMP / optimize)
  int32_t hvalid;      // hinv == inverse of gram[P,P] and P == {slots with weight > 0}
  int32_t hlo_valid;   // hinv_lo holds the low words of that inverse (omp_lh.hip keeps it in double-double); 0 after any
                       // kernel that rewrote hinv in plain doubles
  // multi-kernel OMP step (nnls.hip): decisions handed from kernel to kernel
  int32_t omp_mode;    // OMP_* below
  int32_t omp_slot, omp_fresh, omp_checked, omp_p;
  int32_t omp_ill;     // a Schur complement below 1e-4 of its diagonal was seen: the Gram system is
                       // ill-conditioned, every step takes the refined general solve from now on
  int64_t omp_f;
  double omp_nf, omp_gff, omp_cf, omp_t, omp_inv;
  double tol;          // bc.util.TOL at build() time
  double err;          // ||A w - b||
  double nw;           // ||A w|| (1 when zero, giga.py:23)
  double bnorm;        // ||b||
  double sigma;        // sum of row norms (frankwolfe.py:25)
  int64_t n_exact;     // diagnostics: iterations that needed the exact fp64 scan (candidate window overflow)
  int64_t n_cand;      // diagnostics: candidate rows re-scored in fp64, summed over iterations
  int64_t n_resolved;  // diagnostics: resolve passes
  int64_t n_omp[4];    // diagnostics (omp_lh.hip): OMP steps, columns that left, from-scratch re-solves, extra columns entered
  double qscale;       // norm of the query vector (error bound of the fp32 scan scales with it)
  long long dbg_t[32]; // phase time stamps of the last tail (dev builds with -DBCX_TIMING, tools/tail_timing.py)
};

struct bcx_solver {
  bcx_config cfg;
  hipStream_t stream = nullptr;
  std::string err;
  int ld = 0;               // row stride (elements) of the stored normalised matrix
  int elem = 4;             // bytes per stored element
  int qelem = 4;            // bytes per element of the storage-precision query
  int ld64 = 0;             // row stride (doubles) of A64 / q64 (d rounded up to even)
  void* An = nullptr;       // n_local x ld, fp32 or fp64, rows normalised to unit norm
  double* A64 = nullptr;    // n_local x ld64 raw rows (optional)
  double* norms = nullptr;  // n_local
  double* chunk_sums = nullptr;  // n_chunks x (d+1)
  int64_t n_chunks = 0;
  void* staging = nullptr;  // upload staging when raw rows are not kept
  size_t staging_bytes = 0;
  DevState* st = nullptr;
  double* b = nullptr;      // d
  double* bn = nullptr;     // d (GIGA)
  double* xw = nullptr;     // d
  double* q64 = nullptr;    // 2 x ld64 query vectors (fp64, used by the exact re-score)
  void* qst = nullptr;      // 2 x ld query vectors in storage precision (read by the scan)
  double* tmp = nullptr;    // 4 x d scratch
  void* partials = nullptr;     // PartialView storage
  int n_partials = 0;
  double* rec_local = nullptr;   // (d+4) record produced by this shard when world_size == 1
  double* fin_part = nullptr;    // finalize: sums of groups of chunk sums
  int64_t fin_cap = 0;
  // peer mailbox (bcx_exchange_*): device-side record exchange for world_size > 1
  void* mbox = nullptr;          // this shard's mailbox (fine-grained device memory, exported over hipIpc)
  size_t mbox_bytes = 0;
  std::vector<void*> peer_mbox;  // mapped mailboxes by rank ([rank] == mbox)
  void** peer_tab = nullptr;     // device copy of peer_mbox
  unsigned long long* xseq = nullptr;   // [0] exchanges completed; [1..5] timing statistics, [6] late ranks (bit mask) and [7] the exchange
                                        // number of the last timeout (Mailbox::stat[0..6])
  int32_t* xprobe = nullptr;
  double* rec_gather = nullptr;  // world x (d+4): records of the last exchange (input of the OMP apply kernels)
  bool exchange_ready = false;
  double exchange_timeout_s = 20.0;
  // sparse weight list (selection order), grown on demand
  int64_t cap = 0;
  int64_t* act_idx = nullptr;    // global row index per slot
  double* act_w = nullptr;       // weight per slot
  double* act_rows = nullptr;    // cap x d raw rows of the slots (replicated)
  double* act_norm = nullptr;    // norm per slot
  // OMP / optimize(): Gram system over the slots
  double* gram = nullptr;        // cap x cap
  double* hinv = nullptr;        // cap x cap inverse of the passive block
  double* hinv_lo = nullptr;     // its low words (double-double inverse of the OMP step, omp_lh.hip)
  double* cvec = nullptr;        // cap : a_j . b
  int32_t* plist = nullptr;      // passive list (slot ids)
  int32_t* ppos = nullptr;       // slot -> position in plist or -1
  double* nn_x = nullptr;        // cap
  double* nn_z = nullptr;        // cap
  double* nn_wv = nullptr;       // cap
  double* nn_tmp = nullptr;      // 16*cap: position scratch t0..t3 = the first quarter of the exchange ring (grid_lh.h)
  int32_t* nn_flag = nullptr;    // cap: bit0 in problem set S, bit1 rejected, bit2 remove
  double* nn_wbak = nullptr;     // cap: weights before the step (revert on monotone failure)
  double* nn_xr = nullptr;       // 2*cap: row-phase exchange of the OMP step (omp_lh.hip)
  int64_t gram_cap = 0;
  double* gram_work = nullptr;   // optimize(): slice partials of the Gram kernel (moments.hip), grown on demand
  size_t gram_work_bytes = 0;
  void* warm_buf = nullptr;      // optimize(): buffers of the warm start (csrc/warm.hip), grown on demand
  size_t warm_bytes = 0;
  int64_t opt_warm = 0, opt_warm_failed = 0;   // optimize() calls that started warm / whose warm start was rejected by the closing check
  int64_t k_ub = 0;              // host upper bound of the slot count (grid sizing of the multi-kernel OMP step)
  unsigned long long* grid_counter = nullptr;   // [0] arrival counter of the grid barriers, [1] barrier base of the next OMP step
  uint64_t grid_epoch = 0;       // fused OMP launches since the counter was reset (bcx_build_begin)
  int64_t opt_fallbacks = 0;     // optimize() calls that took the refined solve after the incremental one's check failed (bcx_omp_stats)
  bool grid_dirty = false;       // optimize() advanced grid_counter[0] past the OMP step's base [1]: re-zero both before the next OMP step
  size_t omp_lds_allowed = 0;
  // trace of the current build() call
  int64_t trace_cap = 0;
  int64_t* tr_sel = nullptr;
  double* tr_err = nullptr;
  int32_t* tr_status = nullptr;
  bool finalized = false;
  int64_t rows_loaded = 0;
  // persist.hip: several iterations per launch
  unsigned* pflags = nullptr;    // [0] GO (tail -> scan workgroups), [1 + b] stamp of scan workgroup b; sequence numbers
  uint64_t pseq = 0;             // iterations enqueued that way so far (they never restart)
  long long* pdbg = nullptr;     // dev: time stamps of the last launch (BCX_PERSIST_DBG)
  // measurement
  bool profile = false;
  bool prof_now = false;
  int prof_every = 1;
  int64_t prof_tick = 0;
  double prof_ms = 0.0;
  int64_t prof_launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
  std::vector<int> prof_weight;  // greedy iterations each pair brackets (1; a batch for persist.hip)
  size_t prof_used = 0;
};

// ---- kernel launchers (defined in the .hip files) ------------------------
int bcx_launch_ingest(bcx_solver* s, const void* src, int src_dtype, int64_t ld_src, int64_t row_begin, int64_t rows,
                      int center);
int bcx_launch_finalize(bcx_solver* s, int have_b, const double* gathered, int64_t n_gathered);
int bcx_launch_scan(bcx_solver* s, int exact);
int bcx_launch_resolve(bcx_solver* s, double* send_dev, int exact);
int bcx_launch_begin(bcx_solver* s, int64_t itrs, double tol);
int bcx_launch_apply(bcx_solver* s, const double* recv_dev);
int bcx_launch_tail(bcx_solver* s, int exact);
int bcx_launch_persist(bcx_solver* s, int64_t iters, int64_t* covered);   // persist.hip: one launch of up to `iters` iterations; 1 = not applicable
int bcx_prof_begin(bcx_solver* s, int weight, bool every_launch);          // api.hip: event pair around a scan launch (bcx_profile_scan)
int bcx_prof_end(bcx_solver* s);
int bcx_launch_omp_fused(bcx_solver* s, int exact);   // nnls.hip: scan partials -> OMP step in one launch; 1 = not applicable
int bcx_launch_tail_exchange(bcx_solver* s, int exact);   // resolve + mailbox exchange + apply (world_size > 1)
int bcx_launch_exchange_probe(bcx_solver* s);
Mailbox bcx_mailbox(const bcx_solver* s);
int bcx_launch_resume_exact(bcx_solver* s);
int bcx_launch_error_refresh(bcx_solver* s);
int bcx_launch_optimize(bcx_solver* s, double tol);
// moments.hip: G = rows rows^T (k x k, both triangles) on the fp64 matrix cores; work: bcx_gram_rows_scratch_bytes(k, d) bytes
int64_t bcx_gram_rows_scratch_bytes(int k, int d);
int bcx_gram_rows(hipStream_t st, const double* rows, int k, int d, int64_t ld, double* G, int64_t ldg, double* work);
// gram.hip: the call counter of the stream-K kernel and the check of its time-out word (first 8 bytes of the scratch)
unsigned long long bcx_gram_sk_epoch_now();
int bcx_gram_sk_timed_out(hipStream_t st, const double* work, unsigned long long since);
int bcx_scan_grid(const bcx_solver* s);

#ifdef BCX_TIMING
#define BCX_STAMP(st, i) do { if (threadIdx.x == 0) (st)->dbg_t[i] = wall_clock64(); } while (0)
#else
#define BCX_STAMP(st, i) do {} while (0)
#endif

#define BCX_HIP(call)                                                        \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) {                                                  \
      s->err = std::string(#call) + ": " + hipGetErrorString(_e);            \
      return BCX_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)
