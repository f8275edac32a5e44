/*
 * bcx.h -- C ABI of the MI355X (gfx950) coreset-construction engine.
 *
 * One shared library (libbcx.so, built by hipcc --offload-arch=gfx950) replaces
 * the NumPy/SciPy arithmetic behind the reference's sparse-NNLS solvers.  The
 * reference (trevorcampbell/bayesian-coresets @ v0.9.1) has no FFI of its own
 * (it is pure Python), so each entry point below names the reference interface
 * whose work it takes over; INTEGRATION.md shows the ctypes stub a maintainer
 * of the reference would add.
 *
 * Conventions: plain C types only; every function returns an int status
 * (BCX_OK == 0, negative == error, message via bcx_last_error); no C++
 * exception crosses the boundary; the opaque handle owns all device memory;
 * outputs are caller-allocated HOST buffers unless a parameter says "dev";
 * device pointers are passed as const void* (from torch.Tensor.data_ptr()).
 * One host thread per handle.
 */
#ifndef BCX_H
#define BCX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bcx_solver bcx_solver;

/* ---- enums ---------------------------------------------------------- */
enum { BCX_ALG_GIGA = 0, BCX_ALG_FW = 1, BCX_ALG_OMP = 2 };
enum { BCX_F32 = 0, BCX_F64 = 1, BCX_F16 = 2 };   /* BCX_F16: storage of the normalised rows only */

/* return codes */
enum {
  BCX_OK = 0,
  BCX_ERR_ARG = -1,        /* bad argument / call order */
  BCX_ERR_HIP = -2,        /* HIP runtime failure (message has hipGetErrorString) */
  BCX_ERR_ZERO_ROW = -3,   /* a data row has zero norm: reference raises ValueError (giga.py:11-12) */
  BCX_ERR_ZERO_B = -4,     /* ||b|| == 0 for GIGA: reference raises NumericalPrecisionError (giga.py:16-17) */
  BCX_ERR_NOMEM = -5,
  BCX_ERR_STATE = -6,      /* solver not initialised / already latched where not allowed */
  BCX_ERR_EXCHANGE = -7,   /* peer mailbox: a shard's record did not arrive in time (see bcx_exchange_attach) */
  BCX_ERR_TIMEOUT = -8     /* a wait between workgroups of one launch expired (GPU shared / preempted): the result is not valid */
};

/* per-iteration status written to the trace (snnls.py:41-74 outcome of one loop iteration) */
enum {
  BCX_IT_OK = 0,            /* step accepted */
  BCX_IT_FAIL_SELECT = 1,   /* NumericalPrecisionError raised inside _select   (giga.py:28-29) */
  BCX_IT_FAIL_REWEIGHT = 2, /* NumericalPrecisionError raised inside _reweight (giga.py:50-51, frankwolfe.py:33-34) */
  BCX_IT_FAIL_MONOTONE = 3  /* error increased, weights reverted               (snnls.py:58-61) */
};

typedef struct bcx_config {
  int32_t alg;             /* BCX_ALG_*                                                          */
  int32_t store_dtype;     /* BCX_F32: normalised rows stored fp32 (default); BCX_F64: exact mode;
                              BCX_F16: fp16 rows (half the bytes per iteration; needs keep_exact_rows for the fp64 re-score) */
  int32_t keep_exact_rows; /* 1: also keep the raw fp64 rows resident (exact reweight + fp64 rescue) */
  int32_t device;          /* HIP device ordinal                                                  */
  int32_t d;               /* projection dimension (columns of the N x d vector matrix)           */
  int32_t world_size;      /* number of row shards (1 = single GPU)                               */
  int32_t rank;            /* this shard                                                          */
  int32_t refresh_every;   /* recompute xw = sum_j w_j A[:,j] from the active rows every this many accepted steps (0 = default 64) */
  int64_t n_local;         /* rows held by this shard                                             */
  int64_t n_global;        /* rows over all shards                                                */
  int64_t row_offset;      /* global index of local row 0 (contiguous row blocks, lowest index wins ties) */
} bcx_config;

/* ---- lifetime -------------------------------------------------------- */
/* Replaces SparseNNLS.__init__ storage (snnls/snnls.py:9-16): allocates An (N x d, store_dtype),
 * norms (fp64), optional raw rows (fp64) and the replicated O(d) solver state. */
/* Longest supported row (the projection dimension d of the vectors; the reference takes any): a bound on the argument, not
 * a capacity of any kernel.  Up to 4096 floats / 2048 doubles of stored row the scan keeps the query in registers; longer rows
 * take a one-wave-per-row form with the query in LDS (its tail beyond 144 KiB is read from global memory), the O(d) state
 * kernels keep their five d-vectors in LDS up to d = 3584 and in global scratch beyond, the constructor pass adds column sums
 * in LDS up to d = 18432 and in place beyond, the OMP step takes its multi-kernel form once three d-vectors exceed its LDS
 * budget, and row shards of more than 18432 values exchange their records by all-gather instead of the peer mailbox. */
#define BCX_MAX_ROW_LENGTH 1048576
int bcx_create(const bcx_config* cfg, bcx_solver** out);
int bcx_destroy(bcx_solver* s);
/* Last error text for this handle (or for a failed bcx_create when s == NULL). */
const char* bcx_last_error(const bcx_solver* s);
/* Use the caller's HIP stream (torch.cuda.current_stream().cuda_stream); NULL = default stream. */
int bcx_set_stream(bcx_solver* s, void* hip_stream);

/* ---- ingest: GIGA/FW/OMP constructors (giga.py:8-13, frankwolfe.py:7-13, orthopursuit.py:9-15) --- */
/* Copy `rows` rows starting at local row `row_begin` from src (host or device memory, fp32 or fp64,
 * row-major with leading dimension ld elements), compute their norms (fp64), store the normalised
 * rows and accumulate per-chunk column sums / norm sums.  May be called repeatedly (row shards,
 * streaming upload).  A zero-norm row is reported by bcx_finalize (BCX_ERR_ZERO_ROW). */
int bcx_load_rows(bcx_solver* s, const void* src, int32_t src_is_device, int32_t src_dtype,
                  int64_t row_begin, int64_t rows, int64_t ld);
/* The same with options.  BCX_LOAD_CENTER_ROWS: subtract every row's mean before anything else, i.e. the rows are
 * raw log-likelihoods and the constructor pass applies the projector's centring (projector.py:21) itself -- the row
 * is in registers between the norm and the stores, so a projection that feeds HilbertCoreset is written once and read
 * once instead of taking a separate centring pass.  The caller's buffer is left as it is (uncentred). */
#define BCX_LOAD_CENTER_ROWS 1
int bcx_load_rows_flags(bcx_solver* s, const void* src, int32_t src_is_device, int32_t src_dtype,
                        int64_t row_begin, int64_t rows, int64_t ld, int32_t flags);
/* Number of row chunks on this shard and rows per chunk; chunk sums are (d+1) doubles each:
 * d column sums followed by the sum of row norms.  Device pointer for the all-gather across shards. */
int bcx_chunk_sums(bcx_solver* s, const void** dev_ptr, int64_t* n_chunks, int64_t* chunk_rows);
/* Copy this shard's chunk sums into a caller-owned device buffer (e.g. a torch tensor that is then
 * all-gathered); asynchronous on the stream.  cap_chunks >= n_chunks. */
int bcx_export_chunk_sums(bcx_solver* s, void* dst_dev, int64_t cap_chunks);
/* Finish construction.  b_host (d doubles) overrides the column sums when non-NULL -- the reference
 * solver constructors take b from the caller (snnls.py:9; hilbert.py:24 passes vecs.sum(axis=0)).
 * gathered_sums_dev (optional): chunk sums of ALL shards in global chunk order (n_gathered chunks),
 * summed in that fixed order so b and sum(Anorms) are bit-identical for any shard count.
 * Returns BCX_ERR_ZERO_ROW / BCX_ERR_ZERO_B exactly where the reference constructors raise. */
int bcx_finalize(bcx_solver* s, const double* b_host, const void* gathered_sums_dev, int64_t n_gathered);

/* ---- the hot loop: SparseNNLS.build (snnls.py:31-79) ---------------------------------------- */
/* Begin a build() call of `itrs` loop iterations (resets the per-call retry flag, snnls.py:40;
 * tol is bc.util.TOL read at call time, giga.py:28).  Returns 1 in *skip if the reference would
 * return immediately (latched numeric limit snnls.py:32-34, or no data :36-38). */
int bcx_build_begin(bcx_solver* s, int64_t itrs, double tol, int32_t* skip);
/* Enqueue one greedy iteration up to the shard exchange: correlation scan over the local rows
 * (giga.py:31-38 / frankwolfe.py:16-17 / orthopursuit.py:18-26) + exact fp64 re-score of the
 * candidates; writes this shard's record {score, global index, norm, flags, row[d]} (d+4 doubles)
 * to send_dev.  Asynchronous on the stream. */
int bcx_step_scan(bcx_solver* s, void* send_dev);
/* Enqueue the replicated part: pick the winner among world_size records at recv_dev (max score,
 * lowest global index), OMP negative direction (orthopursuit.py:27-35), reweight
 * (giga.py:40-64 / frankwolfe.py:19-40 / orthopursuit.py:37-42), monotone check / revert / retry /
 * latch (snnls.py:56-74), and the next query vector.  Asynchronous on the stream. */
int bcx_step_apply(bcx_solver* s, const void* recv_dev);
/* Enqueue up to `itrs` whole iterations (scan + resolve + apply) without host round trips: single
 * shard, or row shards after bcx_exchange_attach (the record exchange then happens on the device).
 * bcx_build_enqueue_exact enqueues one iteration with the exact fp64 scan (after *need_exact). */
int bcx_build_enqueue(bcx_solver* s, int64_t itrs);
int bcx_build_enqueue_exact(bcx_solver* s);
/* ---- peer mailbox: device-side record exchange between row shards ------------
 * Replaces the per-iteration all-gather of bcx_step_scan / bcx_step_apply by direct stores into the
 * peers' mailboxes over xGMI, issued by the iteration's own tail kernel, so a sharded build needs no
 * host-side collective and no host round trip per iteration.  One process per GPU on one node:
 *   bcx_exchange_export  allocate this shard's mailbox and return its hipIpc handle (64 bytes);
 *   (the caller all-gathers the handles, e.g. torch.distributed.all_gather_object)
 *   bcx_exchange_attach  map the mailboxes of all shards (handles in rank order, handle_bytes apart);
 *                        timeout_s bounds every wait for a peer (<= 0: default 20 s), after which the
 *                        build stops and bcx_build_poll returns BCX_ERR_EXCHANGE instead of hanging;
 *   bcx_exchange_probe   COLLECTIVE: one exchange with a known payload; *result = 1 if every shard's
 *                        record arrived intact, -1 timeout, -2 payload mismatch;
 *   bcx_exchange_set_timeout  change the bound on every later wait;
 *   bcx_exchange_disable go back to the host-driven exchange (e.g. after a failed probe).
 *   bcx_exchange_stats   device-side time stamps of the exchanges since the last reset: how many, mean / max microseconds
 *                        this shard waited for the slowest peer's record after posting its own, mean / max of the whole
 *                        exchange step (own stores + wait); any output may be NULL; reset != 0 clears the counters.
 * All shards must issue the same sequence of exchanges (they do: the solver state is replicated). */
int bcx_exchange_export(bcx_solver* s, void* handle_out, int32_t handle_bytes);
int bcx_exchange_attach(bcx_solver* s, const void* handles, int32_t handle_bytes, double timeout_s);
int bcx_exchange_probe(bcx_solver* s, int32_t* result);
int bcx_exchange_set_timeout(bcx_solver* s, double timeout_s);
int bcx_exchange_disable(bcx_solver* s);
int bcx_exchange_stats(bcx_solver* s, int64_t* n, double* wait_us_mean, double* wait_us_max, double* total_us_mean,
                       double* total_us_max, int32_t reset);
/* Synchronise and report.  *n_done = loop iterations consumed so far in this build() call;
 * *need_exact = 1 if the engine stopped before an iteration because the fp32 candidate window
 * overflowed (tie-heavy data) -- call bcx_step_scan_exact for that iteration and continue;
 * *limit = reached_numeric_limit (snnls.py:67). */
int bcx_build_poll(bcx_solver* s, int64_t* n_done, int32_t* need_exact, int32_t* limit);
/* Exact (all-fp64) variant of bcx_step_scan used for the overflow fallback and by store_dtype F64. */
int bcx_step_scan_exact(bcx_solver* s, void* send_dev);
/* Copy the trace of this build() call: per loop iteration the selected global index (-1 if select
 * failed), the error after the iteration and a BCX_IT_* status.  Arrays sized >= n_done. */
int bcx_build_trace(bcx_solver* s, int64_t* sel, double* err, int32_t* status, int64_t cap, int64_t* n_out);

/* ---- read-out: weights()/size()/error() (snnls.py:22-29), optimize() (:82-97), reset() (:18-20) -- */
int bcx_active_count(bcx_solver* s, int64_t* k);                 /* entries in the sparse weight list (any weight) */
int bcx_get_weights(bcx_solver* s, int64_t* idx, double* w, int64_t cap, int64_t* k); /* selection order */
int bcx_error(bcx_solver* s, double* err);
int bcx_optimize(bcx_solver* s, double tol, int32_t* accepted);
int bcx_reset(bcx_solver* s);
/* SparseNNLS(A, b, check_error_monotone) (snnls.py:9,16): on = 0 switches the per-iteration error comparison and the
 * revert off (snnls.py:45-47,56-62); the retry flag is then never refreshed, exactly as in the reference where the
 * refresh sits inside the monotone branch.  Default on.  Survives bcx_reset; call between build() calls only. */
int bcx_set_check_monotone(bcx_solver* s, int32_t on);
int bcx_reached_numeric_limit(bcx_solver* s, int32_t* limit);

/* ---- introspection / measurement ------------------------------------------------------------ */
/* Copy replicated vectors to the host for tests: which = 0:b 1:xw 2:query0 3:query1 (d doubles). */
int bcx_get_vector(bcx_solver* s, int32_t which, double* out);
/* Copy row norms of local rows [begin, begin+count) (Anorms, frankwolfe.py:10). */
int bcx_get_norms(bcx_solver* s, int64_t begin, int64_t count, double* out);
/* One N-way correlation scan with a caller-supplied query (d doubles, host): *idx = arg-max_n of
 * An[n] . query (first maximum, global row index), *score = its exact fp64 value.  The select step of
 * SparseVI on projected vectors (sparsevi.py:49-56).  FW / OMP handles only. */
int bcx_argmax_correlation(bcx_solver* s, const double* query_host, int64_t* idx, double* score);
/* Time `reps` launches of the correlation-scan kernel alone with hipEvents on the solver's stream;
 * returns the mean milliseconds per launch and the algorithmic bytes one launch reads. */
int bcx_time_scan(bcx_solver* s, int32_t reps, int32_t exact, double* ms_per_launch, double* bytes_per_launch);
/* Diagnostics since construction: iterations that fell back to the exact fp64 scan, candidate rows re-scored
 * in fp64 (total) and resolve passes (candidates / resolves = mean candidates per iteration). */
int bcx_stats(bcx_solver* s, int64_t* exact_fallbacks, int64_t* candidates, int64_t* resolves);
/* Sum of scan-kernel time recorded by hipEvents during bcx_build_enqueue: on = 1 times every scan launch,
 * on = N > 1 every N-th one (an event pair costs several microseconds of stream time), on = 0 stops. */
/* OMP step diagnostics since construction: out4 = {steps taken, columns that left the passive set, from-scratch re-solves
 * (the incremental inverse had drifted, or a reverted step; also optimize() calls whose incremental solve failed its closing
 * Newton check and were redone by the refined solve), columns that entered beyond the selected one}. */
int bcx_omp_stats(bcx_solver* s, int64_t* out4);
int bcx_profile_scan(bcx_solver* s, int32_t on);
int bcx_profile_read(bcx_solver* s, double* scan_ms_total, int64_t* scan_launches);
/* ---- device-native projection (projector.py:19-21 with the example likelihoods) --------------------
 * vecs[n][s] = loglik(z_n, theta_s) - mean over s, for family 0 = logistic (model_lr.py:25-32),
 * 1 = Poisson/softplus (model_poiss.py:25-38), 2 = Gaussian linear regression (model_linreg.py:4-10;
 * param = sigma^2).  Z_dev: N x ldz doubles (features in columns [0, D), response in column ycol for
 * families 1 and 2), theta_dev: S x ldt doubles.  Stateless; asynchronous on `stream`.
 *   write  : vecs (N x S) into out_dev; rowsum_dev: unused (may be NULL; earlier builds wanted N doubles of scratch)
 *   colsum : sum_n vecs[n][s] without materialising vecs (sparsevi.py:70-74); work_dev = 2048*S doubles
 *   select : arg-max_n vecs[n].resid / ||vecs[n]|| / S (sparsevi.py:49-51) -> result_dev = {double, int64};
 *            work_dev = 2048 doubles + 2048 int64 */
int bcx_project_write(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                      int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                      void* out_dev, int64_t ldo, void* rowsum_dev);
 /* write_raw: the log-likelihoods WITHOUT the centring (one pass less); the rows are centred by whoever reads them, e.g.
  * bcx_load_rows_flags(..., BCX_LOAD_CENTER_ROWS) -- HilbertCoreset behind a device projector. */
int bcx_project_write_raw(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                          int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                          void* out_dev, int64_t ldo);
/* The same for the POINTS of a coreset (sparsevi.py:38-39): rows that every shard holds and projects alike, so the kernel may
 * be chosen by their number (up to 4096 rows: 32 x 32 blocks straight from L2; the data rows of bcx_project_write keep one
 * kernel whatever their shard's size, so that row-sharded builds reproduce the single-shard one bit for bit).  center != 0:
 * centred rows as bcx_project_write, 0: raw as bcx_project_write_raw. */
int bcx_project_write_points(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                             int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                             void* out_dev, int64_t ldo, int32_t center);
int bcx_project_colsum(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                       int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                       void* colsum_dev, void* work_dev);
int bcx_project_select(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                       int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                       const void* resid_dev, double resid_sum, void* result_dev, void* work_dev);
/* The select step with ALL of its scratch supplied by the caller (it then never touches the stream-ordered allocator):
 * work_dev holds bcx_project_select_scratch_bytes(family, N, S) bytes = 2048 doubles + 2048 int64 for the arg-max
 * reduction, then 32 bytes per row and 64-column group for the partial row moments. */
int64_t bcx_project_select_scratch_bytes(int32_t family, int64_t N, int32_t S);
int bcx_project_select_ws(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                          int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                          const void* resid_dev, double resid_sum, void* result_dev, void* work_dev, int64_t work_bytes);
/* Closed-form column sums of the linear-regression family (family 2) from the one-time second moments of the data:
 *   sum_n (y_n - x_n.theta)^2 = yy - 2 theta^T X^T y + theta^T X^T X theta
 * so the 1 + opt_itrs full-data column sums of a SparseVI step (sparsevi.py:23-42, 69-76; model_linreg.py:4-10) cost
 * O(S D^2) each instead of a 2 N D S projection.
 *   bcx_project_moments        M_dev (C x ldm doubles, both triangles) = Z^T Z over the N rows of Z_dev (N x ldz, C <= 1024
 *                              columns used: the D features and the response); fp64 MFMA, fixed summation order;
 *                              work_dev: bcx_project_moments_scratch_bytes(N, C) bytes.  Row shards: sum the M of the shards.
 *   bcx_project_colsum_moments colsum_dev[s] = sum_n vecs[n][s] (centred over s, as bcx_project_colsum returns it) for the
 *                              data whose moments are M_dev (features in [0, D), response at ycol): thetabar, then
 *                              Delta G on the fp64 matrix cores; work_dev: bcx_project_colsum_moments_scratch_bytes(D, S)
 *                              bytes, ZERO before the first call (every call leaves them zero: a zero word is a partial
 *                              sum that has not been written yet). */
int64_t bcx_project_moments_scratch_bytes(int64_t N, int32_t C);
int64_t bcx_project_colsum_moments_scratch_bytes(int32_t D, int32_t S);
int bcx_project_moments(void* stream, const void* Z_dev, int64_t N, int64_t ldz, int32_t C, void* M_dev, int64_t ldm,
                        void* work_dev, int64_t work_bytes);
int bcx_project_colsum_moments(void* stream, const void* M_dev, int64_t ldm, int32_t D, int32_t ycol,
                               const void* theta_dev, int32_t S, int32_t ldt, double sigsq, void* colsum_dev, void* work_dev);
/* ... expanded around a point the caller supplies: tbar_dev = D doubles near the draws (the expansion is exact around any
 * point; bcx_linreg_posterior_draw leaves the mean of its draws), NULL = bcx_project_colsum_moments. */
int bcx_project_colsum_moments_at(void* stream, const void* M_dev, int64_t ldm, int32_t D, int32_t ycol,
                                  const void* theta_dev, int32_t S, int32_t ldt, double sigsq, void* colsum_dev, void* work_dev,
                                  const void* tbar_dev);
/* The two projections SparseVI's weight optimisation needs at every ADAM step (sparsevi.py:35-41 at fresh draws), for the
 * linear-regression family with the data's column sums in closed form: bcx_project_colsum_moments_at (M_dev, ldm, ycol_m,
 * colsum_dev, work_dev, tbar_dev: as there, tbar_dev given) and bcx_project_write_points of the Nc coreset points Zc_dev with
 * center = 0 (out_dev: Nc x ldo raw log-likelihoods) -- as ONE launch when the points take the 32 x 32-block kernel (both
 * only read the draws), else as the two calls.  Same results bit for bit either way. */
int bcx_project_points_colsum_moments(void* stream, const void* Zc_dev, int64_t Nc, int64_t ldzc, int32_t D, int32_t ycol,
                                      const void* theta_dev, int32_t S, int32_t ldt, double sigsq, void* out_dev, int64_t ldo,
                                      const void* M_dev, int64_t ldm, int32_t ycol_m, void* colsum_dev, void* work_dev,
                                      const void* tbar_dev);
/* SparseVI's weight optimisation with the weights resident on the device (sparsevi.py:69-76 -> util/opt.py:4-28): the
 * reference's loop body is projector.update(w, pts) [sampler call, sparsevi.py:25], two projections [sparsevi.py:35-41],
 * the gradient [sparsevi.py:72-74] and one projected-ADAM update [opt.py:19-25]; with these two entry points and the
 * projection calls above a host enqueues opt_itrs such steps and reads the k weights back once (csrc/svi.hip).
 *   bcx_linreg_posterior_draw  theta_dev (S x ld) = S draws from the posterior of the weighted coreset under the Gaussian
 *                              linear-regression model with a N(mu0, Sig0) prior (the sampler of the reference's
 *                              examples/linear_regression/main.py:124-147), theta = mu_w + R_dev Uw^T for the standard-normal
 *                              R_dev (S x ld, 16-byte aligned), as a rank-k correction of the prior's factor Sig0 = U0 U0^T
 *                              through a k x k Cholesky; tbar_dev (D) = the mean of the draws, formed as the "draw" of
 *                              Rbar_dev (ld doubles: the column means of R_dev, supplied by the caller).  Per-point inputs (formed
 *                              by the caller when the points change): K0 = (X U0)(X U0)^T (k x k), xmu0 = X mu0, y (k),
 *                              XU0 = X U0 and XS0 = X Sig0 (k x ld); per-model: U0T = U0^T (D x ld), mu0 (D).  w_dev: the
 *                              k weights (negative entries count as 0).  k <= 64, ld even.
 *   bcx_sparsevi_adam_step     resid = scaling colsum - w corevecs, g = -corevecs resid / S, then opt.py:19-25 on w_dev /
 *                              mom1_dev / mom2_dev (k doubles each) with every weight clamped at 0 (nn_idcs = None).
 *                              core_dev: k x ldc projected coreset points (core_is_raw != 0: as bcx_project_write_raw leaves
 *                              them -- the row means of projector.py:21 are then taken here); sched_dev: 3 doubles per step -- step_sched(i),
 *                              1 - b1^(i+1), 1 - b2^(i+1), evaluated by the host; step: i; trace_dev: NULL or steps x k
 *                              doubles receiving the weights after each step.
 * Asynchronous on `stream`; errors: bcx_project_last_error(). */
int bcx_linreg_posterior_draw(void* stream, int32_t k, int32_t D, int32_t ld, const void* w_dev, const void* K0_dev,
                              const void* xmu0_dev, const void* y_dev, const void* XU0_dev, const void* XS0_dev,
                              const void* U0T_dev, const void* mu0_dev, double sigsq, const void* R_dev, const void* Rbar_dev,
                              int32_t S, void* theta_dev, void* tbar_dev);
/*   bcx_linreg_posterior_apply the same draws for a loop of calls at the same points: G_dev (S x ld) = R U0^T and Gbar_dev (ld, its
 *                              column means) are formed once for all steps (bcx_linreg_posterior_draw with k = 0 and a zero
 *                              prior mean returns R U0^T), a step is then the rank-k correction theta = mu_w + G - (G X^T) B2
 *                              of rows read once.  X_dev: the points' features (k x ld, padding 0).  gscale_dev (ld doubles,
 *                              or NULL): the prior's factor is diag(gscale) (an isotropic / diagonal prior) and G_dev / Gbar_dev
 *                              hold R and its column means themselves -- the scaling is applied as the rows are read, R U0^T
 *                              is never stored.  Serves the (k, ld) for which bcx_linreg_posterior_apply_ok is non-zero
 *                              (k <= 32 and 16 k ld <= 128 KiB of LDS). */
int bcx_linreg_posterior_apply_ok(int32_t k, int32_t ld);
int bcx_linreg_posterior_apply(void* stream, int32_t k, int32_t D, int32_t ld, const void* w_dev, const void* K0_dev,
                               const void* xmu0_dev, const void* y_dev, const void* X_dev, const void* XS0_dev,
                               const void* mu0_dev, double sigsq, const void* G_dev, const void* Gbar_dev, int32_t S,
                               void* theta_dev, void* tbar_dev, const void* gscale_dev);
int bcx_sparsevi_adam_step(void* stream, int32_t k, int32_t S, const void* colsum_dev, double scaling, const void* core_dev,
                           int64_t ldc, void* w_dev, void* mom1_dev, void* mom2_dev, const void* sched_dev, int32_t step,
                           double b1, double b2, double eps, void* trace_dev, int32_t core_is_raw);
/* The sampler's raw material without the framework: bcx_standard_normal fills out_dev with `count` standard normal doubles
 * (counter-based Philox-4x32-10 + Box-Muller: the numbers of pair counters offset .. offset + ceil(count / 2) - 1 under the key
 * `seed`, the same whatever the launch shape; a caller that advances `offset` by ceil(count / 2) per call draws one
 * reproducible stream -- the role of np.random.randn in examples/linear_regression/main.py:147); bcx_column_means writes the
 * column means of `nblocks` blocks of n x ld doubles (block_stride doubles apart) to out_dev[b * out_stride + c] -- the means
 * of every ADAM step's normal numbers in one launch (their image under the posterior's factor is the mean of the draws). */
int bcx_standard_normal(void* stream, uint64_t seed, uint64_t offset, int64_t count, void* out_dev);
int bcx_column_means(void* stream, const void* rows_dev, int32_t nblocks, int32_t n, int32_t ld, int64_t block_stride,
                     void* out_dev, int64_t out_stride);
/* The same ADAM step for up to 4096 weights: up to 16 the single-workgroup kernel above, beyond it two launches of one
 * workgroup per slab of 8 weights (row means + the slabs' shares of w.dot(corevecs); resid, gradient, moments, step), which
 * need work_dev = bcx_sparsevi_adam_scratch_bytes(k, S) bytes of scratch (0 for k <= 16; -1: k or S out of range). */
int64_t bcx_sparsevi_adam_scratch_bytes(int32_t k, int32_t S);
int bcx_sparsevi_adam_step_ws(void* stream, int32_t k, int32_t S, const void* colsum_dev, double scaling, const void* core_dev,
                              int64_t ldc, void* w_dev, void* mom1_dev, void* mom2_dev, const void* sched_dev, int32_t step,
                              double b1, double b2, double eps, void* trace_dev, int32_t core_is_raw, void* work_dev,
                              int64_t work_bytes);
/* The weighted conjugate posterior of Gaussian linear regression for ANY number of weighted points, in the reference's own
 * arithmetic (examples/common/model_linreg.py:26-41 weighted_post returns mup, USigp, LSigpInv; the sampler of
 * examples/linear_regression/main.py:141-147 draws muw + randn . USigw^T):
 *     P = S0inv + X^T diag(max(w, 0)) X / sigsq = L L^T,   U_dev (D x ldu doubles, row-major, UPPER triangular) = USigp = L^-T,
 *     u_dev (D) = L^-1 (rhs0 + X^T (w y) / sigsq),  rhs0 = Sig0^-1 mu0,   mu_dev (D, may be NULL) = mup = U u,
 * from k weights w_dev that live on the device (SparseVI's ADAM loop updates them there), the points' features BY points
 * XT_dev (D x ldx: row a holds feature a of the k points; ldx >= k rounded up to 32, the padding zero, 16-byte aligned), their
 * responses y_dev and the prior precision S0inv_dev (D x lds0).  D <= 1024, k <= 4096; U_dev's lower triangle is never
 * written (zero it once), ldu even; work_dev: bcx_linreg_posterior_factor_scratch_bytes(D) bytes, 16-byte aligned, not
 * shared between streams.  Asynchronous on `stream`: P on the fp64 matrix cores, then ONE cooperative launch of <= 63
 * workgroups (blocked Cholesky whose row operations also produce L^-T and L^-1 rhs: csrc/lrpost.hip), then -- only when
 * mu_dev is given -- the product U u; the cooperative launch's workgroups hand tiles to each other and give up a wait after
 * 2 s: bcx_linreg_posterior_factor_status synchronises the stream and returns BCX_ERR_TIMEOUT for that, BCX_ERR_STATE if a
 * pivot was not positive, BCX_OK otherwise.
 * bcx_linreg_posterior_draw_factored: theta_dev (S x ld) = mup + R U^T = (R + 1 u^T) U^T for standard-normal R_dev (S x ld,
 * 16-byte aligned rows), tbar_dev (D) = mup + Rbar U^T for their column means Rbar_dev (ld) -- the mean of the draws. */
int64_t bcx_linreg_posterior_factor_scratch_bytes(int32_t D);
int bcx_linreg_posterior_factor(void* stream, int32_t k, int32_t D, int32_t ldx, const void* w_dev, const void* XT_dev,
                                const void* y_dev, const void* S0inv_dev, int32_t lds0, const void* rhs0_dev, double sigsq,
                                void* work_dev, int64_t work_bytes, void* U_dev, int64_t ldu, void* u_dev, void* mu_dev);
int bcx_linreg_posterior_factor_status(void* stream, int32_t D, const void* work_dev);
int bcx_linreg_posterior_draw_factored(void* stream, int32_t D, int32_t ld, const void* U_dev, int64_t ldu, const void* u_dev,
                                       const void* R_dev, const void* Rbar_dev, int32_t S, void* theta_dev, void* tbar_dev);
/* The sampler of the reference's logistic / Poisson regression experiment (examples/logistic_poisson_regression/main.py:15-41
 * get_laplace, :155-162 sampler_w) from k weights that live on the device: the Laplace approximation of the weighted posterior
 * of the points pts_dev (k x ldp; logistic: rows y x of D values; Poisson: rows [x, y]) under a standard-normal prior --
 * mu_dev (D) = the mode (damped Newton to |step| < tol, at most max_iter steps; warm != 0: started from the contents of mu_dev,
 * e.g. the previous ADAM step's mode), and the draws theta_dev (S x ld) = mu + R W, tbar_dev (D) = mu + Rbar W with W = L^-1 for
 * the Cholesky factor L of the negative Hessian at the mode (Sigma = W^T W), R_dev (S x ld) standard normal numbers and Rbar_dev
 * their column means.  family 0: logistic, 1: Poisson (softplus rate).  ONE launch of one workgroup (csrc/laplace.hip), the
 * points resident in LDS: serves the (k, D) for which bcx_laplace_sampler_ok is non-zero (D <= 32, the points + four doubles
 * each within 96 KiB).  status_dev (2 int32): [0] 0 converged / 1 iteration limit / 2 no positive definite Newton matrix,
 * [1] Newton steps taken.  Asynchronous on `stream`. */
int bcx_laplace_sampler_ok(int32_t k, int32_t D);
int64_t bcx_laplace_sampler_lds_bytes(int32_t k, int32_t D);
int bcx_laplace_sampler(void* stream, int32_t family, int32_t k, int32_t D, const void* w_dev, const void* pts_dev, int64_t ldp,
                        void* mu_dev, int32_t warm, double tol, int32_t max_iter, const void* R_dev, const void* Rbar_dev,
                        int32_t S, int32_t ld, void* theta_dev, void* tbar_dev, void* status_dev);
/* The dense re-weight's Gram matrix as an operator of its own (optimize() forms it over the active rows, snnls.py:82-97:
 * `nnls(A[:, active], b)` solves the normal equations of that k-column block): G_dev (k x ldg doubles, both triangles) =
 * V V^T for the k rows of d doubles at rows_dev (row stride ld >= d), on the fp64 matrix cores; the d products of an entry
 * are summed in a fixed order for a given (k, d) on a given device (csrc/gram.hip: tiles of G cut into equal ranges of
 * 16-value stages over the resident workgroups, partial tiles added by the workgroup that finishes the tile; rows that are
 * not 16-byte aligned and k < 192: 64 x 64 blocks, slices of the row length added in slice order), G is symmetric bit for
 * bit.  Asynchronous on `stream`.  k <= BCX_GRAM_MAX_ROWS; work_dev: bcx_gram_scratch_bytes(k, d) bytes (up to one 128 x 128
 * tile of doubles per resident workgroup).  Errors: bcx_project_last_error(). */
#define BCX_GRAM_MAX_ROWS 16384
int64_t bcx_gram_scratch_bytes(int32_t k, int32_t d);
int bcx_gram(void* stream, const void* rows_dev, int32_t k, int32_t d, int64_t ld, void* G_dev, int64_t ldg,
             void* work_dev, int64_t work_bytes);
/* bcx_gram is asynchronous; the kernel that balances the tiles over the chip hands partial tiles between workgroups and
 * gives up a wait after 5 s (a GPU shared with another process, preemption).  bcx_gram_check synchronises `stream` and
 * returns BCX_ERR_TIMEOUT if a call of this process that used the scratch `work_dev` gave up -- that call's G is
 * not valid (zero the first 8 bytes of the scratch to re-arm); BCX_OK otherwise.  optimize() (bcx_optimize) makes the same
 * check on its own Gram launches and returns BCX_ERR_TIMEOUT. */
int bcx_gram_check(void* stream, const void* work_dev);
/* Two row passes of the constructor path behind a device projector (reference: projector.py:21, hilbert.py:19-22):
 * bcx_center_rows subtracts every row's mean in place (rows_dev: N x ld doubles, S used per row); bcx_row_sumsq writes every
 * row's sum of squares to out_dev (N doubles) -- the subsample branch drops the rows where it is zero. */
int bcx_center_rows(void* stream, void* rows_dev, int64_t N, int32_t S, int64_t ld);
int bcx_row_sumsq(void* stream, const void* rows_dev, int64_t N, int32_t S, int64_t ld, void* out_dev);
const char* bcx_project_last_error(void);
/* Measurement: hipEvents around the projection kernel alone, recorded on the stream the kernel is launched on.
 * bcx_project_profile(1) starts timing every later projection launch of the calling host thread, (0) stops;
 * bcx_project_profile_read returns the totals since the start: kernel milliseconds, launches, and the algorithmic
 * flops 2 N D S of those launches (the roofline figure of bench.py --config c5 / c3). */
int bcx_project_profile(int32_t on);
int bcx_project_profile_read(double* ms_total, int64_t* launches, double* flops);
/* Library/arch identification, e.g. "bcx 0.1 gfx950". */
const char* bcx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* BCX_H */
