// nnls.hip -- the dense re-weight step: non-negative least squares on the active columns.
//   OMP reweight : w[f] = 1; w[active] = nnls(A[:, active], b)        orthopursuit.py:37-42
//   optimize()   : w[active] = nnls(A[:, active], b), keep unless worse   snnls.py:82-97
// The reference calls scipy.optimize.nnls (Lawson-Hanson active set on the d x k matrix, cold start).
// Here the k active rows are replicated on every shard, their Gram matrix G = V V^T (k x k) and
// c = V b are kept on the device, and the same active-set iteration runs on the normal equations:
// the passive block's inverse H = inv(G[P,P]) is maintained by bordering (add a column) and
// rank-1 deletion (drop a column), every solve is followed by one step of iterative refinement
// against G.  NNLS solutions are unique for independent columns, so warm starting from the
// previous passive set gives the reference's result.  The Gram matrix of optimize() is a small
// GEMM and runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64); everything else is
// latency-bound single-workgroup work.
#include <stddef.h>
#include <stdlib.h>
#include <algorithm>
#include "bcx_internal.h"
#include "dev_util.h"
#include "apply_common.h"
#include "nnls_common.h"
#include "resolve_core.h"

#ifdef BCX_TIMING
#define NN_COUNT(st, i) do { if (threadIdx.x == 0) (st)->dbg_t[12 + (i)] += 1; } while (0)
#else
#define NN_COUNT(st, i) do {} while (0)
#endif

// y[a] = sum_b M[b*ld + a] * v[b], a < p  (M symmetric: reading column a as row entries is coalesced).
// 8 independent loads per step so the L2 latencies overlap; fixed summation order.
static __device__ void mv_sym(const double* M, int64_t ld, int p, const double* v, double* y) {
  for (int a = threadIdx.x; a < p; a += blockDim.x) {
    double acc = 0.0;
    int b = 0;
    for (; b + 8 <= p; b += 8) {
      double m[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) m[t] = M[(size_t)(b + t) * ld + a];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += m[t] * v[b + t];
    }
    for (; b < p; ++b) acc += M[(size_t)b * ld + a] * v[b];
    y[a] = acc;
  }
  __syncthreads();
}
// y[a] = sum_b G[plist[b]][plist[a]] * v[b]
static __device__ void mv_gram(const NnlsArgs& n, int p, const double* v, double* y) {
  for (int a = threadIdx.x; a < p; a += blockDim.x) {
    const int ca = n.plist[a];
    double acc = 0.0;
    int b = 0;
    for (; b + 8 <= p; b += 8) {
      double m[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) m[t] = n.gram[(size_t)n.plist[b + t] * n.ldg + ca];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += m[t] * v[b + t];
    }
    for (; b < p; ++b) acc += n.gram[(size_t)n.plist[b] * n.ldg + ca] * v[b];
    y[a] = acc;
  }
  __syncthreads();
}

// Add slot to the passive set: H' = inverse of the bordered matrix.  False when the column is
// numerically dependent on P (Schur complement not positive).
static __device__ bool border_add(const NnlsArgs& n, int slot, double* scratch) {
  DevState* st = n.a.st;
  const int p = st->np;
  const int64_t ld = n.ldg;
  for (int a = threadIdx.x; a < p; a += blockDim.x) n.t0[a] = n.gram[(size_t)slot * ld + n.plist[a]];
  __syncthreads();
  mv_sym(n.hinv, ld, p, n.t0, n.t1);   // u = H g
  double v[1] = {0.0};
  for (int a = threadIdx.x; a < p; a += blockDim.x) v[0] += n.t0[a] * n.t1[a];
  block_allsum<1>(v, scratch);
  const double gff = n.gram[(size_t)slot * ld + slot];
  const double s = gff - v[0];
  if (!(s > 1e-12 * gff)) return false;
  if (threadIdx.x == 0 && !(s > 1e-4 * gff)) st->omp_ill = 1;
  const double inv = 1.0 / s;
  for (int idx = threadIdx.x; idx < p * p; idx += blockDim.x) {
    const int r = idx / p, c = idx - r * p;
    n.hinv[(size_t)r * ld + c] += n.t1[r] * n.t1[c] * inv;
  }
  for (int a = threadIdx.x; a < p; a += blockDim.x) {
    const double e = -n.t1[a] * inv;
    n.hinv[(size_t)p * ld + a] = e;
    n.hinv[(size_t)a * ld + p] = e;
  }
  if (threadIdx.x == 0) {
    n.hinv[(size_t)p * ld + p] = inv;
    n.plist[p] = slot;
    n.ppos[slot] = p;
    st->np = p + 1;
  }
  __syncthreads();
  return true;
}

// Remove position q from the passive set (rank-1 downdate of H, then move the last position into q).
static __device__ void border_del(const NnlsArgs& n, int q) {
  DevState* st = n.a.st;
  const int p = st->np, last = p - 1;
  const int64_t ld = n.ldg;
  for (int a = threadIdx.x; a < p; a += blockDim.x) n.t0[a] = n.hinv[(size_t)a * ld + q];
  __syncthreads();
  const double hqq = n.t0[q];
  for (int idx = threadIdx.x; idx < p * p; idx += blockDim.x) {
    const int r = idx / p, c = idx - r * p;
    n.hinv[(size_t)r * ld + c] -= n.t0[r] * n.t0[c] / hqq;
  }
  __syncthreads();
  const int gone = n.plist[q];
  if (q != last) {
    for (int a = threadIdx.x; a < p; a += blockDim.x) n.t1[a] = n.hinv[(size_t)last * ld + a];
    __syncthreads();
    for (int a = threadIdx.x; a < last; a += blockDim.x) {
      if (a == q) continue;
      n.hinv[(size_t)q * ld + a] = n.t1[a];
      n.hinv[(size_t)a * ld + q] = n.t1[a];
    }
    if (threadIdx.x == 0) {
      n.hinv[(size_t)q * ld + q] = n.t1[last];
      const int moved = n.plist[last];
      n.plist[q] = moved;
      n.ppos[moved] = q;
    }
  }
  if (threadIdx.x == 0) { n.ppos[gone] = -1; st->np = last; }
  __syncthreads();
}

// z = argmin over the passive set: z = H c_P, then iterative refinement against the Gram matrix itself,
// z += H (c_P - G_PP z), until the residual is at rounding level (at most 4 steps; one suffices unless the
// system is ill-conditioned and the bordered inverse has drifted).
static __device__ void passive_solve(const NnlsArgs& n, double* scratch) {
  const int p = n.a.st->np;
  double cmax = 0.0;
  for (int a = threadIdx.x; a < p; a += blockDim.x) { const double c = n.cvec[n.plist[a]]; n.t0[a] = c; cmax = fmax(cmax, fabs(c)); }
  cmax = block_allmax(cmax, scratch);
  mv_sym(n.hinv, n.ldg, p, n.t0, n.z);
  // well-conditioned systems: one refinement step, no convergence test (3 passes in total);
  // ill-conditioned ones (omp_ill): refine until the residual is at rounding level
  const int max_it = n.a.st->omp_ill ? 4 : 1;
  for (int it = 0; it < max_it; ++it) {
    NN_COUNT(n.a.st, 2);
    mv_gram(n, p, n.z, n.t1);
    double rmax = 0.0;
    for (int a = threadIdx.x; a < p; a += blockDim.x) { const double r = n.t0[a] - n.t1[a]; n.t1[a] = r; rmax = fmax(rmax, fabs(r)); }
    if (max_it > 1) {
      rmax = block_allmax(rmax, scratch);
      if (!(rmax > 1e-14 * cmax)) break;
    } else {
      __syncthreads();
    }
    mv_sym(n.hinv, n.ldg, p, n.t1, n.t2);
    for (int a = threadIdx.x; a < p; a += blockDim.x) n.z[a] += n.t2[a];
    __syncthreads();
  }
}

// Lawson-Hanson active-set iteration over the slots flagged FLAG_INS, warm-started from the
// current passive set / x.  On return x[j] > 0 exactly for j in P, x[j] = 0 elsewhere.
static __device__ void nnls_run(const NnlsArgs& n, int k, double tolscale, double* scratch) {
  DevState* st = n.a.st;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int max_outer = 3 * k + 16;
  for (int outer = 0; outer < max_outer; ++outer) {
    // dual vector w = c - G x on the candidates (S \ P, not rejected): one wave per candidate
    const int p = st->np;
    for (int j = wave; j < k; j += nw) {
      const int fl = n.flag[j];
      if (!(fl & FLAG_INS) || (fl & FLAG_REJ) || n.ppos[j] >= 0) continue;
      double acc = 0.0;
      for (int a = lane; a < p; a += 64) {
        const int ca = n.plist[a];
        acc += n.gram[(size_t)j * n.ldg + ca] * n.x[ca];
      }
      acc = wave_allsum(acc);
      if (lane == 0) n.t2[j] = n.cvec[j] - acc;
    }
    __syncthreads();
    double bv = -INFINITY; int bi = -1;
    for (int j = threadIdx.x; j < k; j += blockDim.x) {
      const int fl = n.flag[j];
      if (!(fl & FLAG_INS) || (fl & FLAG_REJ) || n.ppos[j] >= 0) continue;
      const double wv = n.t2[j];
      if (wv > tolscale * n.a.act_norm[j] && (bi < 0 || wv > bv)) { bv = wv; bi = j; }
    }
    const ArgBest best = block_argbest(bv, bi, scratch);
    if (best.i < 0) break;
    NN_COUNT(st, 0);
    if (!border_add(n, best.i, scratch)) {
      if (threadIdx.x == 0) n.flag[best.i] |= FLAG_REJ;
      __syncthreads();
      continue;
    }
    if (threadIdx.x == 0) n.x[best.i] = 0.0;
    __syncthreads();
    for (int inner = 0; inner < max_outer; ++inner) {
      NN_COUNT(st, 1);
      passive_solve(n, scratch);
      const int pp = st->np;
      // feasibility: all z > 0 ?  else step length alpha = min x/(x - z) over z <= 0
      double amin = INFINITY; int apos = -1;
      for (int a = threadIdx.x; a < pp; a += blockDim.x) {
        const double za = n.z[a];
        if (!(za > 0.0)) {
          const double xa = n.x[n.plist[a]];
          double al = xa / (xa - za);
          if (!(al == al)) al = 0.0;                 // 0/0: a zero weight asked to go further down
          if (apos < 0 || al < amin) { amin = al; apos = a; }
        }
      }
      const ArgBest worst = block_argbest(-amin, apos, scratch);   // smallest alpha, lowest position
      if (worst.i < 0) {
        for (int a = threadIdx.x; a < pp; a += blockDim.x) n.x[n.plist[a]] = n.z[a];
        __syncthreads();
        break;
      }
      const double alpha = -worst.v;
      for (int a = threadIdx.x; a < pp; a += blockDim.x) {
        const int c = n.plist[a];
        const double xa = n.x[c];
        const double xn = xa + alpha * (n.z[a] - xa);
        const bool rm = (a == worst.i) || !(xn > 0.0);
        n.x[c] = rm ? 0.0 : xn;
        if (rm) n.flag[c] |= FLAG_RM;
      }
      __syncthreads();
      // drop flagged positions, highest position first (the element moved into a hole was already checked)
      for (;;) {
        const int pq = st->np;
        int cand = -1;
        for (int a = threadIdx.x; a < pq; a += blockDim.x)
          if (n.flag[n.plist[a]] & FLAG_RM) cand = a > cand ? a : cand;
        const ArgBest top = block_argbest((double)cand, cand, scratch);
        if (top.i < 0) break;
        const int slot = n.plist[top.i];
        if (threadIdx.x == 0) {
          n.flag[slot] &= ~FLAG_RM;
          if (slot == best.i && inner == 0) n.flag[slot] |= FLAG_REJ;   // LH safeguard: do not re-pick at once
        }
        __syncthreads();
        NN_COUNT(st, 3);
        border_del(n, top.i);
      }
      if (st->np == 0) break;
    }
  }
}

// xw = sum_{j in P} x_j * row_j  -> out[0..d), using S column segments of the slot range
static __device__ void passive_combination(const NnlsArgs& n, double* out, double* part) {
  const ApplyArgs& a = n.a;
  const int d = a.d, p = a.st->np;
  int S = blockDim.x / ((d + 63) / 64 * 64);
  if (S < 1) S = 1;
  if (S > 16) S = 16;
  const int cols = (d + 63) / 64 * 64;
  for (int t = threadIdx.x; t < S * cols; t += blockDim.x) {
    const int seg = t / cols, j = t - seg * cols;
    if (j >= d) continue;
    double acc = 0.0;
    for (int q = seg; q < p; q += S) {
      const int c = n.plist[q];
      acc += n.x[c] * a.act_rows[(size_t)c * d + j];
    }
    part[(size_t)seg * d + j] = acc;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    double acc = part[j];
    for (int s = 1; s < S; ++s) acc += part[(size_t)s * d + j];
    out[j] = acc;
  }
  __syncthreads();
}

// Rebuild H = inv(G[P,P]) for P = {slots with weight > 0} by successive bordering.
static __device__ void rebuild_passive(const NnlsArgs& n, int k, double* scratch) {
  DevState* st = n.a.st;
  if (threadIdx.x == 0) st->np = 0;
  for (int j = threadIdx.x; j < k; j += blockDim.x) { n.ppos[j] = -1; n.x[j] = n.a.act_w[j] > 0.0 ? n.a.act_w[j] : 0.0; }
  __syncthreads();
  for (int j = 0; j < k; ++j) {
    if (!(n.a.act_w[j] > 0.0)) continue;
    (void)border_add(n, j, scratch);
  }
  if (threadIdx.x == 0) st->hvalid = 1;
  __syncthreads();
}

// ---- OMP apply --------------------------------------------------------------------------------
// Typical step: one new column joins the support and every weight stays positive.  Then the NNLS
// solution on S u {f} follows from the bordered inverse in closed form (u = H g, s = G_ff - g.u,
// t = (c_f - g.x)/s, x <- x - t u, x_f = t): two passes over H instead of a full active-set solve.
// Anything else (a weight would turn non-positive, dependent column, stale inverse, periodic
// re-solve) goes through nnls_run, which gives the same unique solution.
#define OMP_RESOLVE_EVERY 32

// ---- OMP apply, multi-kernel form ----------------------------------------------------------------
// The step split into phases that use many CUs, chained by ordinary kernel boundaries (decisions travel in
// DevState).  Fallback of the incremental step (omp_lh.hip) for active sets beyond its LDS budget:
//   rows    (grid)  -An[j].r for the active rows and row_j . xf for all slots (+ xf.xf, xf.b)
//   decide  (1 WG)  f, slot, new-slot data, g = G[slot, P]; chooses DONE / FAST_TRY / GENERAL
//   matvec  (grid)  u = H g
//   step    (1 WG)  Schur complement, step t, positivity -> FAST_ACCEPT (x updated) / DONE / GENERAL
//   general (1 WG)  full active-set solve (rare)
//   combine (grid)  xw' = sum_P x_j row_j; extra workgroups apply the bordered update of H
//   finish  (1 WG)  error, monotone check, commit / revert, trace, next query
__global__ __launch_bounds__(256) void omp_rows_kernel(NnlsArgs n) {
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  if (!st->active) return;
  __shared__ int s_win, s_ovf;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, d = a.d;
  if (threadIdx.x == 0) { int o; s_win = omp_pick_record(a, &o); s_ovf = o; }
  __syncthreads();
  if (s_ovf || s_win < 0) return;
  const int k = st->k;
  const int j = blockIdx.x * 4 + wave;
  if (j > k) return;
  const double* xf = a.recs + (size_t)s_win * (d + BCX_REC_HDR) + BCX_REC_HDR;
  const double* row = (j < k) ? a.act_rows + (size_t)j * d : xf;
  const double* other = (j < k) ? a.q64 : a.b;
  double a0 = 0.0, a1 = 0.0;
  for (int i = lane; i < d; i += 64) {
    const double rv = row[i];
    a0 += rv * xf[i];
    a1 += rv * other[i];
  }
  a0 = wave_allsum(a0);
  a1 = wave_allsum(a1);
  if (lane == 0) {
    if (j < k) { n.t3[j] = a0; n.t2[j] = -(a1 / a.act_norm[j]); }
    else { st->omp_gff = a0; st->omp_cf = a1; }
  }
}

__global__ __launch_bounds__(NN_THREADS) void omp_decide_kernel(NnlsArgs n) {
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  if (!st->active) return;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ int s_win, s_ovf, s_slot, s_npos;
  __shared__ unsigned long long s_minidx;
  const int tid = threadIdx.x, d = a.d;
  if (tid == 0) {
    int o; s_win = omp_pick_record(a, &o); s_ovf = o;
    s_slot = 0x7fffffff; s_npos = 0; s_minidx = 0x7fffffffffffffffULL;
    st->omp_mode = OMP_IDLE;
  }
  __syncthreads();
  if (s_ovf) { if (tid == 0) { st->active = 0; st->halt = HALT_NEED_EXACT; } return; }
  if (s_win < 0) { if (tid == 0) { st->active = 0; st->halt = HALT_DONE; } return; }
  const double* rec = a.recs + (size_t)s_win * (d + BCX_REC_HDR);
  const double* xf = rec + BCX_REC_HDR;
  const int k = st->k;
  int npos = 0;
  for (int s = tid; s < k; s += blockDim.x) if (a.act_w[s] > 0.0) ++npos;
  if (npos) atomicAdd(&s_npos, npos);
  __syncthreads();
  const bool checked = s_npos > 0;
  int64_t f = (int64_t)rec[1];
  const double nf = rec[2];
  if (checked) {
    double bv = -INFINITY; int bi = -1; int64_t bidx = 0;
    for (int j = tid; j < k; j += blockDim.x) {
      if (!(a.act_w[j] > 0.0)) continue;
      const double vv = n.t2[j];
      if (bi < 0 || vv > bv || (vv == bv && a.act_idx[j] < bidx)) { bv = vv; bi = j; bidx = a.act_idx[j]; }
    }
    const double vmax = block_allmax(bi >= 0 ? bv : -INFINITY, scratch);
    if (bi >= 0 && bv == vmax) atomicMin(&s_minidx, (unsigned long long)bidx);
    __syncthreads();
    if (!(rec[0] >= vmax)) f = (int64_t)s_minidx;          // orthopursuit.py:32-35
  }
  for (int s = tid; s < k; s += blockDim.x) if (a.act_idx[s] == f) atomicMin(&s_slot, s);
  __syncthreads();
  int slot = s_slot == 0x7fffffff ? -1 : s_slot;
  const bool fresh = slot < 0;
  if (fresh) slot = k;
  for (int j = tid; j < k; j += blockDim.x) n.wbak[j] = a.act_w[j];
  if (fresh) {
    for (int i = tid; i < d; i += blockDim.x) a.act_rows[(size_t)slot * d + i] = xf[i];
    for (int j = tid; j < k; j += blockDim.x) {
      const double g = n.t3[j];
      n.gram[(size_t)slot * n.ldg + j] = g;
      n.gram[(size_t)j * n.ldg + slot] = g;
    }
    if (tid == 0) {
      a.act_idx[slot] = f; a.act_norm[slot] = nf; a.act_w[slot] = 0.0; n.ppos[slot] = -1; n.x[slot] = 0.0;
      n.gram[(size_t)slot * n.ldg + slot] = st->omp_gff;
      n.cvec[slot] = st->omp_cf;
    }
  }
  __syncthreads();
  const int p = st->np;
  int mode;
  if (!st->hvalid) mode = OMP_GENERAL;
  else if (n.ppos[slot] >= 0) mode = OMP_DONE;
  else if (st->omp_ill) mode = OMP_GENERAL;             // f already carries weight: nothing changes
  else if ((st->since_refresh % OMP_RESOLVE_EVERY) == OMP_RESOLVE_EVERY - 1) mode = OMP_GENERAL;
  else {
    mode = OMP_FAST_TRY;
    for (int q = tid; q < p; q += blockDim.x) n.t0[q] = n.gram[(size_t)slot * n.ldg + n.plist[q]];
  }
  if (tid == 0) {
    st->omp_mode = mode; st->omp_slot = slot; st->omp_fresh = fresh; st->omp_checked = checked;
    st->omp_f = f; st->omp_nf = nf; st->omp_p = p;
  }
}

// u[a] = sum_b H[b][a] g[b]: one workgroup per 64 columns, the four waves take b = w, w+4, ...
__global__ __launch_bounds__(256) void omp_mv_kernel(NnlsArgs n) {
  DevState* st = n.a.st;
  if (!st->active || st->omp_mode != OMP_FAST_TRY) return;
  __shared__ double seg[4][64];
  const int p = st->omp_p;
  const int a0 = blockIdx.x * 64;
  if (a0 >= p) return;
  const int col = a0 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  double acc = 0.0;
  if (col < p) {
    int b = w;
    for (; b + 28 < p; b += 32) {
      double m[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) m[t] = n.hinv[(size_t)(b + 4 * t) * n.ldg + col];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += m[t] * n.t0[b + 4 * t];
    }
    for (; b < p; b += 4) acc += n.hinv[(size_t)b * n.ldg + col] * n.t0[b];
  }
  seg[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && col < p) n.t1[col] = ((seg[0][threadIdx.x] + seg[1][threadIdx.x]) + seg[2][threadIdx.x]) + seg[3][threadIdx.x];
}

__global__ __launch_bounds__(NN_THREADS) void omp_step_kernel(NnlsArgs n) {
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  if (!st->active || st->omp_mode != OMP_FAST_TRY) return;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ int s_bad;
  const int tid = threadIdx.x, d = a.d;
  const int p = st->omp_p, slot = st->omp_slot;
  const int k1 = st->k + (st->omp_fresh ? 1 : 0);
  if (tid == 0) s_bad = 0;
  double r[2] = {0.0, 0.0};
  for (int q = tid; q < p; q += blockDim.x) { r[0] += n.t0[q] * n.t1[q]; r[1] += n.t0[q] * n.x[n.plist[q]]; }
  block_allsum<2>(r, scratch);
  const double eps = 2.220446049250313e-16;
  const double tolscale = 10.0 * eps * (double)(d > k1 ? d : k1) * st->bnorm;
  const double gff = n.gram[(size_t)slot * n.ldg + slot];
  const double sc = gff - r[0];
  const double wvf = n.cvec[slot] - r[1];
  int mode = OMP_GENERAL;
  if (!(wvf > tolscale * a.act_norm[slot])) {
    mode = OMP_DONE;                                        // dual not positive: f gets weight 0
  } else if (!(sc > 1e-4 * gff)) {
    if (tid == 0) st->omp_ill = 1;                        // nearly dependent column: refined general solve
  } else {
    const double t = wvf / sc;
    for (int q = tid; q < p; q += blockDim.x)
      if (!(n.x[n.plist[q]] - t * n.t1[q] > 0.0)) s_bad = 1;
    __syncthreads();
    if (!s_bad && t > 0.0) {
      for (int q = tid; q < p; q += blockDim.x) n.x[n.plist[q]] -= t * n.t1[q];
      if (tid == 0) {
        n.plist[p] = slot; n.ppos[slot] = p; n.x[slot] = t;
        st->np = p + 1;
        st->omp_t = t; st->omp_inv = 1.0 / sc;
      }
      mode = OMP_FAST_ACCEPT;
    }
  }
  __syncthreads();
  if (tid == 0) st->omp_mode = mode;
}

// H <- [[H + u u^T / s, -u/s], [-u^T/s, 1/s]]   (p = size before the insertion); runs inside the
// combine launch on the workgroups beyond the column blocks
static __device__ void omp_rank1_part(const NnlsArgs& n, int wg, int nwg) {
  DevState* st = n.a.st;
  if (st->omp_mode != OMP_FAST_ACCEPT) return;
  const int p = st->omp_p;
  const double inv = st->omp_inv;
  const int64_t total = (int64_t)p * p;
  for (int64_t idx = (int64_t)wg * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)nwg * blockDim.x) {
    const int rr = (int)(idx / p), cc = (int)(idx - (int64_t)rr * p);
    n.hinv[(size_t)rr * n.ldg + cc] += n.t1[rr] * n.t1[cc] * inv;
  }
  for (int q = wg * blockDim.x + threadIdx.x; q < p; q += nwg * blockDim.x) {
    const double e = -n.t1[q] * inv;
    n.hinv[(size_t)p * n.ldg + q] = e;
    n.hinv[(size_t)q * n.ldg + p] = e;
  }
  if (wg == 0 && threadIdx.x == 0) n.hinv[(size_t)p * n.ldg + p] = inv;
}

__global__ __launch_bounds__(NN_THREADS) void omp_general_kernel(NnlsArgs n) {
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  if (!st->active || st->omp_mode != OMP_GENERAL) return;
  __shared__ double scratch[BCX_SCRATCH];
  const int tid = threadIdx.x, d = a.d;
  const int k = st->k, slot = st->omp_slot;
  const int k1 = k + (st->omp_fresh ? 1 : 0);
  if (!st->hvalid) rebuild_passive(n, k, scratch);
  for (int j = tid; j < k1; j += blockDim.x) {
    const bool in = (j == slot) || (a.act_w[j] > 0.0);
    n.flag[j] = in ? FLAG_INS : 0;
    if (n.ppos[j] < 0) n.x[j] = 0.0;
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16;
  const double tolscale = 10.0 * eps * (double)(d > k1 ? d : k1) * st->bnorm;
  nnls_run(n, k1, tolscale, scratch);
}

// tmp[j] = sum_{q < np} x[plist[q]] * rows[plist[q]][j]: one workgroup per 64 columns
__global__ __launch_bounds__(256) void omp_combine_kernel(NnlsArgs n, int ncol_wg) {
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  if (!st->active) return;
  if ((int)blockIdx.x >= ncol_wg) { omp_rank1_part(n, blockIdx.x - ncol_wg, gridDim.x - ncol_wg); return; }
  __shared__ double seg[4][64];
  const int d = a.d, p = st->np;
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), w = threadIdx.x >> 6;
  double acc = 0.0;
  if (col < d) {
    int q = w;
    for (; q + 28 < p; q += 32) {
      double m[8], xv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) { const int c = n.plist[q + 4 * t]; xv[t] = n.x[c]; m[t] = a.act_rows[(size_t)c * d + col]; }
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += xv[t] * m[t];
    }
    for (; q < p; q += 4) { const int c = n.plist[q]; acc += n.x[c] * a.act_rows[(size_t)c * d + col]; }
  }
  seg[w][threadIdx.x & 63] = acc;
  __syncthreads();
  if (w == 0 && col < d) a.tmp[col] = ((seg[0][threadIdx.x] + seg[1][threadIdx.x]) + seg[2][threadIdx.x]) + seg[3][threadIdx.x];
}

__global__ __launch_bounds__(BCX_APPLY_THREADS) void omp_finish_kernel(NnlsArgs n) {
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  if (!st->active) return;
  __shared__ double scratch[BCX_SCRATCH];
  const int tid = threadIdx.x, d = a.d;
  const int k = st->k;
  const int k1 = k + (st->omp_fresh ? 1 : 0);
  const bool checked = st->omp_checked != 0;
  const double err0 = st->err;
  double v[2] = {0.0, 0.0};
  for (int j = tid; j < d; j += blockDim.x) {
    const double x = a.tmp[j], rr = x - a.b[j];
    v[0] += rr * rr; v[1] += x * x;
  }
  block_allsum<2>(v, scratch);
  const double new_err = sqrt(v[0]);
  int status = BCX_IT_OK;
  if (checked && !st->no_monotone && new_err > err0) status = BCX_IT_FAIL_MONOTONE;      // snnls.py:56-58
  if (status == BCX_IT_OK) {
    for (int j = tid; j < k1; j += blockDim.x) a.act_w[j] = (n.ppos[j] >= 0) ? n.x[j] : 0.0;
    for (int j = tid; j < d; j += blockDim.x) a.xw[j] = a.tmp[j];
    if (tid == 0) {
      st->k = k1;
      st->err = new_err;
      const double nwn = sqrt(v[1]);
      st->nw = nwn == 0.0 ? 1.0 : nwn;
      st->since_refresh += 1;
      if (checked && !st->no_monotone) st->retried = 0;
    }
  } else {
    for (int j = tid; j < k; j += blockDim.x) a.act_w[j] = n.wbak[j];
    if (tid == 0) st->hvalid = 0;
  }
  __syncthreads();
  if (tid == 0) {
    const int64_t it = st->it;
    a.tr_sel[it] = st->omp_f; a.tr_err[it] = st->err; a.tr_status[it] = status;
    st->it = it + 1;
    st->exact_mode = 0;
    st->omp_mode = OMP_IDLE;
    st->hlo_valid = 0;
    if (status != BCX_IT_OK) {
      if (st->retried) { st->limit = 1; st->active = 0; st->halt = HALT_LIMIT; }
      else st->retried = 1;
    }
  }
  __syncthreads();
  if (!st->active) return;
  prepare_next(a, scratch);
}

void fill_nnls_args(bcx_solver* s, NnlsArgs& n, const double* recs) {
  fill_apply_args(s, n.a, recs);
  n.a.refresh_every = 0;   // OMP recomputes xw from the passive set on every step
  n.gram = s->gram; n.hinv = s->hinv; n.hlo = s->hinv_lo; n.ldg = s->gram_cap;
  n.cvec = s->cvec; n.plist = s->plist; n.ppos = s->ppos;
  n.x = s->nn_x; n.z = s->nn_z;
  n.t0 = s->nn_tmp; n.t1 = s->nn_tmp + s->gram_cap; n.t2 = s->nn_tmp + 2 * s->gram_cap;
  n.t3 = s->nn_tmp + 3 * s->gram_cap;
  n.flag = s->nn_flag; n.wbak = s->nn_wbak;
  n.xr = s->nn_xr;
}

int bcx_launch_apply_omp(bcx_solver* s, const double* recv_dev) {
  NnlsArgs n;
  fill_nnls_args(s, n, recv_dev);
  s->k_ub += 1;                               // this step may add one slot
  const int64_t kub = s->k_ub;
  // default: the incremental multi-workgroup step of omp_lh.hip; beyond its LDS budget (or with BCX_OMP_MULTI=1 /
  // BCX_OMP_FORM=multi, dev) the multi-kernel form below (plain-double inverse)
  static const char* form = bcx_dev_env("BCX_OMP_FORM");
  static const bool legacy = bcx_dev_env("BCX_OMP_MULTI") != nullptr || (form && form[0] == 'm');
  if (!legacy && s->grid_counter) {
    const int rc = bcx_launch_omp_lh(s, n, nullptr);
    if (rc <= 0) return rc;
  }
  const int d = s->cfg.d;
  hipLaunchKernelGGL(omp_rows_kernel, dim3((unsigned)((kub + 1 + 3) / 4)), dim3(256), 0, s->stream, n);
  hipLaunchKernelGGL(omp_decide_kernel, dim3(1), dim3(NN_THREADS), 0, s->stream, n);
  hipLaunchKernelGGL(omp_mv_kernel, dim3((unsigned)((kub + 63) / 64)), dim3(256), 0, s->stream, n);
  hipLaunchKernelGGL(omp_step_kernel, dim3(1), dim3(NN_THREADS), 0, s->stream, n);
  hipLaunchKernelGGL(omp_general_kernel, dim3(1), dim3(NN_THREADS), 0, s->stream, n);
  const int ncol = (d + 63) / 64;
  const int64_t r1 = std::max<int64_t>(1, std::min<int64_t>((kub * kub + 2047) / 2048, 512));
  hipLaunchKernelGGL(omp_combine_kernel, dim3((unsigned)(ncol + r1)), dim3(256), 0, s->stream, n, ncol);
  hipLaunchKernelGGL(omp_finish_kernel, dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, n);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

// Single shard: the whole tail of an OMP iteration (resolve of the scan's partials + negative direction + incremental
// Lawson-Hanson step) as ONE launch of omp_lh_kernel.  1 = not applicable (its LDS budget, or the dev knobs that select the
// multi-kernel form): the caller launches resolve_kernel and bcx_launch_apply instead.
int bcx_launch_omp_fused(bcx_solver* s, int exact) {
  static const char* form = bcx_dev_env("BCX_OMP_FORM");
  static const bool off = bcx_dev_env("BCX_OMP_MULTI") != nullptr || (form && form[0] == 'm') || bcx_dev_env("BCX_OMP_UNFUSED") != nullptr;
  if (off || !s->grid_counter) return 1;
  NnlsArgs n;
  fill_nnls_args(s, n, nullptr);
  ResolveArgs r;
  bcx_fill_resolve_args(s, r, nullptr, exact);
  r.need_score = 1;                          // the score is compared with the negative direction (orthopursuit.py:32)
  s->k_ub += 1;                              // this step may add one slot
  const int rc = bcx_launch_omp_lh(s, n, &r);
  if (rc == 1) s->k_ub -= 1;
  return rc;
}

// ---- optimize(): Gram matrix on the fp64 matrix cores + cold-start NNLS ---------------------------
typedef double v4d __attribute__((ext_vector_type(4)));

// G[i][j] = row_i . row_j for i, j < k on v_mfma_f64_16x16x4_f64.  A wave owns a 32 x 32 block of G (2 x 2 MFMA tiles)
// and only blocks on or above the diagonal are computed, the mirror image is stored with them.  The four k-slots of an
// MFMA step are fed in a permuted k order -- lane group lk supplies k = 8t + 2 lk (+1) to steps 2t (2t+1) -- so one
// 16-byte load per operand tile feeds two steps: 4 loads per 8 MFMAs (round 1: one wave per 16 x 16 tile, 2 scalar
// loads per MFMA, 7.8 TFLOP/s).  Rows are d doubles apart; 8-byte loads when d is odd.
struct __attribute__((aligned(8))) gpd2 { double x, y; };
__global__ __launch_bounds__(256) void gram_mfma_kernel(const double* __restrict__ rows, int k, int d,
                                                        double* __restrict__ G, int64_t ldg, int nblk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  // upper-triangular block index -> (I, J), I <= J < nblk
  int t = blockIdx.x * 4 + wave;
  if (t >= nblk * (nblk + 1) / 2) return;
  int I = 0;
  while (t >= nblk - I) { t -= nblk - I; ++I; }
  const int J = I + t;
  const double* pa[2];
  const double* pb[2];
  bool va[2], vb[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int ri = I * 32 + 16 * u + li, rj = J * 32 + 16 * u + li;
    va[u] = ri < k; vb[u] = rj < k;
    pa[u] = rows + (size_t)(va[u] ? ri : 0) * d;
    pb[u] = rows + (size_t)(vb[u] ? rj : 0) * d;
  }
  v4d acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = (v4d){0.0, 0.0, 0.0, 0.0};
  const bool even = (d & 1) == 0;
  for (int k0 = 0; k0 < d; k0 += 8) {
    const int c = k0 + 2 * lk;
    gpd2 av[2], bv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (even && c + 1 < d) { av[u] = *(const gpd2*)(pa[u] + c); bv[u] = *(const gpd2*)(pb[u] + c); }
      else {
        av[u].x = c < d ? pa[u][c] : 0.0; av[u].y = c + 1 < d ? pa[u][c + 1] : 0.0;
        bv[u].x = c < d ? pb[u][c] : 0.0; bv[u].y = c + 1 < d ? pb[u][c + 1] : 0.0;
      }
      if (!va[u]) { av[u].x = 0.0; av[u].y = 0.0; }
      if (!vb[u]) { bv[u].x = 0.0; bv[u].y = 0.0; }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].x, bv[v].x, acc[u][v], 0, 0, 0);
        acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u].y, bv[v].y, acc[u][v], 0, 0, 0);
      }
  }
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = I * 32 + 16 * u + lk + 4 * r, col = J * 32 + 16 * v + li;
        if (row < k && col < k) {
          G[(size_t)row * ldg + col] = acc[u][v][r];
          if (I != J) G[(size_t)col * ldg + row] = acc[u][v][r];
        }
      }
}

__global__ __launch_bounds__(NN_THREADS) void optimize_kernel(NnlsArgs n, double tol) {
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  __shared__ double scratch[BCX_SCRATCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, d = a.d;
  const int k = st->k;
  refresh_state(a, scratch, k > 0);                 // prev_cost = error()   snnls.py:84
  const double prev_cost = st->err;
  for (int j = tid; j < k; j += blockDim.x) n.wbak[j] = a.act_w[j];
  for (int j = tid; j < d; j += blockDim.x) a.tmp[2 * (size_t)d + j] = a.xw[j];
  for (int j = wave; j < k; j += nw) {              // c = V b
    double acc = 0.0;
    for (int i = lane; i < d; i += 64) acc += a.act_rows[(size_t)j * d + i] * a.b[i];
    acc = wave_allsum(acc);
    if (lane == 0) n.cvec[j] = acc;
  }
  if (tid == 0) st->np = 0;
  for (int j = tid; j < k; j += blockDim.x) {
    n.ppos[j] = -1; n.x[j] = 0.0;
    n.flag[j] = (a.act_w[j] > 0.0) ? FLAG_INS : 0;   // nz_idcs = w > 0   snnls.py:86
  }
  __syncthreads();
  const double eps = 2.220446049250313e-16;
  const double tolscale = 10.0 * eps * (double)(d > k ? d : k) * st->bnorm;
  nnls_run(n, k, tolscale, scratch);
  for (int j = tid; j < k; j += blockDim.x)
    if (n.flag[j] & FLAG_INS) a.act_w[j] = (n.ppos[j] >= 0) ? n.x[j] : 0.0;
  __syncthreads();
  passive_combination(n, a.xw, a.tmp + 4 * (size_t)d);
  refresh_state(a, scratch, false);
  const double new_cost = st->err;
  if (new_cost > prev_cost * (1.0 + tol)) {         // snnls.py:91-97
    for (int j = tid; j < k; j += blockDim.x) a.act_w[j] = n.wbak[j];
    for (int j = tid; j < d; j += blockDim.x) a.xw[j] = a.tmp[2 * (size_t)d + j];
    __syncthreads();
    refresh_state(a, scratch, false);
    if (tid == 0) { st->limit = 1; st->hvalid = 0; }
  } else if (tid == 0) {
    st->hvalid = 1;
    st->hlo_valid = 0;
    st->since_refresh = 0;
  }
}

int bcx_launch_optimize(bcx_solver* s, double tol) {
  DevState h;
  BCX_HIP(hipStreamSynchronize(s->stream));
  BCX_HIP(hipMemcpy(&h, s->st, sizeof h, hipMemcpyDeviceToHost));
  const int k = h.k;
  const unsigned long long gram_since = bcx_gram_sk_epoch_now();
  // every place below that has synchronised the stream anyway: did a Gram launch of THIS call give up a wait (csrc/gram.hip)?
  auto gram_ok = [&]() -> int {
    if (!s->gram_work) return BCX_OK;
    const int t = bcx_gram_sk_timed_out(s->stream, s->gram_work, gram_since);
    if (t < 0) return t;
    if (t) { s->err = "optimize: the Gram kernel timed out waiting for a peer workgroup's partial tile (GPU shared or preempted); result discarded"; return BCX_ERR_TIMEOUT; }
    return BCX_OK;
  };
  if (k > 0) {
    static const bool old_gram = bcx_dev_env("BCX_GRAM_DIRECT") != nullptr;     // dev: round 2's kernel (operands straight from L2)
    const int kp64 = (k + 63) / 64 * 64;       // (the warm start forms the inverse of a kp64 x kp64 block with the same kernel)
    const size_t need = std::max((size_t)bcx_gram_rows_scratch_bytes(k, s->cfg.d), (size_t)bcx_gram_rows_scratch_bytes(kp64, kp64));
    if (!old_gram && s->gram_work_bytes < need) {
      if (s->gram_work) BCX_HIP(hipFree(s->gram_work));
      s->gram_work = nullptr; s->gram_work_bytes = 0;
      BCX_HIP(hipMalloc((void**)&s->gram_work, need));
      s->gram_work_bytes = need;
    }
    if (old_gram) {
      const int nblk = (k + 31) / 32;
      const int nwave = nblk * (nblk + 1) / 2;
      hipLaunchKernelGGL(gram_mfma_kernel, dim3((nwave + 3) / 4), dim3(256), 0, s->stream, s->act_rows, k,
                         s->cfg.d, s->gram, (int64_t)s->gram_cap, nblk);
      BCX_HIP(hipGetLastError());
    } else {
      const int rc = bcx_gram_rows(s->stream, s->act_rows, k, s->cfg.d, (int64_t)s->cfg.d, s->gram, (int64_t)s->gram_cap, s->gram_work);
      if (rc != BCX_OK) { s->err = "optimize: Gram kernel launch failed"; return rc; }
    }
  }
  if (k > 0) {
    // incremental Lawson-Hanson on the double-double inverse (omp_lh.hip), started WARM from the largest independent part of
    // the support (warm.hip: blocked Cholesky + fp64-MFMA inverse) where that applies; where the closing Newton check of
    // the warm run fails, the same kernel from the empty passive set; where that one's fails -- or its LDS budget does not
    // hold k slots -- the refined multi-workgroup solve of nnls_grid.hip, then the single-workgroup form
    static const bool cold_only = bcx_dev_env("BCX_OPT_COLD") != nullptr;      // dev / tests: never start warm
    const int32_t* warm_p = nullptr;
    if (!cold_only && s->gram_work) {
      const size_t need_w = bcx_warm_bytes(k);
      if (s->warm_bytes < need_w) {
        if (s->warm_buf) BCX_HIP(hipFree(s->warm_buf));
        s->warm_buf = nullptr; s->warm_bytes = 0;
        BCX_HIP(hipMalloc(&s->warm_buf, need_w));
        s->warm_bytes = need_w;
      }
      const int wrc = bcx_warm_start(s, k, s->warm_buf, s->gram_work, &warm_p);
      if (wrc < 0) return wrc;
      if (wrc != 0) warm_p = nullptr;
    }
    int rc = bcx_launch_optimize_lh(s, tol, k, warm_p);
    if (rc < 0) return rc;
    if (rc == 0 && warm_p) {
      s->opt_warm += 1;
      BCX_HIP(hipStreamSynchronize(s->stream));
      BCX_HIP(hipMemcpy(&h, s->st, sizeof h, hipMemcpyDeviceToHost));
      { const int g = gram_ok(); if (g != BCX_OK) return g; }
      if (h.omp_mode != OMP_OPT_FALLBACK) return BCX_OK;
      s->opt_warm_failed += 1;
      BCX_HIP(hipMemsetAsync((char*)s->st + offsetof(DevState, omp_mode), 0, sizeof(int32_t), s->stream));
      rc = bcx_launch_optimize_lh(s, tol, k, nullptr);
      if (rc < 0) return rc;
    }
    if (rc == 0) {
      BCX_HIP(hipStreamSynchronize(s->stream));
      BCX_HIP(hipMemcpy(&h, s->st, sizeof h, hipMemcpyDeviceToHost));
      { const int g = gram_ok(); if (g != BCX_OK) return g; }
      if (h.omp_mode != OMP_OPT_FALLBACK) return BCX_OK;
      s->opt_fallbacks += 1;
      BCX_HIP(hipMemsetAsync((char*)s->st + offsetof(DevState, omp_mode), 0, sizeof(int32_t), s->stream));
    }
    { const int g = gram_ok(); if (g != BCX_OK) return g; }      // (the forms below read the same G)
    rc = bcx_launch_optimize_grid(s, tol, k);             // 1 = not applicable
    if (rc <= 0) return rc;
  }
  NnlsArgs n;
  fill_nnls_args(s, n, nullptr);
  hipLaunchKernelGGL(optimize_kernel, dim3(1), dim3(NN_THREADS), 0, s->stream, n, tol);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}
