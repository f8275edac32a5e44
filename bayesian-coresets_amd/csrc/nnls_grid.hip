// nnls_grid.hip -- optimize(): w[active] = nnls(A[:, active], b) (snnls.py:82-97) as ONE launch of OPT_WGS
// co-resident workgroups.
//
// The active-set iteration of nnls.hip is O(k^3) passes over the k x k inverse H; one workgroup streams
// it at ~0.1 TB/s.  Here every workgroup runs the SAME control flow on replicated O(k) state kept in its
// own LDS (passive list, slot -> position, flags, x, z, two work vectors) and only the O(p^2) passes over
// H / G are partitioned (column blocks of the mat-vecs, row blocks of the rank-1 updates), with a grid
// barrier after each pass.  Scalars (Schur complements, step lengths, arg-max choices) are recomputed by
// every workgroup from the same data in the same order, so all workgroups -- and all shards of a
// row-sharded run, which replicate this solve -- take identical decisions bit for bit; no atomics touch
// floating-point data.
//
// Cold start: instead of Lawson-Hanson's empty passive set (k outer iterations of 6 passes each), all
// columns of the support join the passive set by bordering (2 passes each) with x = the current weights
// (feasible), and the iteration continues from there with its inner loop.  The NNLS minimiser is unique
// for independent columns, so the result is the reference's; dependent columns (k > d) are refused by the
// bordering test; the solve then restarts from the empty passive set in Lawson-Hanson's own order.
#include <hip/hip_runtime.h>
#include "nnls_common.h"

#define OPT_WGS 16
#define OPT_MAX_K 2048     // LDS per slot: 4 doubles + 3 ints = 44 bytes

// Cross-XCD rule of this kernel (per-XCD L2s are not coherent with each other, and an acquire fence drops the
// reader's L1 but not a stale line in its XCD's L2):
//   * H is OWNER-COMPUTES: row rr of the inverse is read and written only by wave (rr mod 16*G) -- the mat-vec
//     y = H v takes whole rows (H is symmetric), the rank-1 up/down-dates touch own rows, a new or moved row /
//     column entry is written by the row's owner.  No line of H is ever shared between workgroups.
//   * everything that does cross workgroups is a small EXCHANGE vector, written with write-through (sc1) stores
//     and read with sc1 loads after a grid barrier; consecutive exchanges rotate through four buffers, so a
//     buffer is rewritten only three barriers after it was read and no trailing barrier is needed.
// (A first version shared H between workgroups with plain accesses and fences only: 0.3 % of the runs of one
//  test problem re-read a stale line and took a different path; tests/race_hunt.py.)
static __device__ __forceinline__ double xld(const double* p) { return coh_load(p); }
static __device__ __forceinline__ void xst(double* p, double v) { coh_store(p, v); }

struct Grid {
  GridSync gs;
  int bi;        // barriers passed so far in this launch (identical in every workgroup)
  int xi;        // exchanges so far (ring position)
  int* s_flag;
  bool ok;       // false after a barrier timed out: everything below becomes a no-op
};
static __device__ __forceinline__ void gsync(Grid& g) {
  if (!g.ok) { __syncthreads(); return; }
  g.bi += 1;
  if (!grid_barrier(g.gs, g.bi, g.s_flag)) g.ok = false;
}
static __device__ __forceinline__ double* xbuf(const NnlsArgs& n, Grid& g) {   // next exchange buffer (gram_cap doubles)
  double* ring[4] = {n.t0, n.t1, n.t2, n.t3};
  return ring[(g.xi++) & 3];
}

struct Rep {             // replicated solver state in LDS
  double *t0, *t1, *x, *z;   // t0/t1/z by position, x by slot
  double* rv;                // d doubles: residual b - A z of the refinement step
  int *cs, *pos, *fl;        // position -> slot, slot -> position (-1), flags by slot
};

static __device__ __forceinline__ bool owns_row(int rr) {
  const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6;
  return rr % ((int)gridDim.x * nw) == (int)blockIdx.x * nw + wave;
}

// out[rr] = sum_cc H[rr][cc] v[cc] for the rows this wave owns (v in LDS); out is an exchange buffer
static __device__ void g_mv_rows(const NnlsArgs& n, int p, const double* v, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int rr = blockIdx.x * nw + wave; rr < p; rr += gridDim.x * nw) {
    const double* hrow = n.hinv + (size_t)rr * n.ldg;
    double acc = 0.0;
    for (int c0 = 0; c0 < p; c0 += 512) {
      double m[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) { const int c = c0 + t * 64 + lane; m[t] = c < p ? hrow[c] : 0.0; }
#pragma unroll
      for (int t = 0; t < 8; ++t) { const int c = c0 + t * 64 + lane; if (c < p) acc += m[t] * v[c]; }
    }
    acc = wave_allsum(acc);
    if (lane == 0) xst(&out[rr], acc);
  }
}

// Add `slot` to the passive set (bordered inverse).  False: numerically dependent on P.  1 barrier.
static __device__ bool g_border_add(const NnlsArgs& n, const Rep& r, int& p, int& ill, int slot, Grid& g,
                                    double* scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int64_t ld = n.ldg;
  for (int a = tid; a < p; a += blockDim.x) r.t0[a] = n.gram[(size_t)slot * ld + r.cs[a]];
  __syncthreads();
  double* X = xbuf(n, g);
  g_mv_rows(n, p, r.t0, X);                          // u = H g
  gsync(g);
  double v[1] = {0.0};
  for (int a = tid; a < p; a += blockDim.x) { const double u = xld(&X[a]); r.t1[a] = u; v[0] += r.t0[a] * u; }
  block_allsum<1>(v, scratch);
  const double gff = n.gram[(size_t)slot * ld + slot];
  const double s = gff - v[0];
  if (!(s > 1e-12 * gff)) return false;
  if (!(s > 1e-4 * gff)) ill = 1;
  const double inv = 1.0 / s;
  // H <- [[H + u u^T / s, -u/s], [-u^T/s, 1/s]]: every wave updates the rows it owns, incl. their new column p
  for (int rr = blockIdx.x * nw + wave; rr < p; rr += gridDim.x * nw) {
    const double ur = r.t1[rr] * inv;
    double* hrow = n.hinv + (size_t)rr * ld;
    for (int cc = lane; cc < p; cc += 64) hrow[cc] += ur * r.t1[cc];
    if (lane == 0) hrow[p] = -ur;
  }
  if (owns_row(p)) {
    double* hrow = n.hinv + (size_t)p * ld;
    for (int cc = lane; cc < p; cc += 64) hrow[cc] = -r.t1[cc] * inv;
    if (lane == 0) hrow[p] = inv;
  }
  __syncthreads();
  if (tid == 0) { r.cs[p] = slot; r.pos[slot] = p; }
  p += 1;
  __syncthreads();
  return true;
}

// Remove position q (rank-1 downdate, then the last position moves into q).  1-2 barriers.
static __device__ void g_border_del(const NnlsArgs& n, const Rep& r, int& p, int q, Grid& g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int64_t ld = n.ldg;
  const int last = p - 1;
  double* X = xbuf(n, g);
  if (owns_row(q)) {                                // row q == column q (symmetric): its owner publishes it
    const double* hrow = n.hinv + (size_t)q * ld;
    for (int cc = lane; cc < p; cc += 64) xst(&X[cc], hrow[cc]);
  }
  gsync(g);
  for (int a = tid; a < p; a += blockDim.x) r.t0[a] = xld(&X[a]);
  __syncthreads();
  const double hqq = r.t0[q];
  for (int rr = blockIdx.x * nw + wave; rr < p; rr += gridDim.x * nw) {
    const double tr = r.t0[rr];
    double* hrow = n.hinv + (size_t)rr * ld;
    for (int cc = lane; cc < p; cc += 64) hrow[cc] -= tr * r.t0[cc] / hqq;
  }
  const int gone = r.cs[q];
  if (q != last) {
    double* Y = xbuf(n, g);
    if (owns_row(last)) {                           // (same wave just finished the downdate of this row)
      const double* hrow = n.hinv + (size_t)last * ld;
      for (int cc = lane; cc < p; cc += 64) xst(&Y[cc], hrow[cc]);
    }
    gsync(g);
    for (int a = tid; a < p; a += blockDim.x) r.t1[a] = xld(&Y[a]);
    __syncthreads();
    for (int rr = blockIdx.x * nw + wave; rr < last; rr += gridDim.x * nw) {   // column q of the rows I own
      if (rr != q && lane == 0) n.hinv[(size_t)rr * ld + q] = r.t1[rr];
    }
    if (owns_row(q)) {
      double* hrow = n.hinv + (size_t)q * ld;
      for (int cc = lane; cc < last; cc += 64) if (cc != q) hrow[cc] = r.t1[cc];
      if (lane == 0) hrow[q] = r.t1[last];
    }
    __syncthreads();
    if (tid == 0) { const int moved = r.cs[last]; r.cs[q] = moved; r.pos[moved] = q; }
  }
  __syncthreads();
  if (tid == 0) r.pos[gone] = -1;
  p = last;
  __syncthreads();
}

// z = argmin on the passive set: z = H c_P, then refinement (as passive_solve in nnls.hip).  3+ barriers.
static __device__ void g_passive_solve(const NnlsArgs& n, const Rep& r, int p, int ill, Grid& g, double (*seg)[64],
                                       double* scratch) {
  const int tid = threadIdx.x;
  double cmax = 0.0;
  for (int q = tid; q < p; q += blockDim.x) { const double c = xld(&n.cvec[r.cs[q]]); r.t0[q] = c; cmax = fmax(cmax, fabs(c)); }
  cmax = block_allmax(cmax, scratch);
  double* X = xbuf(n, g);
  g_mv_rows(n, p, r.t0, X);
  gsync(g);
  for (int q = tid; q < p; q += blockDim.x) r.z[q] = xld(&X[q]);
  __syncthreads();
  // Refinement with the residual formed in DATA space, t1 = V_P (b - V_P^T z) (corrected semi-normal
  // equations): the Gram form c - G z loses cond(G) = cond(V)^2 digits, which shows as soon as the support
  // approaches d columns and the true residual is tiny; this form keeps the reference's (QR-based) accuracy.
  const ApplyArgs& a = n.a;
  const int d = a.d, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int max_it = ill ? 4 : 2;
  for (int it = 0; it < max_it; ++it) {
    for (int cb = blockIdx.x; cb * 64 < d; cb += gridDim.x) {      // A z on this workgroup's column blocks
      const int col = cb * 64 + lane;
      double acc = 0.0;
      if (col < d) {
        int q = wave;
        for (; q + 7 * nw < p; q += 8 * nw) {
          double m[8];
#pragma unroll
          for (int t = 0; t < 8; ++t) m[t] = a.act_rows[(size_t)r.cs[q + t * nw] * d + col];
#pragma unroll
          for (int t = 0; t < 8; ++t) acc += r.z[q + t * nw] * m[t];
        }
        for (; q < p; q += nw) acc += r.z[q] * a.act_rows[(size_t)r.cs[q] * d + col];
      }
      seg[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && col < d) {
        double t = seg[0][lane];
        for (int w = 1; w < nw; ++w) t += seg[w][lane];
        xst(&a.tmp[col], t);
      }
      __syncthreads();
    }
    gsync(g);
    for (int j = tid; j < d; j += blockDim.x) r.rv[j] = a.b[j] - xld(&a.tmp[j]);
    __syncthreads();
    double* Y = xbuf(n, g);
    for (int q = blockIdx.x * nw + wave; q < p; q += gridDim.x * nw) {   // V_P (b - A z): one wave per row
      const double* row = a.act_rows + (size_t)r.cs[q] * d;
      double acc = 0.0;
      for (int i0 = 0; i0 < d; i0 += 512) {
        double m[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int i = i0 + t * 64 + lane; m[t] = i < d ? row[i] : 0.0; }
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int i = i0 + t * 64 + lane; if (i < d) acc += m[t] * r.rv[i]; }
      }
      acc = wave_allsum(acc);
      if (lane == 0) xst(&Y[q], acc);
    }
    gsync(g);
    double rmax = 0.0;
    for (int q = tid; q < p; q += blockDim.x) { const double rvv = xld(&Y[q]); r.t1[q] = rvv; rmax = fmax(rmax, fabs(rvv)); }
    rmax = block_allmax(rmax, scratch);
    if (!(rmax > 1e-14 * cmax)) break;
    double* W = xbuf(n, g);
    g_mv_rows(n, p, r.t1, W);
    gsync(g);
    for (int q = tid; q < p; q += blockDim.x) r.z[q] += xld(&W[q]);
    __syncthreads();
  }
}

// Lawson-Hanson over the slots flagged FLAG_INS, from the passive set / x held in `r` (nnls_run of nnls.hip,
// same safeguards).  start_inner: the passive set was filled without solving -- begin with the inner loop.
static __device__ void g_nnls(const NnlsArgs& n, const Rep& r, int& p, int& ill, int k, double tolscale, bool start_inner,
                              Grid& g, double (*seg)[64], double* scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int max_outer = 3 * k + 16;
  for (int outer = 0; outer < max_outer && g.ok; ++outer) {
    int best = -1;
    if (!(outer == 0 && start_inner)) {
      // dual w = c - G x on the candidates: one wave per candidate, candidates dealt round-robin to all waves
      int nc = 0;
      for (int j = tid; j < k; j += blockDim.x) {
        const int fl = r.fl[j];
        if ((fl & FLAG_INS) && !(fl & FLAG_REJ) && r.pos[j] < 0) ++nc;
      }
      double cnt[1] = {(double)nc};
      block_allsum<1>(cnt, scratch);
      if (cnt[0] == 0.0) break;
      double* D = xbuf(n, g);
      for (int j = blockIdx.x * nw + wave; j < k; j += gridDim.x * nw) {
        const int fl = r.fl[j];
        if (!(fl & FLAG_INS) || (fl & FLAG_REJ) || r.pos[j] >= 0) continue;
        double acc = 0.0;
        for (int a = lane; a < p; a += 64) {
          const int ca = r.cs[a];
          acc += n.gram[(size_t)j * n.ldg + ca] * r.x[ca];
        }
        acc = wave_allsum(acc);
        if (lane == 0) xst(&D[j], xld(&n.cvec[j]) - acc);
      }
      gsync(g);
      double bv = -INFINITY; int bi = -1;
      for (int j = tid; j < k; j += blockDim.x) {
        const int fl = r.fl[j];
        if (!(fl & FLAG_INS) || (fl & FLAG_REJ) || r.pos[j] >= 0) continue;
        const double wv = xld(&D[j]);
        if (wv > tolscale * n.a.act_norm[j] && (bi < 0 || wv > bv)) { bv = wv; bi = j; }
      }
      const ArgBest pick = block_argbest(bv, bi, scratch);
      if (pick.i < 0) break;
      best = pick.i;
      if (!g_border_add(n, r, p, ill, best, g, scratch)) {
        if (tid == 0) r.fl[best] |= FLAG_REJ;
        __syncthreads();
        continue;
      }
      if (tid == 0) r.x[best] = 0.0;
      __syncthreads();
    }
    for (int inner = 0; inner < max_outer && g.ok; ++inner) {
      g_passive_solve(n, r, p, ill, g, seg, scratch);
      double amin = INFINITY; int apos = -1;
      for (int a = tid; a < p; a += blockDim.x) {
        const double za = r.z[a];
        if (!(za > 0.0)) {
          const double xa = r.x[r.cs[a]];
          double al = xa / (xa - za);
          if (!(al == al)) al = 0.0;
          if (apos < 0 || al < amin) { amin = al; apos = a; }
        }
      }
      const ArgBest worst = block_argbest(-amin, apos, scratch);
      if (worst.i < 0) {
        for (int a = tid; a < p; a += blockDim.x) r.x[r.cs[a]] = r.z[a];
        __syncthreads();
        break;
      }
      const double alpha = -worst.v;
      for (int a = tid; a < p; a += blockDim.x) {
        const int c = r.cs[a];
        const double xa = r.x[c];
        const double xn = xa + alpha * (r.z[a] - xa);
        const bool rm = (a == worst.i) || !(xn > 0.0);
        r.x[c] = rm ? 0.0 : xn;
        if (rm) r.fl[c] |= FLAG_RM;
      }
      __syncthreads();
      for (;;) {
        int cand = -1;
        for (int a = tid; a < p; a += blockDim.x)
          if (r.fl[r.cs[a]] & FLAG_RM) cand = a > cand ? a : cand;
        const ArgBest top = block_argbest((double)cand, cand, scratch);
        if (top.i < 0) break;
        const int slot = r.cs[top.i];
        if (tid == 0) {
          r.fl[slot] &= ~FLAG_RM;
          if (slot == best && inner == 0) r.fl[slot] |= FLAG_REJ;
        }
        __syncthreads();
        g_border_del(n, r, p, top.i, g);
        if (!g.ok) break;
      }
      if (p == 0) break;
    }
  }
}

__global__ __launch_bounds__(NN_THREADS) void optimize_grid_kernel(NnlsArgs n, GridSync gs, double tol, int kcap, int dpad) {
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  extern __shared__ double dyn[];
  Rep r;
  r.t0 = dyn; r.t1 = dyn + kcap; r.x = dyn + 2 * (size_t)kcap; r.z = dyn + 3 * (size_t)kcap;
  r.rv = dyn + 4 * (size_t)kcap;
  r.cs = (int*)(r.rv + dpad); r.pos = r.cs + kcap; r.fl = r.pos + kcap;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ double seg[NN_THREADS / 64][64];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, d = a.d;
  const int wg = blockIdx.x, nwg = gridDim.x;
  Grid g; g.gs = gs; g.bi = 0; g.xi = 0; g.s_flag = &s_flag; g.ok = true;
  const int k = st->k;
  double prev_cost = 0.0;
  if (wg == 0) {
    refresh_state(a, scratch, k > 0);                 // prev_cost = error()   snnls.py:84
    prev_cost = st->err;
    for (int j = tid; j < k; j += blockDim.x) n.wbak[j] = a.act_w[j];
    for (int j = tid; j < d; j += blockDim.x) a.tmp[2 * (size_t)d + j] = a.xw[j];
  }
  for (int j = wg * nw + wave; j < k; j += nwg * nw) {      // c = V b
    double acc = 0.0;
    for (int i = lane; i < d; i += 64) acc += a.act_rows[(size_t)j * d + i] * a.b[i];
    acc = wave_allsum(acc);
    if (lane == 0) xst(&n.cvec[j], acc);
  }
  for (int j = tid; j < k; j += blockDim.x) {
    r.pos[j] = -1; r.x[j] = 0.0;
    r.fl[j] = (a.act_w[j] > 0.0) ? FLAG_INS : 0;            // nz_idcs = w > 0   snnls.py:86
  }
  gsync(g);
  const double eps = 2.220446049250313e-16;
  const double tolscale = 10.0 * eps * (double)(d > k ? d : k) * st->bnorm;
  int p = 0, ill = st->omp_ill;
  bool refused = false;
  for (int j = 0; j < k && g.ok && !refused; ++j)
    if (r.fl[j] & FLAG_INS) refused = !g_border_add(n, r, p, ill, j, g, scratch);
  if (refused) {
    // dependent columns (k > d): an arbitrary maximal independent subset is a poor starting basis, so fall
    // back to Lawson-Hanson's own order -- empty passive set, columns enter by largest dual
    __syncthreads();
    for (int j = tid; j < k; j += blockDim.x) r.pos[j] = -1;
    p = 0;
    __syncthreads();
  } else {
    for (int j = tid; j < k; j += blockDim.x) r.x[j] = (r.pos[j] >= 0) ? a.act_w[j] : 0.0;
    __syncthreads();
  }
  g_nnls(n, r, p, ill, k, tolscale, p > 0, g, seg, scratch);
#ifdef BCX_DEBUG_GRID
  {
    // per-workgroup digest of the replicated state: all workgroups must agree
    double hx = 0.0; long long hc = 0;
    for (int j = 0; j < k; ++j) { hx += r.x[j] * (j + 1); }
    for (int q = 0; q < p; ++q) hc = hc * 31 + r.cs[q];
    if (tid == 0) { n.wbak[k + 8 + 4 * wg] = hx; n.wbak[k + 8 + 4 * wg + 1] = (double)hc; n.wbak[k + 8 + 4 * wg + 2] = (double)p; n.wbak[k + 8 + 4 * wg + 3] = (double)g.bi; }
  }
#endif
  if (wg != 0) return;
  // ---- workgroup 0: publish the passive data, new weights, accept / revert (snnls.py:88-97) ----------
  if (!g.ok) { if (tid == 0) { st->hvalid = 0; st->halt = HALT_GRID_TIMEOUT; } return; }
  for (int q = tid; q < p; q += blockDim.x) n.plist[q] = r.cs[q];
  for (int j = tid; j < k; j += blockDim.x) {
    n.ppos[j] = r.pos[j];
    n.x[j] = r.x[j];
    if (r.fl[j] & FLAG_INS) a.act_w[j] = (r.pos[j] >= 0) ? r.x[j] : 0.0;
  }
  if (tid == 0) { st->np = p; if (ill) st->omp_ill = 1; }
  __syncthreads();
  for (int j = tid; j < d; j += blockDim.x) {              // xw = sum_P x_j row_j, fixed position order
    double acc = 0.0;
    int q = 0;
    for (; q + 8 <= p; q += 8) {
      double m[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) m[t] = a.act_rows[(size_t)r.cs[q + t] * d + j];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += r.x[r.cs[q + t]] * m[t];
    }
    for (; q < p; ++q) acc += r.x[r.cs[q]] * a.act_rows[(size_t)r.cs[q] * d + j];
    a.xw[j] = acc;
  }
  __syncthreads();
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
    st->since_refresh = 0;
  }
}

int bcx_launch_optimize_grid(bcx_solver* s, double tol, int k) {
  if (k >= OPT_MAX_K || getenv("BCX_OPT_SINGLE")) return 1;   // caller uses the single-workgroup kernel
  if (!s->grid_counter) {
    BCX_HIP(hipMalloc((void**)&s->grid_counter, sizeof(unsigned long long)));
  }
  BCX_HIP(hipMemsetAsync(s->grid_counter, 0, sizeof(unsigned long long), s->stream));
  s->grid_epoch = 0;
  NnlsArgs n;
  fill_nnls_args(s, n, nullptr);
  const int kcap = (k + 1 + 63) / 64 * 64;
  const int dpad = (s->cfg.d + 63) / 64 * 64;
  const size_t lds = (size_t)kcap * (4 * sizeof(double) + 3 * sizeof(int)) + (size_t)dpad * sizeof(double);
  static bool allowed = false;
  if (!allowed) {
    BCX_HIP(hipFuncSetAttribute((const void*)optimize_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)((size_t)OPT_MAX_K * (4 * sizeof(double) + 3 * sizeof(int)) + (size_t)BCX_MAX_D * sizeof(double))));
    allowed = true;
  }
  GridSync gs;
  gs.counter = s->grid_counter;
  gs.base = 0;
  gs.timeout_ticks = 1000000000LL;   // 10 s
  hipLaunchKernelGGL(optimize_grid_kernel, dim3(OPT_WGS), dim3(NN_THREADS), lds, s->stream, n, gs, tol, kcap, dpad);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}
