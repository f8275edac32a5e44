// omp_lh.hip -- the OMP re-weight step  w[f] = 1; w[active] = nnls(A[:, active], b)  (orthopursuit.py:37-42) as ONE launch
// of 16 co-resident workgroups that keeps the Lawson-Hanson iteration incremental from end to end.
//
// The active set of the call is S = P u {f}: P = the columns that carry weight (the previous step's optimum: their duals
// are zero) and the newly selected column f.  So Lawson-Hanson's first pick is f, and every later change of the passive
// set is a single column entering or leaving.  Both have closed forms on the inverse H of the passive Gram block:
//   enter f :  u = H g, s = G_ff - g.u, t = (c_f - g.x) / s,   z = [x - t u ; t],   H <- bordered inverse
//   leave q :  h = H[:, q],                                    z <- z - (z_q / h_qq) h,   H <- rank-1 downdate
// so no step solves from scratch: the least-squares solution z of the current passive set is carried along with H, and
// the error of a closed-form update is proportional to the CHANGE it makes.
//
// What that needs is an H that does not drift.  In plain doubles it does: a nearly dependent column enters with a Schur
// complement s ~ 1e-10 G_ff, the entries of H grow to 1/s, and when the column leaves again the downdate cancels them to
// eps / s absolute -- on the Laplace-projected vectors of configs[2] (numerical rank ~100) H was good to four digits after
// a hundred steps, and a NumPy model of this step (tools/omp_lh_proto.py) against the CPU oracle needed 2-5 refinement
// iterations after EVERY change to keep the selections (round 2's kernel: four per solve, in one workgroup: 67-218 us per
// step from the first small Schur complement on, against 26 us for the closed form).  H is therefore kept in
// DOUBLE-DOUBLE (hinv + hinv_lo, ~32 digits): the mat-vec u = H g, the Schur complement, the bordered update and the
// downdate run on (hi, lo) pairs (error-free products by FMA, ~20-40 flops per element of a p x p matrix: nothing beside
// the memory latency that bounds this kernel), everything else -- weights, Gram entries, duals -- stays double.  H is then
// the exact inverse of the stored Gram block to working precision, the closed forms reproduce what the reference's
// solver computes from scratch (scipy.optimize.nnls solves the same normal equations G[P,P] z = c[P]) and no refinement
// pass exists: the model matched the oracle's 250 selections and its error to 8e-13 on every data set tried, where the
// best plain-double policy reached 7e-11 at 2.5 refinement iterations per change.
// A drift monitor costs nothing: the rows phase forms row_j . r for every passive row anyway (OMP's negative-direction
// select, orthopursuit.py:27-30), which is the gradient of the carried solution; dz = H_hi (V_P r) rides along in the pass
// that forms u = H g.  It stays at the Gram-vs-data discrepancy (<= 1e-7 of the weights on configs[2]); above 1e-4 -- or
// after a reverted step / a rejected optimize() -- the step first re-solves on P from scratch with the data-space
// refinement of optimize() (g_passive_solve).
//
// Cost: the usual step (f enters, every weight stays positive) is 3 grid barriers; every column that leaves adds 1-2
// (publish its row of H, move the last position into the hole).
//
// Replicated control: every workgroup holds the O(k) state in LDS (passive list, weights, z, feasible point) and takes the
// same decisions from the same data in the same order, so all workgroups -- and all shards of a row-sharded run -- stay
// bit-identical; H is owner-computes by rows and everything that crosses workgroups is an exchange vector (grid_lh.h).
// The number of barriers of a launch depends on the data, so the barrier base lives in device memory: workgroup 0 advances
// it at the end of the launch by what the launch used.
#include <stdlib.h>
#include <string.h>
#include "grid_lh.h"
#include "resolve_core.h"

#define OMPL_WGS 16
#define OMPL_STAMP(st, i) do { if (blockIdx.x == 0) BCX_STAMP(st, i); } while (0)
#define OMPL_LDS_MAX (150 * 1024)
#ifdef BCX_TIMING
// dev builds: one record per step (tools/omp_hist.py)
__device__ long long g_omp_log[4096][24];
extern "C" int bcx_debug_omp_log(long long* out, int n) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_omp_log), (size_t)n * 24 * sizeof(long long)) == hipSuccess ? 0 : -2;
}
#endif

// ---- double-double arithmetic (error-free transformations; explicit _rn intrinsics: no contraction across them) ----------
struct dd { double h, l; };
static __device__ __forceinline__ dd dd_make(double h, double l) { dd r; r.h = h; r.l = l; return r; }
static __device__ __forceinline__ dd two_sum(double a, double b) {
  const double s = __dadd_rn(a, b), bb = __dsub_rn(s, a);
  return dd_make(s, __dadd_rn(__dsub_rn(a, __dsub_rn(s, bb)), __dsub_rn(b, bb)));
}
static __device__ __forceinline__ dd quick_two_sum(double a, double b) {   // |a| >= |b|
  const double s = __dadd_rn(a, b);
  return dd_make(s, __dsub_rn(b, __dsub_rn(s, a)));
}
static __device__ __forceinline__ dd two_prod(double a, double b) {
  const double p = __dmul_rn(a, b);
  return dd_make(p, __fma_rn(a, b, -p));
}
static __device__ __forceinline__ dd dd_add(dd a, dd b) {
  dd s = two_sum(a.h, b.h);
  const dd t = two_sum(a.l, b.l);
  s.l = __dadd_rn(s.l, t.h);
  s = quick_two_sum(s.h, s.l);
  s.l = __dadd_rn(s.l, t.l);
  return quick_two_sum(s.h, s.l);
}
static __device__ __forceinline__ dd dd_neg(dd a) { return dd_make(-a.h, -a.l); }
static __device__ __forceinline__ dd dd_mul_d(dd a, double b) {
  dd p = two_prod(a.h, b);
  p.l = __fma_rn(a.l, b, p.l);
  return quick_two_sum(p.h, p.l);
}
static __device__ __forceinline__ dd dd_mul(dd a, dd b) {
  dd p = two_prod(a.h, b.h);
  p.l = __dadd_rn(p.l, __fma_rn(a.h, b.l, __dmul_rn(a.l, b.h)));
  return quick_two_sum(p.h, p.l);
}
static __device__ __forceinline__ dd dd_recip(dd a) {      // one Newton step on the double reciprocal: ~31 digits
  const double x = 1.0 / a.h;
  const dd r = dd_add(dd_make(1.0, 0.0), dd_neg(dd_mul_d(a, x)));
  return dd_add(dd_make(x, 0.0), dd_mul_d(r, x));
}
// butterfly all-reduce of a double-double over the wave (fixed association order; every lane ends with the total)
static __device__ __forceinline__ dd dd_wave_allsum(dd v) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const dd o = dd_make(__shfl_xor(v.h, off, BCX_WAVE), __shfl_xor(v.l, off, BCX_WAVE));
    v = dd_add(v, o);
  }
  return v;
}

// dev builds (-DBCX_OPT_PROFILE, tools/optimize_ab.sh): wall-clock ticks of optimize_lh_kernel by phase, accumulated in
// registers of workgroup 0 (prof[0..9]; prof[10] = the previous stamp) and left in DevState::dbg_t[0..7], [19], [31]
static __device__ __forceinline__ void optp(long long* prof, int i) {
  if (prof) { const long long now = wall_clock64(); prof[i] += now - prof[10]; prof[10] = now; }
}

// Elements of a row of H a lane keeps in flight in the owner-computes passes (hi and lo words: 2 * OMPL_NB loads per round
// trip).  Written one element at a time (load, arithmetic, store) the compiler keeps the order -- the hi and lo rows may alias
// as far as it knows -- and a pass over a row of 1024 entries was 16 dependent round trips.  8 with up to 512 threads per
// workgroup, 2 with 1024 (128 VGPRs per lane: 3 and more spill).
#define OMPL_NB_FOR(threads) ((threads) >= 1024 ? 2 : 8)
struct OmpLds {
  double *g, *gl, *u, *ul, *x, *z, *xs;   // g / u (hi, lo) / z / xs by position, x by slot
  double *xfs, *qs, *bs;                  // winner's row, residual query (later: refinement scratch), b
  int *cs, *pos, *fl;                     // position -> slot, slot -> position (-1), flags by slot
};

// (u, mon)[rr] = (H[rr] . v in double-double, H_hi[rr] . m in double) for the rows this wave owns; m may be null
template <int OMPL_NB> static __device__ __forceinline__ void ompl_mv_rows(const NnlsArgs& n, int p, const double* v, const double* m, double* Uh, double* Ul, double* M) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int rr = blockIdx.x * nw + wave; rr < p; rr += gridDim.x * nw) {
    const double* hh = n.hinv + (size_t)rr * n.ldg;
    const double* hl = n.hlo + (size_t)rr * n.ldg;
    dd acc = dd_make(0.0, 0.0);
    double mon = 0.0;
    for (int c0 = 0; c0 < p; c0 += 64 * OMPL_NB) {
      double a[OMPL_NB], b[OMPL_NB];
#pragma unroll
      for (int t = 0; t < OMPL_NB; ++t) { const int c = c0 + t * 64 + lane; a[t] = c < p ? hh[c] : 0.0; b[t] = c < p ? hl[c] : 0.0; }
#pragma unroll
      for (int t = 0; t < OMPL_NB; ++t) {
        const int c = c0 + t * 64 + lane;
        if (c < p) {
          acc = dd_add(acc, dd_mul_d(dd_make(a[t], b[t]), v[c]));
          if (m) mon += a[t] * m[c];
        }
      }
    }
    acc = dd_wave_allsum(acc);
    if (m) mon = wave_allsum(mon);
    if (lane == 0) { xst(&Uh[rr], acc.h); xst(&Ul[rr], acc.l); if (m) xst(&M[rr], mon); }
  }
}

// H <- [[H + u u^T / s, -u/s], [-u^T/s, 1/s]] for the column entering at position p (inv = 1/s): every wave updates the rows
// it owns, in double-double
template <int OMPL_NB> static __device__ __forceinline__ void ompl_border_apply(const NnlsArgs& n, const OmpLds& L, int p, dd inv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t ld = n.ldg;
  for (int rr = blockIdx.x * nw + wave; rr < p; rr += gridDim.x * nw) {
    const dd wr = dd_mul(dd_make(L.u[rr], L.ul[rr]), inv);
    double* hh = n.hinv + (size_t)rr * ld;
    double* hl = n.hlo + (size_t)rr * ld;
    for (int c0 = 0; c0 < p; c0 += 64 * OMPL_NB) {
      // three phases without a branch in the first two: loads (index clamped into the row), arithmetic, then the stores
      double a[OMPL_NB], b[OMPL_NB];
#pragma unroll
      for (int t = 0; t < OMPL_NB; ++t) { const int cc = min(c0 + t * 64 + lane, p - 1); a[t] = hh[cc]; b[t] = hl[cc]; }
#pragma unroll
      for (int t = 0; t < OMPL_NB; ++t) {
        const int cc = min(c0 + t * 64 + lane, p - 1);
        const dd v = dd_add(dd_make(a[t], b[t]), dd_mul(wr, dd_make(L.u[cc], L.ul[cc])));
        a[t] = v.h; b[t] = v.l;
      }
#pragma unroll
      for (int t = 0; t < OMPL_NB; ++t) { const int cc = c0 + t * 64 + lane; if (cc < p) { hh[cc] = a[t]; hl[cc] = b[t]; } }
    }
    if (lane == 0) { hh[p] = -wr.h; hl[p] = -wr.l; }
  }
  if (owns_row(p)) {
    double* hh = n.hinv + (size_t)p * ld;
    double* hl = n.hlo + (size_t)p * ld;
    for (int cc = lane; cc < p; cc += 64) {
      const dd w = dd_mul(dd_make(L.u[cc], L.ul[cc]), inv);
      hh[cc] = -w.h; hl[cc] = -w.l;
    }
    if (lane == 0) { hh[p] = inv.h; hl[p] = inv.l; }
  }
}

// the whole workgroup: an exchange vector pair (hi, lo) into LDS, four sc1 loads of each in flight per thread
static __device__ __forceinline__ void ompl_fetch_pair(const double* Xh, const double* Xl, double* dh, double* dl, int p) {
  for (int a0 = 0; a0 < p; a0 += 4 * (int)blockDim.x) {
    double h[4], l[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int a = min(a0 + t * (int)blockDim.x + (int)threadIdx.x, p - 1); h[t] = xld(&Xh[a]); l[t] = xld(&Xl[a]); }
#pragma unroll
    for (int t = 0; t < 4; ++t) { const int a = a0 + t * (int)blockDim.x + (int)threadIdx.x; if (a < p) { dh[a] = h[t]; dl[a] = l[t]; } }
  }
}

// one wave: a row of H (hi, lo) into an exchange buffer pair, all loads of a batch before its write-through stores
template <int OMPL_NB> static __device__ __forceinline__ void ompl_publish_row(const double* hh, const double* hl, double* Xh, double* Xl, int p) {
  const int lane = threadIdx.x & 63;
  for (int c0 = 0; c0 < p; c0 += 64 * OMPL_NB) {
    double a[OMPL_NB], b[OMPL_NB];
#pragma unroll
    for (int t = 0; t < OMPL_NB; ++t) { const int cc = min(c0 + t * 64 + lane, p - 1); a[t] = hh[cc]; b[t] = hl[cc]; }
#pragma unroll
    for (int t = 0; t < OMPL_NB; ++t) { const int cc = c0 + t * 64 + lane; if (cc < p) { xst(&Xh[cc], a[t]); xst(&Xl[cc], b[t]); } }
  }
}

// Position q leaves the passive set: closed-form update of the carried solution z, rank-1 downdate of H (double-double),
// the last position moves into the hole (z, the feasible point xs and the lists move with it).  ONE barrier: the owners
// publish row q and the not yet downdated row `last` together, every workgroup downdates its copy of the latter itself.
template <int OMPL_NB> static __device__ __forceinline__ void ompl_remove(const NnlsArgs& n, const OmpLds& L, int& p, int q, Grid& G, long long* prof = nullptr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int64_t ld = n.ldg;
  const int last = p - 1;
  double* Xh = xbuf(n, G);
  double* Xl = xbuf(n, G);
  double* Yh = xbuf(n, G);
  double* Yl = xbuf(n, G);
  if (owns_row(q)) {                                // row q == column q (symmetric): its owner publishes it
    const double* hh = n.hinv + (size_t)q * ld;
    const double* hl = n.hlo + (size_t)q * ld;
    ompl_publish_row<OMPL_NB>(hh, hl, Xh, Xl, p);
  }
  if (q != last && owns_row(last)) {
    const double* hh = n.hinv + (size_t)last * ld;
    const double* hl = n.hlo + (size_t)last * ld;
    ompl_publish_row<OMPL_NB>(hh, hl, Yh, Yl, p);
  }
  optp(prof, 6);
  gsync(G);
  optp(prof, 7);
  ompl_fetch_pair(Xh, Xl, L.g, L.gl, p);                                                        // h = H[:, q] (g is free by now)
  __syncthreads();
  const dd hqq = dd_make(L.g[q], L.gl[q]);
  const dd iq = dd_recip(hqq);
  const double f = L.z[q] / hqq.h;
  if (q != last) {
    // the downdated row `last` (what moves into the hole), formed by everybody from the two published rows
    const dd fl = dd_mul(dd_make(L.g[last], L.gl[last]), iq);
    for (int a0 = 0; a0 < p; a0 += 4 * (int)blockDim.x) {
      double yh[4], yl[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) { const int a = min(a0 + t * (int)blockDim.x + tid, p - 1); yh[t] = xld(&Yh[a]); yl[t] = xld(&Yl[a]); }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int a = a0 + t * (int)blockDim.x + tid;
        if (a < p) {
          const dd v = dd_add(dd_make(yh[t], yl[t]), dd_neg(dd_mul(fl, dd_make(L.g[a], L.gl[a]))));
          L.u[a] = v.h; L.ul[a] = v.l;
        }
      }
    }
  }
  __syncthreads();                                                   // every thread has read z[q] (f) before z[q] is rewritten
  for (int a = tid; a < p; a += blockDim.x) L.z[a] -= f * L.g[a];    // least-squares solution on P \ {q}
  __syncthreads();
  optp(prof, 8);
  for (int rr = blockIdx.x * nw + wave; rr < last; rr += gridDim.x * nw) {
    // rows that stay: downdate; row q takes the downdated row `last`, column q of every row its entry of it
    double* hh = n.hinv + (size_t)rr * ld;
    double* hl = n.hlo + (size_t)rr * ld;
    if (rr == q) {
      for (int cc = lane; cc < last; cc += 64) {
        const int src = cc == q ? last : cc;
        hh[cc] = L.u[src]; hl[cc] = L.ul[src];
      }
    } else {
      const dd fr = dd_mul(dd_make(L.g[rr], L.gl[rr]), iq);
      for (int c0 = 0; c0 < last; c0 += 64 * OMPL_NB) {
        double a[OMPL_NB], b[OMPL_NB];
#pragma unroll
        for (int t = 0; t < OMPL_NB; ++t) { const int cc = min(c0 + t * 64 + lane, last - 1); a[t] = hh[cc]; b[t] = hl[cc]; }
        const double mvh = L.u[rr], mvl = L.ul[rr];                   // (read only where q != last: column q takes the moved row's entry)
#pragma unroll
        for (int t = 0; t < OMPL_NB; ++t) {
          const int cc = min(c0 + t * 64 + lane, last - 1);
          const dd v = dd_add(dd_make(a[t], b[t]), dd_neg(dd_mul(fr, dd_make(L.g[cc], L.gl[cc]))));
          const bool hole = cc == q && q != last;
          a[t] = hole ? mvh : v.h; b[t] = hole ? mvl : v.l;
        }
#pragma unroll
        for (int t = 0; t < OMPL_NB; ++t) { const int cc = c0 + t * 64 + lane; if (cc < last) { hh[cc] = a[t]; hl[cc] = b[t]; } }
      }
    }
  }
  const int gone = L.cs[q];
  __syncthreads();
  if (tid == 0) {
    if (q != last) {
      const int moved = L.cs[last];
      L.cs[q] = moved; L.pos[moved] = q;
      L.z[q] = L.z[last]; L.xs[q] = L.xs[last];
    }
    L.pos[gone] = -1; L.x[gone] = 0.0;
  }
  p = last;
  __syncthreads();
  optp(prof, 9);
}

// Lawson-Hanson inner loop on the carried data: z = least-squares solution on the passive set, xs = a feasible point.
// Columns leave until z > 0; then x <- z.  `entered`: the slot that just entered (left again at once => never re-picked).
// n_out: members of the call's active set that are outside P and may still be picked (kept by every workgroup alike).
template <int OMPL_NB> static __device__ __forceinline__ int ompl_inner(const NnlsArgs& n, const OmpLds& L, int& p, int entered, int max_it, int& n_out, Grid& G,
                                 double* scratch, long long* prof = nullptr) {
  const int tid = threadIdx.x;
  int removed = 0;
  for (int inner = 0; inner < max_it && G.ok && p > 0; ++inner) {
    double amin = INFINITY; int apos = -1;
    for (int a = tid; a < p; a += blockDim.x) {
      const double za = L.z[a];
      if (!(za > 0.0)) {
        const double xa = L.xs[a];
        double al = xa / (xa - za);
        if (!(al == al)) al = 0.0;                 // 0/0: a zero weight asked to go further down
        if (apos < 0 || al < amin) { amin = al; apos = a; }
      }
    }
    const ArgBest worst = block_argbest(-amin, apos, scratch);   // smallest alpha, lowest position
    if (worst.i < 0) break;
    const double alpha = -worst.v;
    for (int a = tid; a < p; a += blockDim.x) {
      const double xa = L.xs[a];
      const double xn = xa + alpha * (L.z[a] - xa);
      const bool rm = (a == worst.i) || !(xn > 0.0);
      L.xs[a] = rm ? 0.0 : xn;
      if (rm) L.fl[L.cs[a]] |= FLAG_RM;
    }
    __syncthreads();
    for (;;) {                                     // highest position first: the one moved into a hole was already checked
      int cand = -1;
      for (int a = tid; a < p; a += blockDim.x)
        if (L.fl[L.cs[a]] & FLAG_RM) cand = a > cand ? a : cand;
      const ArgBest top = block_argbest((double)cand, cand, scratch);
      if (top.i < 0) break;
      const int slot = L.cs[top.i];
      const bool rej = slot == entered && inner == 0;              // LH safeguard: do not re-pick at once
      if (tid == 0) {
        L.fl[slot] &= ~FLAG_RM;
        if (rej) L.fl[slot] |= FLAG_REJ;
      }
      if (!rej) ++n_out;
      __syncthreads();
      optp(prof, 5);
      ompl_remove<OMPL_NB>(n, L, p, top.i, G, prof);
      ++removed;
      if (!G.ok) break;
    }
  }
  for (int a = tid; a < p; a += blockDim.x) L.x[L.cs[a]] = L.z[a];
  __syncthreads();
  optp(prof, 5);
  return removed;
}

// fused != 0 (single shard): the step starts from the scan's partials -- every workgroup runs the resolve phase itself
// (resolve_core.h: the same winner everywhere, nothing is exchanged) instead of reading the record a resolve_kernel
// launch left behind: one launch and one dependent launch boundary less per iteration.
// THREADS = the launch's workgroup width (256 / 512 / 1024): the register budget follows it (512 / 256 / 128 VGPRs).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void omp_lh_kernel(NnlsArgs n, GridSync gs, unsigned long long* base_ptr, int kcap, int dpad,
                                                            int force_resolve, int fused, ResolveArgs rsv) {
  constexpr int OMPL_NB = OMPL_NB_FOR(THREADS);
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, d = a.d;
  const int wg = blockIdx.x, nwg = gridDim.x;
  if (!st->active) return;
  extern __shared__ double dyn[];
  OmpLds L;
  L.g = dyn; L.gl = dyn + kcap; L.u = dyn + 2 * (size_t)kcap; L.ul = dyn + 3 * (size_t)kcap;
  L.x = dyn + 4 * (size_t)kcap; L.z = dyn + 5 * (size_t)kcap; L.xs = dyn + 6 * (size_t)kcap;
  L.xfs = dyn + 7 * (size_t)kcap; L.qs = L.xfs + dpad; L.bs = L.qs + dpad;
  L.cs = (int*)(L.bs + dpad); L.pos = L.cs + kcap; L.fl = L.pos + kcap;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ double seg[THREADS / 64 < 2 ? 2 : THREADS / 64][64];
  __shared__ double w_v[THREADS / 64];
  __shared__ long long w_i[THREADS / 64];
  __shared__ int w_s[THREADS / 64], w_np[THREADS / 64], w_m[THREADS / 64];
  __shared__ int s_win, s_ovf, s_flag;
  __shared__ Winner s_winner;
  OMPL_STAMP(st, 0);
  // The replicated state of the step -- solver scalars, query, b, the passive lists -- does not depend on the resolve phase:
  // its loads are issued first and ride out their round trip under the resolve phase's own (two dependent round trips
  // less between the winner and the rows phase).  Registers hold the first PRE_D x THREADS elements of the vectors and
  // PRE_K x THREADS list entries (every workgroup width covers its slot range); longer ones follow after the resolve.
  constexpr int PRE_D = 4, PRE_K = 2;
  double pre_q[PRE_D], pre_b[PRE_D], pre_x[PRE_K];
  int pre_cs[PRE_K], pre_pos[PRE_K];
#pragma unroll
  for (int t = 0; t < PRE_D; ++t) {
    const int i = tid + t * THREADS;
    pre_q[t] = i < d ? a.q64[i] : 0.0;
    pre_b[t] = i < d ? a.b[i] : 0.0;
  }
#pragma unroll
  for (int t = 0; t < PRE_K; ++t) {
    const int j = tid + t * THREADS;
    const bool ok = j < kcap && (int64_t)j < n.ldg;          // (inside the allocations whatever k is)
    pre_cs[t] = ok ? n.plist[j] : 0;
    pre_pos[t] = ok ? n.ppos[j] : -1;
    pre_x[t] = ok ? n.x[j] : 0.0;
  }
  const int k = st->k;
  int p = st->np;
  const double err0 = st->err, bnorm0 = st->bnorm;
  const int hvalid0 = st->hvalid, hlo0 = st->hlo_valid;
  // (what the end of the step needs of the solver state -- only workgroup 0's thread 0 changes it, at the very end)
  const int64_t it0 = st->it, itrs0 = st->itrs;
  const int since0 = st->since_refresh, no_mono0 = st->no_monotone;
  if (fused) {
    resolve_core<BCX_MAX_PARTIALS / THREADS>(rsv, &s_winner, L.xfs, scratch);          // winner's raw row lands in L.xfs
    if (tid == 0) { s_ovf = s_winner.flags == BCX_REC_OVERFLOW; s_win = s_winner.flags == BCX_REC_VALID ? 0 : -1; }
  } else if (tid == 0) { int o; s_win = omp_pick_record(a, &o); s_ovf = o; }
  __syncthreads();
  if (s_ovf || s_win < 0) {
    if (wg == 0 && tid == 0) { st->active = 0; st->halt = s_ovf ? HALT_NEED_EXACT : HALT_DONE; }
    return;
  }
  Grid G;
  G.gs = gs; G.gs.base = *base_ptr; G.bi = 0; G.xi = 0; G.s_flag = &s_flag; G.ok = true;
  const double* rec = a.recs + (size_t)(fused ? 0 : s_win) * (d + BCX_REC_HDR);
  const double rec_score = fused ? s_winner.score : rec[0];
  const int64_t rec_row = fused ? s_winner.gidx : (int64_t)rec[1];
  const double rec_norm = fused ? s_winner.norm : rec[2];
  const double* xf = rec + BCX_REC_HDR;
#pragma unroll
  for (int t = 0; t < PRE_D; ++t) {
    const int i = tid + t * THREADS;
    if (i < d) { L.qs[i] = pre_q[t]; L.bs[i] = pre_b[t]; }
  }
  for (int i = tid + PRE_D * THREADS; i < d; i += THREADS) { L.qs[i] = a.q64[i]; L.bs[i] = a.b[i]; }
  if (!fused) for (int i = tid; i < d; i += THREADS) L.xfs[i] = xf[i];
#pragma unroll
  for (int t = 0; t < PRE_K; ++t) {
    const int j = tid + t * THREADS;
    if (j < p) L.cs[j] = pre_cs[t];
    if (j <= k) {
      const int pj = (j < k && hvalid0) ? pre_pos[t] : -1;
      L.pos[j] = pj;
      L.x[j] = pj >= 0 ? pre_x[t] : 0.0;
      L.fl[j] = 0;
    }
  }
  for (int q = tid + PRE_K * THREADS; q < p; q += THREADS) L.cs[q] = n.plist[q];
  for (int j = tid + PRE_K * THREADS; j <= k; j += THREADS) {
    const int pj = (j < k && hvalid0) ? n.ppos[j] : -1;
    L.pos[j] = pj;
    L.x[j] = pj >= 0 ? n.x[j] : 0.0;
    L.fl[j] = 0;
  }
  __syncthreads();
  Rep R;
  R.t0 = L.g; R.t1 = L.u; R.x = L.x; R.z = L.z; R.rv = L.qs; R.cs = L.cs; R.pos = L.pos; R.fl = L.fl;
  bool resolve = force_resolve != 0;
  // ---- rows phase: row_j . x_f and row_j . r for every slot; one wave per row, all loads of a 512-element stretch in
  // flight together ----------------------------------------------------------------------------------------------------
  double* T3 = n.xr;                 // row_j . x_f  (the Gram row of a new slot)
  double* T2 = n.xr + n.ldg;         // row_j . r    (select of the negative direction; gradient of the carried solution)
  for (int j = wg * nw + wave; j < k; j += nwg * nw) {
    const double* row = a.act_rows + (size_t)j * d;
    double a0 = 0.0, a1 = 0.0;
    for (int i0 = 0; i0 < d; i0 += 512) {
      double rv[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) { const int i = i0 + t * 64 + lane; rv[t] = i < d ? row[i] : 0.0; }
#pragma unroll
      for (int t = 0; t < 8; ++t) { const int i = i0 + t * 64 + lane; if (i < d) { a0 += rv[t] * L.xfs[i]; a1 += rv[t] * L.qs[i]; } }
    }
    a0 = wave_allsum(a0);
    a1 = wave_allsum(a1);
    if (lane == 0) { xst(&T3[j], a0); xst(&T2[j], a1); }
  }
  // a stale passive set (a reverted step, a rejected optimize()): P = {slots with weight > 0}, H by successive bordering
  if (!hvalid0) {
    p = 0;
    for (int j = tid; j < k; j += blockDim.x) L.x[j] = a.act_w[j] > 0.0 ? a.act_w[j] : 0.0;
    __syncthreads();
    int ill = 0;
    for (int j = 0; j < k && G.ok; ++j) {
      if (!(L.x[j] > 0.0)) continue;
      if (!g_border_add(n, R, p, ill, j, G, scratch)) { __syncthreads(); if (tid == 0) L.x[j] = 0.0; __syncthreads(); }
    }
    resolve = true;
  }
  if (!hvalid0 || !hlo0) {
    // H was (re)written in plain doubles (the rebuild above, optimize(), the multi-kernel form): its low words are zero
    for (int rr = wg * nw + wave; rr < p; rr += nwg * nw)
      for (int cc = lane; cc < p; cc += 64) n.hlo[(size_t)rr * n.ldg + cc] = 0.0;
  }
  OMPL_STAMP(st, 1);
  gsync(G);                                                                                   // ---- B1
  OMPL_STAMP(st, 2);
  // ---- decide (every workgroup, one pass over the slots) -----------------------------------------------------------------
  const int64_t fpos = rec_row;
  const double nf = rec_norm;
  int npos = 0, match = 0x7fffffff;
  double bv = -INFINITY; long long bidx = -1; int bslot = -1;
  for (int j = tid; j < k; j += blockDim.x) {
    const long long gj = a.act_idx[j];
    if (gj == fpos && j < match) match = j;
    if (L.x[j] > 0.0) {
      ++npos;
      const double vv = -(xld(&T2[j]) / a.act_norm[j]);
      if (negbest_better(bv, bidx, vv, gj)) { bv = vv; bidx = gj; bslot = j; }
    }
  }
  double g2[2] = {0.0, 0.0};                                   // xf . xf, xf . b
  for (int i = tid; i < d; i += blockDim.x) { g2[0] += L.xfs[i] * L.xfs[i]; g2[1] += L.xfs[i] * L.bs[i]; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(bv, off, BCX_WAVE);
    const long long oi = __shfl_xor(bidx, off, BCX_WAVE);
    const int os = __shfl_xor(bslot, off, BCX_WAVE);
    if (negbest_better(bv, bidx, ov, oi)) { bv = ov; bidx = oi; bslot = os; }
    npos += __shfl_xor(npos, off, BCX_WAVE);
    match = min(match, __shfl_xor(match, off, BCX_WAVE));
  }
  g2[0] = wave_allsum(g2[0]);
  g2[1] = wave_allsum(g2[1]);
  if (lane == 0) { w_v[wave] = bv; w_i[wave] = bidx; w_s[wave] = bslot; w_np[wave] = npos; w_m[wave] = match; seg[0][wave] = g2[0]; seg[1][wave] = g2[1]; }
  __syncthreads();
  bv = w_v[0]; bidx = w_i[0]; bslot = w_s[0]; npos = w_np[0]; match = w_m[0];
  double xfxf = seg[0][0], xfb = seg[1][0];
  for (int w = 1; w < nw; ++w) {
    if (negbest_better(bv, bidx, w_v[w], w_i[w])) { bv = w_v[w]; bidx = w_i[w]; bslot = w_s[w]; }
    npos += w_np[w];
    match = min(match, w_m[w]);
    xfxf += seg[0][w]; xfb += seg[1][w];
  }
  __syncthreads();
  const bool checked = npos > 0;
  int64_t f = fpos;
  int slot = match == 0x7fffffff ? -1 : match;
  if (checked && !(rec_score >= bv)) { f = bidx; slot = bslot; }   // orthopursuit.py:32-35
  const bool fresh = slot < 0;
  if (fresh) slot = k;
  const int k1 = fresh ? k + 1 : k;
  const double gff = fresh ? xfxf : n.gram[(size_t)slot * n.ldg + slot];
  const double cf = fresh ? xfb : n.cvec[slot];
  const double nslot = fresh ? nf : a.act_norm[slot];
  const bool done = !fresh && L.pos[slot] >= 0;               // f already carries weight: the NNLS problem is unchanged
  const double eps = 2.220446049250313e-16;
  const double tolscale = 10.0 * eps * (double)(d > k1 ? d : k1) * bnorm0;
  // a new slot's data: row, Gram row / column (from the rows phase), c = row . b -- one workgroup, for later launches
  // (inside this launch the new Gram entries are read from T3)
  if (fresh && wg == nwg - 1) {
    for (int i = tid; i < d; i += blockDim.x) a.act_rows[(size_t)slot * d + i] = L.xfs[i];
    for (int j = tid; j < k; j += blockDim.x) {
      const double gv = xld(&T3[j]);
      n.gram[(size_t)slot * n.ldg + j] = gv;
      n.gram[(size_t)j * n.ldg + slot] = gv;
    }
    if (tid == 0) {
      a.act_idx[slot] = f; a.act_norm[slot] = nf;       // (weight, position: workgroup 0 at the commit)
      n.gram[(size_t)slot * n.ldg + slot] = gff;
      n.cvec[slot] = cf;
    }
  }
  OMPL_STAMP(st, 3);
  int n_removed = 0, n_entered = 0, did_resolve = 0;
  if (!done) {
    // members of the call's active set S = P u {f}
    for (int q = tid; q < p; q += blockDim.x) L.fl[L.cs[q]] = FLAG_INS;
    if (tid == 0) L.fl[slot] |= FLAG_INS;
    // ---- u = H g (double-double) and the drift monitor dz = H_hi (V_P r) in one pass over the rows this wave owns ---------
    for (int q = tid; q < p; q += blockDim.x) {
      L.g[q] = fresh ? xld(&T3[L.cs[q]]) : n.gram[(size_t)slot * n.ldg + L.cs[q]];
      L.xs[q] = xld(&T2[L.cs[q]]);                       // gradient of the carried solution (xs is free until the step)
    }
    __syncthreads();
    double* Uh = xbuf(n, G);
    double* Ul = xbuf(n, G);
    double* DZ = xbuf(n, G);
    ompl_mv_rows<OMPL_NB>(n, p, L.g, L.xs, Uh, Ul, DZ);
    OMPL_STAMP(st, 4);
    gsync(G);                                                                                 // ---- B2
    OMPL_STAMP(st, 5);
    // ---- the carried solution: z = x on P, feasible point xs = x; drift check ---------------------------------------------
    double cm = 0.0, xm = 0.0;
    for (int q = tid; q < p; q += blockDim.x) {
      const double dq = xld(&DZ[q]), xq = L.x[L.cs[q]];
      L.u[q] = xld(&Uh[q]); L.ul[q] = xld(&Ul[q]);
      L.xs[q] = xq;
      L.z[q] = xq;
      cm = fmax(cm, fabs(dq)); xm = fmax(xm, fabs(xq));
    }
    cm = block_allmax(cm, scratch);
    xm = block_allmax(xm, scratch);
    if (!(cm <= 1e-4 * xm) && p > 0) resolve = true;      // H no longer a good inverse of G_PP (NaN fails the '<=' too)
      int cand = slot, n_out = 1;
    bool have_u = true;
    if (resolve && p > 0) {
      // from scratch on P: z = H c_P refined in data space until the gradient is at rounding level (as optimize() does),
      // columns leave if that asks for it; then f enters from the re-solved state
      did_resolve = 1;
      g_passive_solve(n, R, p, 1, G, seg, scratch);       // (uses g / u / qs as scratch; H's high words)
      n_removed += ompl_inner<OMPL_NB>(n, L, p, -1, 3 * k1 + 16, n_out, G, scratch);
      for (int q = tid; q < p; q += blockDim.x) { L.xs[q] = L.x[L.cs[q]]; L.z[q] = L.xs[q]; }
      __syncthreads();
      have_u = false;
    }
    // ---- Lawson-Hanson outer loop: one column enters per pass (the first one is f) --------------------------------------
    for (int outer = 0; outer < 3 * k1 + 16 && G.ok && cand >= 0; ++outer) {
      const bool cfresh = fresh && cand == slot;
      if (!have_u) {
        for (int q = tid; q < p; q += blockDim.x) {
          const int c = L.cs[q];
          L.g[q] = cfresh ? xld(&T3[c]) : ((fresh && c == slot) ? xld(&T3[cand]) : n.gram[(size_t)cand * n.ldg + c]);
        }
        __syncthreads();
        double* U2h = xbuf(n, G);
        double* U2l = xbuf(n, G);
        ompl_mv_rows<OMPL_NB>(n, p, L.g, nullptr, U2h, U2l, nullptr);
        gsync(G);
        ompl_fetch_pair(U2h, U2l, L.u, L.ul, p);
        __syncthreads();
      }
      have_u = false;
      // Schur complement s = G_cc - g.u in double-double (g.u agrees with G_cc to all but the last digits when the column
      // is nearly dependent), dual w = c_c - g.z in double as the reference forms it
      dd gu = dd_make(0.0, 0.0);
      double r1[1] = {0.0};
      for (int q = tid; q < p; q += blockDim.x) {
        gu = dd_add(gu, dd_mul_d(dd_make(L.u[q], L.ul[q]), L.g[q]));
        r1[0] += L.g[q] * L.z[q];
      }
      gu = dd_wave_allsum(gu);
      if (lane == 0) { seg[0][wave] = gu.h; seg[1][wave] = gu.l; }
      block_allsum<1>(r1, scratch);                       // (its barriers also publish seg)
      gu = dd_make(seg[0][0], seg[1][0]);
      for (int w = 1; w < nw; ++w) gu = dd_add(gu, dd_make(seg[0][w], seg[1][w]));
      __syncthreads();
      const double gcc = cand == slot ? gff : n.gram[(size_t)cand * n.ldg + cand];
      const double ccand = cand == slot ? cf : n.cvec[cand];
      const double ncand = cand == slot ? nslot : a.act_norm[cand];
      const dd sc = dd_add(dd_make(gcc, 0.0), dd_neg(gu));
      const double wv = ccand - r1[0];
      bool entered = false;
      if (!(wv > tolscale * ncand)) {
        // dual not positive: the column stays at weight 0 (and no other candidate can have a larger dual: it was the arg-max)
        cand = -1;
      } else if (!(sc.h > 1e-12 * gcc)) {
        if (tid == 0) L.fl[cand] |= FLAG_REJ;             // numerically dependent on P
        --n_out;
        __syncthreads();
      } else {
        const double t = wv / sc.h;
        const dd inv = dd_recip(sc);
        for (int q = tid; q < p; q += blockDim.x) L.z[q] -= t * L.u[q];
        ompl_border_apply<OMPL_NB>(n, L, p, inv);
        if (tid == 0) { L.z[p] = t; L.xs[p] = 0.0; L.cs[p] = cand; L.pos[cand] = p; }
        p += 1;
        ++n_entered;
        --n_out;
        entered = true;
        __syncthreads();
      }
      if (cand < 0) break;                                // nothing entered, nothing can leave: x stays as it is, bit for bit
      n_removed += ompl_inner<OMPL_NB>(n, L, p, entered ? cand : -1, 3 * k1 + 16, n_out, G, scratch);
      for (int q = tid; q < p; q += blockDim.x) { L.xs[q] = L.x[L.cs[q]]; L.z[q] = L.xs[q]; }
      __syncthreads();
      // next candidate: the member of S without weight that has the largest positive dual (only columns that left in this
      // call can qualify); duals c_j - G[j, P] x are formed by every workgroup, one wave per candidate
      cand = -1;
      if (n_out <= 0) break;
      for (int j = wave; j < k1; j += nw) {
        const int fl = L.fl[j];
        if (!(fl & FLAG_INS) || (fl & FLAG_REJ) || L.pos[j] >= 0) continue;
        const bool jf = fresh && j == slot;
        double acc = 0.0;
        for (int q = lane; q < p; q += 64) {
          const int c = L.cs[q];
          const double gv = jf ? xld(&T3[c]) : ((fresh && c == slot) ? xld(&T3[j]) : n.gram[(size_t)j * n.ldg + c]);
          acc += gv * L.x[c];
        }
        acc = wave_allsum(acc);
        if (lane == 0) L.g[j] = (j == slot ? cf : n.cvec[j]) - acc;      // (g by SLOT here; refilled by position above)
      }
      __syncthreads();
      double dbv = -INFINITY; int dbi = -1;
      for (int j = tid; j < k1; j += blockDim.x) {
        const int fl = L.fl[j];
        if (!(fl & FLAG_INS) || (fl & FLAG_REJ) || L.pos[j] >= 0) continue;
        const double wvj = L.g[j];
        const double nj = j == slot ? nslot : a.act_norm[j];
        if (wvj > tolscale * nj && (dbi < 0 || wvj > dbv)) { dbv = wvj; dbi = j; }
      }
      const ArgBest pick = block_argbest(dbv, dbi, scratch);
      cand = pick.i;
    }
  }
  OMPL_STAMP(st, 6);
  // ---- xw' = sum_P x_j row_j on 64-column blocks ------------------------------------------------------------------------
  for (int q = tid; q < p; q += blockDim.x) L.z[q] = L.x[L.cs[q]];
  __syncthreads();
  const int pf = (fresh && L.pos[slot] >= 0) ? L.pos[slot] : -1;      // the new slot's row is in LDS, not in act_rows yet
  for (int cb = wg; cb * 64 < d; cb += nwg) {
    const int col = cb * 64 + lane;
    double acc = 0.0;
    if (col < d) {
      int q = wave;
      for (; q + 7 * nw < p; q += 8 * nw) {
        double m[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { const int qq = q + t * nw; m[t] = qq == pf ? L.xfs[col] : a.act_rows[(size_t)L.cs[qq] * d + col]; }
#pragma unroll
        for (int t = 0; t < 8; ++t) acc += L.z[q + t * nw] * m[t];
      }
      for (; q < p; q += nw) acc += L.z[q] * (q == pf ? L.xfs[col] : a.act_rows[(size_t)L.cs[q] * d + col]);
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
  OMPL_STAMP(st, 7);
  if (wg != 0) { if (G.ok) grid_arrive(G.gs, 1); return; }
  gsync(G);                                                                                   // ---- last barrier
  OMPL_STAMP(st, 8);
  if (!G.ok) { if (tid == 0) { st->active = 0; st->halt = HALT_GRID_TIMEOUT; st->hvalid = 0; } return; }
  // ---- workgroup 0: error, monotone check, commit or revert, trace, next query ----------------------------------------------
  double v[2] = {0.0, 0.0};
  for (int j = tid; j < d; j += blockDim.x) {
    const double x = xld(&a.tmp[j]), rr = x - L.bs[j];
    L.qs[j] = x;
    v[0] += rr * rr; v[1] += x * x;
  }
  block_allsum<2>(v, scratch);
  const double new_err = sqrt(v[0]);
  int status = BCX_IT_OK;
  if (checked && !no_mono0 && new_err > err0) status = BCX_IT_FAIL_MONOTONE;      // snnls.py:56-58
  // The usual end of a step -- accepted, more iterations to go, no periodic refresh of xw due -- writes the next select's
  // query b - xw' (prepare_query, orthopursuit.py:18) in the pass that commits xw', from the copies in LDS: prepare_next
  // would read the counters, b and xw' back from memory (three dependent round trips, ~2 us of every step).
  const bool fast_next = status == BCX_IT_OK && it0 + 1 < itrs0 && !(a.refresh_every > 0 && since0 + 1 >= a.refresh_every);
  if (status == BCX_IT_OK) {
    for (int q = tid; q < p; q += blockDim.x) n.plist[q] = L.cs[q];
    for (int j = tid; j < k1; j += blockDim.x) {
      const int pj = L.pos[j];
      n.ppos[j] = pj;
      n.x[j] = pj >= 0 ? L.x[j] : 0.0;
      a.act_w[j] = pj >= 0 ? L.x[j] : 0.0;
    }
    for (int j = tid; j < d; j += blockDim.x) {
      a.xw[j] = L.qs[j];
      if (fast_next) store_query(a, 0, j, L.bs[j] - L.qs[j]);
    }
    if (tid == 0) {
      st->k = k1;
      st->np = p;
      st->hvalid = 1;
      st->hlo_valid = 1;
      st->err = new_err;
      const double nwn = sqrt(v[1]);
      st->nw = nwn == 0.0 ? 1.0 : nwn;
      st->since_refresh = since0 + 1;
      if (checked && !no_mono0) st->retried = 0;
      if (fast_next) st->qscale = new_err;
    }
  } else if (tid == 0) {
    st->hvalid = 0;        // weights were not touched; H and the lists changed: rebuilt from the weights next time
  }
  if (tid == 0) {
    *base_ptr = G.gs.base + (unsigned long long)G.bi * (unsigned long long)nwg;
    a.tr_sel[it0] = f; a.tr_err[it0] = status == BCX_IT_OK ? new_err : err0; a.tr_status[it0] = status;
    st->it = it0 + 1;
    st->exact_mode = 0;
    st->omp_mode = OMP_IDLE;
    st->n_omp[0] += 1;
    st->n_omp[1] += n_removed;
    st->n_omp[2] += did_resolve;
    st->n_omp[3] += (n_entered > 1) ? n_entered - 1 : 0;
    if (status != BCX_IT_OK) {
      if (st->retried) { st->limit = 1; st->active = 0; st->halt = HALT_LIMIT; }
      else st->retried = 1;
    }
  }
  OMPL_STAMP(st, 9);
#ifdef BCX_TIMING
  const int64_t log_it = it0;
#endif
  if (!fast_next) {
    __syncthreads();
    if (st->active) { [[clang::always_inline]] prepare_next(a, scratch); }
  }
  OMPL_STAMP(st, 10);
#ifdef BCX_TIMING
  __syncthreads();
  if (tid == 0 && log_it >= 0 && log_it < 4096) {
    long long* Lg = g_omp_log[log_it];
    Lg[0] = log_it; Lg[1] = k; Lg[2] = st->np; Lg[3] = done ? 1 : 2; Lg[4] = did_resolve ? 4 : 3; Lg[5] = status; Lg[6] = st->np; Lg[7] = G.bi;
    for (int i = 0; i < 12; ++i) Lg[8 + i] = i <= 10 ? st->dbg_t[i] - st->dbg_t[0] : 0;
    Lg[19] = blockDim.x; Lg[20] = n_entered; Lg[21] = n_removed; Lg[22] = did_resolve; Lg[23] = G.bi;
  }
#endif
}

// ---- optimize(): w[active] = nnls(A[:, active], b) (snnls.py:82-97) on the same incremental machinery ------------------
// Lawson-Hanson from the empty passive set in its own order -- the dual of every candidate, the largest enters, columns
// leave while the solution is not positive -- with the carried solution z and the double-double inverse of this file:
// entering costs two grid barriers (duals; u = H g), leaving one.  No solve from scratch and no refinement inside the
// iteration (nnls_grid.hip re-solves z = H c with two data-space refinement passes after every change: 185 us per
// entering column at k = 1497, d = 1024).  At the end ONE data-space Newton step on the passive set, x += H (V_P (b - V_P^T x)),
// measures what the Gram-space recurrences lost -- it is applied when it is a correction (<= 1e-3 of the weights and every
// weight stays positive); otherwise the launch restores the weights and reports OMP_OPT_FALLBACK, and the host runs the
// refined solve of nnls_grid.hip instead.
#define OPTL_MAX_WGS 128
#define OPTL_BATCH 8
// warm_p != null (csrc/warm.hip ran on the stream before this launch): the passive set does not start empty -- n.plist holds
// *warm_p slots in order, n.hinv the (plain-double) inverse of their Gram block, low words zero.  The feasible point is the
// current weights on those slots, z = H c their least-squares solution; the columns whose solution is not positive leave
// through the same inner loop as after any entering column, and the iteration goes on from there.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void optimize_lh_kernel(NnlsArgs n, GridSync gs, double tol, int kcap, const int32_t* warm_p) {
  constexpr int OMPL_NB = OMPL_NB_FOR(THREADS);
#ifdef BCX_OPT_PROFILE
  long long prof_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long* const prof = prof_;
#else
  long long* const prof = nullptr;
#endif
  const ApplyArgs& a = n.a;
  DevState* st = a.st;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, d = a.d;
  const int wg = blockIdx.x, nwg = gridDim.x;
  extern __shared__ double dyn[];
  OmpLds L;
  L.g = dyn; L.gl = dyn + kcap; L.u = dyn + 2 * (size_t)kcap; L.ul = dyn + 3 * (size_t)kcap;
  L.x = dyn + 4 * (size_t)kcap; L.z = dyn + 5 * (size_t)kcap; L.xs = dyn + 6 * (size_t)kcap;
  L.xfs = nullptr; L.qs = nullptr; L.bs = nullptr;
  double* Ld = dyn + 7 * (size_t)kcap;                      // duals of the current pass, by slot
  L.cs = (int*)(dyn + 8 * (size_t)kcap); L.pos = L.cs + kcap; L.fl = L.pos + kcap;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ double seg[THREADS / 64 < 2 ? 2 : THREADS / 64][64];
  __shared__ int s_flag;
  Grid G;
  G.gs = gs; G.bi = 0; G.xi = 0; G.s_flag = &s_flag; G.ok = true;
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
  double cnt[1] = {0.0};
  for (int j = tid; j < k; j += blockDim.x) {
    const int ins = a.act_w[j] > 0.0;                       // nz_idcs = w > 0   snnls.py:86
    L.pos[j] = -1; L.x[j] = 0.0; L.fl[j] = ins ? FLAG_INS : 0;
    cnt[0] += ins;
  }
  block_allsum<1>(cnt, scratch);
  int n_out = (int)cnt[0];
  gsync(G);
  const double eps = 2.220446049250313e-16;
  const double bnorm = st->bnorm;
  const double tolscale = 10.0 * eps * (double)(d > k ? d : k) * bnorm;
  int p = 0;
  const int max_outer = 3 * k + 16;
  int dg_p0 = 0, dg_p1 = 0, dg_in = 0, dg_out = 0, dg_outer = 0;      // diagnostics of the call (DevState::dbg_t[20..], bcx_debug_stamps)
  if (warm_p) {
    p = *warm_p;
    if (p > k) p = 0;                                        // (never: guards the LDS arrays)
    for (int q = tid; q < p; q += blockDim.x) {
      const int slot = n.plist[q];
      L.cs[q] = slot; L.pos[slot] = q;
      L.xs[q] = a.act_w[slot];
      L.g[q] = xld(&n.cvec[slot]);
    }
    __syncthreads();
    n_out -= p;
    if (p > 0) {
      double* Zh = xbuf(n, G);
      double* Zl = xbuf(n, G);
      ompl_mv_rows<OMPL_NB>(n, p, L.g, nullptr, Zh, Zl, nullptr);     // z = H c
      gsync(G);
      for (int q = tid; q < p; q += blockDim.x) L.z[q] = xld(&Zh[q]);
      __syncthreads();
      dg_p0 = p;
      optp(prof, 11);
      dg_out += ompl_inner<OMPL_NB>(n, L, p, -1, max_outer, n_out, G, scratch, prof);
      dg_p1 = p;
      for (int q = tid; q < p; q += blockDim.x) { L.xs[q] = L.x[L.cs[q]]; L.z[q] = L.xs[q]; }
      __syncthreads();
      optp(prof, 5);
    }
  }
  optp(prof, 11);
  bool more = true;
  for (int outer = 0; outer < max_outer && G.ok && n_out > 0 && more; ++outer) {
    ++dg_outer;
    optp(prof, 11);
    // duals w_j = c_j - G[j, P] x of the members without weight: one wave per candidate, dealt over all workgroups
    double* D = xbuf(n, G);
    for (int j = wg * nw + wave; j < k; j += nwg * nw) {
      const int fl = L.fl[j];
      if (!(fl & FLAG_INS) || (fl & FLAG_REJ) || L.pos[j] >= 0) continue;
      const double* grow = n.gram + (size_t)j * n.ldg;
      double acc = 0.0;
      for (int q0 = 0; q0 < p; q0 += 256) {
        double gv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int q = q0 + t * 64 + lane; gv[t] = q < p ? grow[L.cs[q]] : 0.0; }
#pragma unroll
        for (int t = 0; t < 4; ++t) { const int q = q0 + t * 64 + lane; if (q < p) acc += gv[t] * L.z[q]; }
      }
      acc = wave_allsum(acc);
      if (lane == 0) xst(&D[j], xld(&n.cvec[j]) - acc);
    }
    gsync(G);
    for (int j = tid; j < k; j += blockDim.x) {
      const int fl = L.fl[j];
      const bool cnd = (fl & FLAG_INS) && !(fl & FLAG_REJ) && L.pos[j] < 0;
      const double dj = cnd ? xld(&D[j]) : -INFINITY;
      Ld[j] = (cnd && dj > tolscale * a.act_norm[j]) ? dj : -INFINITY;     // (the threshold here: the picks below stay in LDS)
    }
    __syncthreads();
    // Up to OPTL_BATCH columns enter per dual pass, in the order of these duals.  Lawson-Hanson asks for A column with a
    // positive dual (the largest is its heuristic): from the second of a batch on the list is stale, so the candidate's
    // exact dual at the current solution is formed first -- replicated, O(p), no barrier -- and a candidate whose dual
    // is no longer positive waits for the next pass.  One grid barrier less per entering column.
    optp(prof, 0);                                             // the dual pass
    for (int bth = 0; bth < OPTL_BATCH && G.ok; ++bth) {
      optp(prof, 11);
      double dbv = -INFINITY; int dbi = -1;
      for (int j = tid; j < k; j += blockDim.x) {
        const double wvj = Ld[j];
        if (wvj > -INFINITY && (dbi < 0 || wvj > dbv)) { dbv = wvj; dbi = j; }
      }
      const ArgBest pick = block_argbest(dbv, dbi, scratch);
      const int cand = pick.i;
      if (cand < 0) { if (bth == 0) more = false; break; }
      if (tid == 0) Ld[cand] = -INFINITY;
      // (the candidate's scalars travel with the gather below: one round trip to memory instead of four dependent ones)
      const double ncand = a.act_norm[cand], ccand = xld(&n.cvec[cand]), gcc = n.gram[(size_t)cand * n.ldg + cand];
      // g = G[cand, P], exact dual c_cand - g . z
      double r1[1] = {0.0};
      for (int q = tid; q < p; q += blockDim.x) { const double gq = n.gram[(size_t)cand * n.ldg + L.cs[q]]; L.g[q] = gq; r1[0] += gq * L.z[q]; }
      block_allsum<1>(r1, scratch);
      const double wv = ccand - r1[0];
      if (!(wv > tolscale * ncand)) continue;
      optp(prof, 1);                                           // pick + gather g + exact dual
      if (p > 0) {
        double* Uh = xbuf(n, G);
        double* Ul = xbuf(n, G);
        ompl_mv_rows<OMPL_NB>(n, p, L.g, nullptr, Uh, Ul, nullptr);
        optp(prof, 2);
        gsync(G);
        optp(prof, 3);
        ompl_fetch_pair(Uh, Ul, L.u, L.ul, p);
        __syncthreads();
      }
      dd gu = dd_make(0.0, 0.0);
      for (int q = tid; q < p; q += blockDim.x) gu = dd_add(gu, dd_mul_d(dd_make(L.u[q], L.ul[q]), L.g[q]));
      gu = dd_wave_allsum(gu);
      if (lane == 0) { seg[0][wave] = gu.h; seg[1][wave] = gu.l; }
      __syncthreads();
      gu = dd_make(seg[0][0], seg[1][0]);
      for (int w = 1; w < nw; ++w) gu = dd_add(gu, dd_make(seg[0][w], seg[1][w]));
      __syncthreads();
      const dd sc = dd_add(dd_make(gcc, 0.0), dd_neg(gu));
      if (!(sc.h > 1e-12 * gcc)) {                            // numerically dependent on P: never enters
        if (tid == 0) L.fl[cand] |= FLAG_REJ;
        --n_out;
        __syncthreads();
        continue;
      }
      const double t = wv / sc.h;
      const dd inv = dd_recip(sc);
      for (int q = tid; q < p; q += blockDim.x) L.z[q] -= t * L.u[q];      // (z == xs == x on P at this point)
      ompl_border_apply<OMPL_NB>(n, L, p, inv);
      if (tid == 0) { L.z[p] = t; L.xs[p] = 0.0; L.cs[p] = cand; L.pos[cand] = p; }
      p += 1;
      --n_out;
      ++dg_in;
      __syncthreads();
      optp(prof, 4);
      dg_out += ompl_inner<OMPL_NB>(n, L, p, cand, max_outer, n_out, G, scratch, prof);
      for (int q = tid; q < p; q += blockDim.x) { L.xs[q] = L.x[L.cs[q]]; L.z[q] = L.xs[q]; }
      __syncthreads();
    }
  }
  // ---- one data-space Newton step on the passive set: xw0 = V_P^T x | gv = V_P (b - xw0) | dz = H gv ---------------------
  auto combine = [&](double* out) {                          // out[col] = sum_q z[q] row_{cs[q]}[col], fixed position order
    for (int cb = wg; cb * 64 < d; cb += nwg) {
      const int col = cb * 64 + lane;
      double acc = 0.0;
      if (col < d) {
        int q = wave;
        for (; q + 7 * nw < p; q += 8 * nw) {
          double m[8];
#pragma unroll
          for (int t8 = 0; t8 < 8; ++t8) m[t8] = a.act_rows[(size_t)L.cs[q + t8 * nw] * d + col];
#pragma unroll
          for (int t8 = 0; t8 < 8; ++t8) acc += L.z[q + t8 * nw] * m[t8];
        }
        for (; q < p; q += nw) acc += L.z[q] * a.act_rows[(size_t)L.cs[q] * d + col];
      }
      seg[wave][lane] = acc;
      __syncthreads();
      if (wave == 0 && col < d) {
        double tsum = seg[0][lane];
        for (int w = 1; w < nw; ++w) tsum += seg[w][lane];
        xst(&out[col], tsum);
      }
      __syncthreads();
    }
  };
  // (after a warm start H began as a plain-double inverse: a second step takes what the first left -- each contracts the
  // error by ~cond(G) eps -- and costs three barriers)
  bool fallback = false;
  double dm = 0.0, xm = 0.0, negs = 0.0;
  for (int pass = 0; pass < (warm_p ? 2 : 1); ++pass) {
    combine(a.tmp);
    gsync(G);
    double* GV = xbuf(n, G);
    for (int q = wg * nw + wave; q < p; q += nwg * nw) {
      const double* row = a.act_rows + (size_t)L.cs[q] * d;
      double acc = 0.0;
      for (int i = lane; i < d; i += 64) acc += row[i] * (a.b[i] - xld(&a.tmp[i]));
      acc = wave_allsum(acc);
      if (lane == 0) xst(&GV[q], acc);
    }
    gsync(G);
    for (int q = tid; q < p; q += blockDim.x) L.g[q] = xld(&GV[q]);
    __syncthreads();
    double* Zh = xbuf(n, G);
    double* Zl = xbuf(n, G);
    ompl_mv_rows<OMPL_NB>(n, p, L.g, nullptr, Zh, Zl, nullptr);
    gsync(G);
    double dm1 = 0.0, xm1 = 0.0;
    int neg = 0;
    for (int q = tid; q < p; q += blockDim.x) {
      const double dz = xld(&Zh[q]), xq = L.z[q];
      dm1 = fmax(dm1, fabs(dz)); xm1 = fmax(xm1, fabs(xq));
      L.u[q] = xq + dz;
      if (!(xq + dz > 0.0)) neg = 1;
    }
    dm1 = block_allmax(dm1, scratch);
    xm1 = block_allmax(xm1, scratch);
    const double negs1 = block_allmax((double)neg, scratch);
    const bool bad = !G.ok || (p > 0 && (!(dm1 <= 1e-3 * xm1) || negs1 > 0.0));      // (the same in every workgroup)
    if (pass == 0) { fallback = bad; dm = dm1; xm = xm1; negs = negs1; }
    if (bad) break;
    for (int q = tid; q < p; q += blockDim.x) { L.z[q] = L.u[q]; L.x[L.cs[q]] = L.u[q]; }
    __syncthreads();
  }
  combine(a.tmp);                                            // V_P^T x of the weights that are committed
#ifdef BCX_OPT_PROFILE
  // every workgroup's own wait at the barrier after u = H g and its time in the phases between the barriers (ticks), into the
  // weight back-up buffer behind the k weights' slots that a revert would read (profile builds only; bcx_debug_wbak)
  if (tid == 0 && wg != 0 && wg < 256 && k >= 1024) {
    n.wbak[wg] = (double)prof_[3];
    n.wbak[256 + wg] = (double)(prof_[1] + prof_[2] + prof_[4] + prof_[5] + prof_[6] + prof_[8] + prof_[9]);
    n.wbak[512 + wg] = (double)prof_[7];
  }
#endif
  if (wg != 0) { if (G.ok) grid_arrive_at(G.gs, G.bi + 1); return; }
  gsync(G);
  // ---- workgroup 0: publish the passive data, new weights, accept / revert (snnls.py:88-97) ------------------------------
  if (tid == 0) {
    st->dbg_t[20] = warm_p ? 1 : 0; st->dbg_t[21] = dg_p0; st->dbg_t[22] = dg_p1; st->dbg_t[23] = dg_in; st->dbg_t[24] = dg_out;
    st->dbg_t[25] = p; st->dbg_t[26] = fallback; st->dbg_t[27] = (long long)(1e15 * (xm > 0.0 ? dm / xm : 0.0)); st->dbg_t[28] = (long long)negs;
    st->dbg_t[29] = dg_outer; st->dbg_t[30] = n_out;
#ifdef BCX_OPT_PROFILE
    for (int i = 0; i < 8; ++i) st->dbg_t[i] = prof_[i];
    st->dbg_t[19] = prof_[8]; st->dbg_t[31] = prof_[9];
#endif
    if (warm_p) for (int i = 0; i < 11; ++i) st->dbg_t[8 + i] = st->dbg_t[20 + i];       // (kept when a cold run follows)
  }
  if (!G.ok) { if (tid == 0) { st->hvalid = 0; st->halt = HALT_GRID_TIMEOUT; } return; }
  if (fallback) {
    if (tid == 0) { st->omp_mode = OMP_OPT_FALLBACK; st->hvalid = 0; }
    return;                                                  // weights and xw untouched: the host launches nnls_grid.hip's solve
  }
  for (int q = tid; q < p; q += blockDim.x) n.plist[q] = L.cs[q];
  for (int j = tid; j < k; j += blockDim.x) {
    n.ppos[j] = L.pos[j];
    n.x[j] = L.pos[j] >= 0 ? L.x[j] : 0.0;
    if (L.fl[j] & FLAG_INS) a.act_w[j] = (L.pos[j] >= 0) ? L.x[j] : 0.0;
  }
  for (int j = tid; j < d; j += blockDim.x) a.xw[j] = xld(&a.tmp[j]);
  if (tid == 0) st->np = p;
  __syncthreads();
  refresh_state(a, scratch, false);
  const double new_cost = st->err;
  if (new_cost > prev_cost * (1.0 + tol)) {                  // snnls.py:91-97
    for (int j = tid; j < k; j += blockDim.x) a.act_w[j] = n.wbak[j];
    for (int j = tid; j < d; j += blockDim.x) a.xw[j] = a.tmp[2 * (size_t)d + j];
    __syncthreads();
    refresh_state(a, scratch, false);
    if (tid == 0) { st->limit = 1; st->hvalid = 0; }
  } else if (tid == 0) {
    st->hvalid = warm_p ? 0 : 1;                             // (a warm start's H began as a plain-double inverse: the next OMP step re-solves)
    st->hlo_valid = warm_p ? 0 : 1;                          // H is the double-double inverse of gram[P, P]
    st->since_refresh = 0;
  }
}

// 1 = not applicable (LDS budget, dev knob BCX_OPT_GRID=1): the caller takes nnls_grid.hip's kernel
int bcx_launch_optimize_lh(bcx_solver* s, double tol, int k, const int32_t* warm_p) {
  if (bcx_dev_env("BCX_OPT_GRID") || bcx_dev_env("BCX_OPT_SINGLE")) return 1;
  // Small independent supports (k <= d, k <= 512) stay with nnls_grid.hip: all their columns join the passive set by
  // bordering, one barrier each and no dual passes (k = 400, d = 512: 3.9 ms against 7.8 here); from there on -- and for
  // every k > d, where that kernel has to take Lawson-Hanson's order with a refined solve per column -- this one is
  // faster (k = 999, d = 512: 12.7 against 28.3 ms; k = 1497, d = 1024: 45 against 191 ms).  BCX_OPT_LH=1 forces it (tests).
  if (!warm_p && k <= s->cfg.d && k <= 512 && !bcx_dev_env("BCX_OPT_LH")) return 1;
  const int kcap = (k + 1 + 63) / 64 * 64;
  const size_t lds = (size_t)kcap * (8 * sizeof(double) + 3 * sizeof(int));
  if (lds > OMPL_LDS_MAX) return 1;
  if (!s->grid_counter) BCX_HIP(hipMalloc((void**)&s->grid_counter, BCX_GRID_WORDS * sizeof(unsigned long long)));
  BCX_HIP(hipMemsetAsync(s->grid_counter, 0, BCX_GRID_WORDS * sizeof(unsigned long long), s->stream));
  s->grid_epoch = 0;
  NnlsArgs n;
  fill_nnls_args(s, n, nullptr);
  GridSync gs;
  gs.counter = s->grid_counter;
  gs.base = 0;
  gs.timeout_ticks = 1000000000LL;     // 10 s
  // The barriers of this kernel carry NO release / acquire fences (dev: BCX_GRID_FENCE=1 puts them back).  Everything that
  // crosses workgroups here is an exchange vector written write-through (sc1 stores, drained by every wave before its
  // workgroup arrives: grid_publish) and read with sc1 loads after the wait -- the hand-off form csrc/lrpost.hip and
  // csrc/persist.hip use -- and H is touched by its rows' owner waves only, so the fences order nothing that is read.  What
  // they cost is H: the release writes the XCD's dirty rows back (16 MB per pass over the inverse at k = 1024), the acquire
  // drops them from the L2 before the owner reads them again.  Measured (k = 1497, d = 1024, tools/optimize_ab.sh):
  // 64 workgroups x 1024 threads 36.3 ms with, 34.5 without; 128 x 512 26.7 with, 22.4 without (the phase clock: the barrier
  // after u = H g 16 -> 10 us for EVERY workgroup alike -- it is the barrier's own cost, not a straggler).  Repeated calls
  // give one outcome in both forms (tests/race_hunt.py, tests/race_hunt_warm.py).
  static const bool fence = bcx_dev_env("BCX_GRID_FENCE") != nullptr;
  gs.fences = fence ? 1 : 0;
  static const bool flat = bcx_dev_env("BCX_GRID_FLAT") != nullptr;    // dev: every workgroup polls the arrival counter (the form of the other kernels)
  gs.gen = flat ? nullptr : s->grid_counter + 16;
  static const int forced_wgs = bcx_dev_env("BCX_OPT_WGS") ? atoi(bcx_dev_env("BCX_OPT_WGS")) : 0;     // dev
  // Workgroup shape (round 6, tools/optimize_ab.sh, k = 1497, d = 1024, 311 columns enter and 311 leave): what bounds a pivot is
  // the number of dependent round trips of its passes over H, and the batches that cut them (OMPL_NB) need registers --
  // 1024 threads x 64 workgroups 35.6 ms, 512 x 64 31.0, 512 x 96 31.2, 512 x 128 26.7, 256 x 128 30.3, 256 x 256 30.7;
  // k = 999, d = 512: 1024 x 32 4.0 ms, 512 x 64 3.6, 512 x 32 3.8, 256 x 64 3.8.
  const int wgs = forced_wgs > 0 ? forced_wgs : (k <= 256 ? 16 : (k <= 768 ? 32 : (k <= 1024 ? 64 : OPTL_MAX_WGS)));
  static const int forced_thr = bcx_dev_env("BCX_OPT_THREADS") ? atoi(bcx_dev_env("BCX_OPT_THREADS")) : 0;   // dev: 256 / 512 / 1024
  const int threads = forced_thr > 0 ? forced_thr : (k <= 192 ? 256 : 512);
#define OPTL_LAUNCH(T)                                                                                                          \
  do {                                                                                                                          \
    if (lds > 48 * 1024) BCX_HIP(hipFuncSetAttribute((const void*)optimize_lh_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL(optimize_lh_kernel<T>, dim3(wgs), dim3(T), lds, s->stream, n, gs, tol, kcap, warm_p);                  \
  } while (0)
  if (threads == 256) OPTL_LAUNCH(256); else if (threads == 512) OPTL_LAUNCH(512); else OPTL_LAUNCH(1024);
#undef OPTL_LAUNCH
  BCX_HIP(hipGetLastError());
  s->grid_dirty = true;
  return BCX_OK;
}

int bcx_launch_omp_lh(bcx_solver* s, NnlsArgs& n, const ResolveArgs* fused) {
  static const int force = bcx_dev_env("BCX_OMP_FORCE_RESOLVE") ? atoi(bcx_dev_env("BCX_OMP_FORCE_RESOLVE")) : 0;   // tests: re-solve every N-th step
  const int64_t kub = s->k_ub;
  const int kcap = (int)((kub + 1 + 63) / 64 * 64);
  const int dpad = (s->cfg.d + 63) / 64 * 64;
  const size_t lds = (size_t)kcap * (7 * sizeof(double) + 3 * sizeof(int)) + 3 * (size_t)dpad * sizeof(double);
  if (lds > OMPL_LDS_MAX) return 1;
  if (lds > s->omp_lds_allowed) {
    BCX_HIP(hipFuncSetAttribute((const void*)omp_lh_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OMPL_LDS_MAX));
    BCX_HIP(hipFuncSetAttribute((const void*)omp_lh_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OMPL_LDS_MAX));
    BCX_HIP(hipFuncSetAttribute((const void*)omp_lh_kernel<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)OMPL_LDS_MAX));
    s->omp_lds_allowed = OMPL_LDS_MAX;
  }
  if (s->grid_dirty) {                 // optimize() ran since the last bcx_build_begin: arrivals and base restart together
    BCX_HIP(hipMemsetAsync(s->grid_counter, 0, 2 * sizeof(unsigned long long), s->stream));
    s->grid_dirty = false;
  }
  GridSync gs;
  gs.counter = s->grid_counter;
  gs.base = 0;                         // (read from device memory by the kernel: s->grid_counter[1])
  gs.timeout_ticks = 1000000000LL;     // 10 s
  // dev: BCX_GRID_NOFENCE=1 drops the barriers' release / acquire fences (every cross-workgroup datum of this file is an
  // exchange vector written write-through and read with sc1 loads, so they order nothing that is read).  Measured round 6:
  // the OMP step of configs[2] takes 35.9 us either way (16 workgroups, an inverse of a few hundred KB), so here they stay;
  // optimize_lh_kernel above, whose inverse is 16 MB, runs without them.
  static const bool nofence = bcx_dev_env("BCX_GRID_NOFENCE") != nullptr;
  gs.fences = nofence ? 0 : 1;
  gs.gen = nullptr;                    // (16 workgroups poll the arrival counter itself)
  s->grid_epoch += 1;
  const int fr = (force > 0 && (s->grid_epoch % force) == 0) ? 1 : 0;
  // Workgroup width.  The step is a chain of short latency-bound phases separated by workgroup barriers and block
  // reductions, whose cost grows with the number of waves; the parallel work (one wave per slot row / per row of H) is
  // 16 workgroups x waves.  Measured on the configs[2] vectors (tools/omp_hist.py, k <= 140): closed-form step 43.6 us
  // with 1024 threads, 30.2 with 512, 28.8 with 256.  Wider workgroups once a wave would own more than ~3 rows.
  static const int forced_threads = bcx_dev_env("BCX_OMP_THREADS") ? atoi(bcx_dev_env("BCX_OMP_THREADS")) : 0;   // dev: 256 / 512 / 1024
  const int threads = forced_threads ? forced_threads : (kub <= 192 ? 256 : (kub <= 448 ? 512 : NN_THREADS));
  ResolveArgs rsv;
  if (fused) rsv = *fused; else memset(&rsv, 0, sizeof(rsv));
  if (threads != 256 && threads != 512 && threads != 1024) { s->err = "BCX_OMP_THREADS must be 256, 512 or 1024"; return BCX_ERR_ARG; }
  if (threads == 256)
    hipLaunchKernelGGL(omp_lh_kernel<256>, dim3(OMPL_WGS), dim3(256), lds, s->stream, n, gs, s->grid_counter + 1, kcap, dpad, fr, fused ? 1 : 0, rsv);
  else if (threads == 512)
    hipLaunchKernelGGL(omp_lh_kernel<512>, dim3(OMPL_WGS), dim3(512), lds, s->stream, n, gs, s->grid_counter + 1, kcap, dpad, fr, fused ? 1 : 0, rsv);
  else
    hipLaunchKernelGGL(omp_lh_kernel<1024>, dim3(OMPL_WGS), dim3(1024), lds, s->stream, n, gs, s->grid_counter + 1, kcap, dpad, fr, fused ? 1 : 0, rsv);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}
