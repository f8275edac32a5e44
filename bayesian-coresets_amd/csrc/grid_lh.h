// grid_lh.h -- primitives of the multi-workgroup active-set solves (optimize(): nnls_grid.hip; the OMP step: omp_lh.hip):
// grid barriers over an arrival counter, rotating exchange vectors, the replicated LDS state and the owner-computes
// passes over the inverse H of the passive Gram block.
#pragma once
#include <hip/hip_runtime.h>
#include "nnls_common.h"

// Cross-XCD rule of this kernel (per-XCD L2s are not coherent with each other, and an acquire fence drops the
// reader's L1 but not a stale line in its XCD's L2):
//   * H is OWNER-COMPUTES: row rr of the inverse is read and written only by wave (rr mod 16*G) -- the mat-vec
//     y = H v takes whole rows (H is symmetric), the rank-1 up/down-dates touch own rows, a new or moved row /
//     column entry is written by the row's owner.  No line of H is ever shared between workgroups.
//   * everything that does cross workgroups is a small EXCHANGE vector, written with write-through (sc1) stores
//     and read with sc1 loads after a grid barrier; consecutive exchanges rotate through GRID_RING buffers, so a
//     buffer is rewritten only several barriers after it was read (see GRID_RING) and no trailing barrier is needed.
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
#define GRID_RING 16  // exchange buffers (gram_cap doubles each, contiguous from n.t0).  The busiest user, ompl_remove of
                      // omp_lh.hip, takes FOUR buffers per barrier (hi and lo words of two rows): back-to-back removals come
                      // back to a buffer after 4 barriers.  (A reuse distance of 2 barriers is already ordered -- the reads of
                      // an exchange precede the reader's arrival at the next barrier -- the rest is margin.)
static __device__ __forceinline__ double* xbuf(const NnlsArgs& n, Grid& g) {   // next exchange buffer (gram_cap doubles)
  return n.t0 + (size_t)((g.xi++) & (GRID_RING - 1)) * (size_t)n.ldg;
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
static __device__ __forceinline__ void g_mv_rows(const NnlsArgs& n, int p, const double* v, double* out) {
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
static __device__ __forceinline__ bool g_border_add(const NnlsArgs& n, const Rep& r, int& p, int& ill, int slot, Grid& g,
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
static __device__ __forceinline__ void g_border_del(const NnlsArgs& n, const Rep& r, int& p, int q, Grid& g) {
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
static __device__ __forceinline__ void g_passive_solve(const NnlsArgs& n, const Rep& r, int p, int ill, Grid& g, double (*seg)[64],
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
