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

#include "grid_lh.h"

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
    st->hlo_valid = 0;
    st->since_refresh = 0;
  }
}

int bcx_launch_optimize_grid(bcx_solver* s, double tol, int k) {
  if (k >= OPT_MAX_K || bcx_dev_env("BCX_OPT_SINGLE")) return 1;   // caller uses the single-workgroup kernel
  if (!s->grid_counter) {
    BCX_HIP(hipMalloc((void**)&s->grid_counter, BCX_GRID_WORDS * sizeof(unsigned long long)));
  }
  BCX_HIP(hipMemsetAsync(s->grid_counter, 0, 2 * sizeof(unsigned long long), s->stream));   // [0] arrivals, [1] the OMP step's barrier base
  s->grid_epoch = 0;
  NnlsArgs n;
  fill_nnls_args(s, n, nullptr);
  const int kcap = (k + 1 + 63) / 64 * 64;
  const int dpad = (s->cfg.d + 63) / 64 * 64;
  const size_t lds = (size_t)kcap * (4 * sizeof(double) + 3 * sizeof(int)) + (size_t)dpad * sizeof(double);
  if (lds > 150 * 1024) return 1;      // (long rows with a large support: the single-workgroup kernel)
  if (lds > 48 * 1024)
    BCX_HIP(hipFuncSetAttribute((const void*)optimize_grid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  GridSync gs;
  gs.counter = s->grid_counter;
  gs.base = 0;
  gs.timeout_ticks = 1000000000LL;   // 10 s
  gs.fences = 1;
  gs.gen = nullptr;
  hipLaunchKernelGGL(optimize_grid_kernel, dim3(OPT_WGS), dim3(NN_THREADS), lds, s->stream, n, gs, tol, kcap, dpad);
  BCX_HIP(hipGetLastError());
  s->grid_dirty = true;   // counter[0] now holds this launch's arrivals; an OMP step that follows without a build_begin re-zeroes
  return BCX_OK;
}
