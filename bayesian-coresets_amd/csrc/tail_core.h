// tail_core.h -- the replicated O(d) state machine of a GIGA / Frank-Wolfe iteration after the winner is known (apply_core) and
// what its kernels fetch before the scan's partials arrive: shared by resolve.hip (one launch per iteration) and persist.hip
// (the tail workgroup of a launch that covers several iterations).
#pragma once
#include "bcx_internal.h"
#include "dev_util.h"
#include "apply_common.h"
#include "resolve_core.h"

// ------------------------------------------------------------------------------------------
// apply (GIGA / Frank-Wolfe); OMP lives in nnls.hip
// ------------------------------------------------------------------------------------------
struct StateVecs {   // LDS copies, d doubles each
  double *xw, *b, *bn, *xf, *tx;
};

// BIG (5 d doubles beyond the LDS budget, d > BCX_LDS_VEC_MAX_D): the five vectors live in the solver's global scratch
// (a.tmp + 8 d ...) instead -- one workgroup on one CU: its waves share the L1, and __syncthreads() orders the accesses
template <bool BIG> static __device__ __forceinline__ double* vec_base(const ApplyArgs& a, double* dyn) {
  return BIG ? a.tmp + 8 * (size_t)a.d : dyn;
}
static __device__ __forceinline__ StateVecs carve(double* dyn, int d) {
  StateVecs v;
  v.xw = dyn; v.b = dyn + d; v.bn = dyn + 2 * (size_t)d; v.xf = dyn + 3 * (size_t)d; v.tx = dyn + 4 * (size_t)d;
  return v;
}

static __device__ __forceinline__ void stage_state(const ApplyArgs& a, const StateVecs& v) {
  for (int j = threadIdx.x; j < a.d; j += blockDim.x) {
    v.xw[j] = a.xw[j];
    v.b[j] = a.b[j];
    v.bn[j] = a.bn[j];
  }
}

// Reweight with the winner (f, nf, xf in LDS), commit or fail, trace, next query.
// The sparse weight list (slot -> global row, weight) does not depend on the scan: the kernels fetch it into
// registers at entry, so the look-up of the winner's slot costs no memory round trip after the winner is known.
#define SLOT_PRE 4
struct SlotPre {
  long long id[SLOT_PRE];
  double w[SLOT_PRE];
  int k;
  // the scalars of the replicated state the step needs, fetched in the same early round trip
  double nw, err0, bnorm, sigma, tol;
  int64_t it, itrs;
  int retried, since;
};
static __device__ __forceinline__ SlotPre slot_prefetch(const ApplyArgs& a) {
  SlotPre p;
  const DevState* st = a.st;
  p.k = st->k;
  p.nw = st->nw; p.err0 = st->err; p.bnorm = st->bnorm; p.sigma = st->sigma; p.tol = st->tol;
  p.it = st->it; p.itrs = st->itrs; p.retried = st->retried; p.since = st->since_refresh;
#pragma unroll
  for (int t = 0; t < SLOT_PRE; ++t) {
    const int s = threadIdx.x + t * blockDim.x;
    const bool ok = s < p.k;
    p.id[t] = ok ? a.act_idx[s] : -1;
    p.w[t] = ok ? a.act_w[s] : 0.0;
  }
  return p;
}

template <int ALG>
static __device__ void apply_core(const ApplyArgs& a, const StateVecs& v, int64_t f, double nf, double* scratch,
                                  const SlotPre& pre) {
  DevState* st = a.st;
  __shared__ int s_slot, s_npos;
  __shared__ double s_wf;
  const int tid = threadIdx.x, d = a.d;
  const int k = pre.k;
  const double nw = pre.nw, err0 = pre.err0, bnorm = pre.bnorm, sigma = pre.sigma, tol = pre.tol;
  const int64_t it = pre.it, itrs = pre.itrs;
  const int retried = pre.retried, since = pre.since;
  if (tid == 0) { s_slot = 0x7fffffff; s_npos = 0; }
  __syncthreads();
  // slot of f in the sparse weight list, and size() > 0  (snnls.py:44)
  {
    int npos = 0, slot = 0x7fffffff;
    double wf = 0.0;
    if (k <= SLOT_PRE * (int)blockDim.x) {
#pragma unroll
      for (int t = 0; t < SLOT_PRE; ++t) {
        if (pre.id[t] == f && pre.id[t] >= 0) { slot = tid + t * blockDim.x; wf = pre.w[t]; }
        if (pre.w[t] > 0.0) ++npos;
      }
    } else {
      for (int s = tid; s < k; s += blockDim.x) {
        const int64_t id = a.act_idx[s];
        const double w = a.act_w[s];
        if (id == f) { slot = s; wf = w; }
        if (w > 0.0) ++npos;
      }
    }
    if (slot != 0x7fffffff) { atomicMin(&s_slot, slot); s_wf = wf; }   // global rows are unique among the slots
    // (one LDS atomic per wave: 256 same-address atomics serialise for ~1 us)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) npos += __shfl_xor(npos, off, BCX_WAVE);
    if ((tid & 63) == 0 && npos) atomicAdd(&s_npos, npos);
  }
  __syncthreads();
  BCX_STAMP(st, 5);
  const bool checked = s_npos > 0;
  const int slot = s_slot == 0x7fffffff ? -1 : s_slot;
  const double wf_old = slot >= 0 ? s_wf : 0.0;
  int status = BCX_IT_OK;
  double alpha = 0.0, beta = 0.0;
  if (ALG == BCX_ALG_GIGA) {
    // giga.py:42-61
    double r[3] = {0.0, 0.0, 0.0};
    for (int j = tid; j < d; j += blockDim.x) {
      const double xh = v.xw[j] / nw, fh = v.xf[j] / nf, bj = v.bn[j];
      r[0] += bj * fh; r[1] += bj * xh; r[2] += xh * fh;
    }
    block_allsum<3>(r, scratch);
    const double gA = r[0] - r[1] * r[2];
    const double gB = r[1] - r[0] * r[2];
    if (gA <= 0.0 || gB < 0.0) {
      status = BCX_IT_FAIL_REWEIGHT;
    } else {
      const double ca = gB / (gA + gB) / nw;
      const double cb = gA / (gA + gB) / nf;
      double u[2] = {0.0, 0.0};
      for (int j = tid; j < d; j += blockDim.x) {
        const double x = ca * v.xw[j] + cb * v.xf[j];
        u[0] += x * x;
        u[1] += x * v.bn[j];
      }
      block_allsum<2>(u, scratch);
      const double nx = sqrt(u[0]);
      const double scale = bnorm / nx * (u[1] / nx);     // giga.py:58
      alpha = ca * scale; beta = cb * scale;
    }
  } else {
    // frankwolfe.py:19-37
    if (!checked) {
      alpha = 0.0; beta = sigma / nf;
    } else {
      const double sc = sigma / nf;
      double r[2] = {0.0, 0.0};
      for (int j = tid; j < d; j += blockDim.x) {
        const double vv = sc * v.xf[j] - v.xw[j];
        r[0] += vv * (v.b[j] - v.xw[j]);
        r[1] += vv * vv;
      }
      block_allsum<2>(r, scratch);
      const double gnum = r[0], gden = r[1];
      if (gnum < 0.0 || gden == 0.0 || gnum > gden) status = BCX_IT_FAIL_REWEIGHT;
      else { alpha = 1.0 - gnum / gden; beta = sc * gnum / gden; }
    }
  }
  BCX_STAMP(st, 6);
  double new_err = err0, new_nw = nw, tdot = 0.0, wf_new = 0.0;
  if (status == BCX_IT_OK) {
    // w <- alpha*w ; w[f] <- max(0, w[f] + beta)   giga.py:63-64 / frankwolfe.py:39-40
    const double wf_scaled = alpha * wf_old;
    wf_new = fmax(0.0, wf_scaled + beta);
    const double delta = wf_new - wf_scaled;
    double r[3] = {0.0, 0.0, 0.0};
    for (int j = tid; j < d; j += blockDim.x) {
      const double x = alpha * v.xw[j] + delta * v.xf[j];   // = A w'
      v.tx[j] = x;
      const double e = x - v.b[j];
      r[0] += e * e; r[1] += x * x;
      if (ALG == BCX_ALG_GIGA) r[2] += v.bn[j] * x;
    }
    block_allsum<3>(r, scratch);
    new_err = sqrt(r[0]);
    const double n2 = sqrt(r[1]);
    new_nw = n2 == 0.0 ? 1.0 : n2;
    tdot = r[2];
    if (checked && !st->no_monotone && new_err > err0) status = BCX_IT_FAIL_MONOTONE;   // snnls.py:56-58
  }
  BCX_STAMP(st, 7);
  bool limit = false;
  if (status == BCX_IT_OK) {
    for (int s = tid; s < k; s += blockDim.x)
      if (s != slot) a.act_w[s] = alpha * a.act_w[s];
    for (int j = tid; j < d; j += blockDim.x) { const double x = v.tx[j]; a.xw[j] = x; v.xw[j] = x; }
    const int dst = slot >= 0 ? slot : k;
    if (slot < 0)
      for (int j = tid; j < d; j += blockDim.x) a.act_rows[(size_t)dst * d + j] = v.xf[j];
    if (tid == 0) {
      a.act_w[dst] = wf_new;
      if (slot < 0) { a.act_idx[dst] = f; a.act_norm[dst] = nf; st->k = k + 1; }
      st->err = new_err;
      st->nw = new_nw;
      st->since_refresh = since + 1;
      if (checked && !st->no_monotone) st->retried = 0;    // snnls.py:62 (inside the monotone-check branch)
    }
  } else {
    limit = retried != 0;                                  // snnls.py:63-72
  }
  if (tid == 0) {
    a.tr_sel[it] = f; a.tr_err[it] = (status == BCX_IT_OK) ? new_err : err0; a.tr_status[it] = status;
    st->it = it + 1;
    st->exact_mode = 0;
    if (status != BCX_IT_OK) {
      if (limit) { st->limit = 1; st->active = 0; st->halt = HALT_LIMIT; }
      else st->retried = 1;
    }
  }
  BCX_STAMP(st, 8);
  if (limit) return;
  // ---- next query ----
  const bool fast = status == BCX_IT_OK && it + 1 < itrs &&
                    !(a.refresh_every > 0 && since + 1 >= a.refresh_every);
  if (fast) {
    if (ALG != BCX_ALG_GIGA) {
      for (int j = tid; j < d; j += blockDim.x) store_query(a, 0, j, v.b[j] - v.xw[j]);   // frankwolfe.py:16
      if (tid == 0) st->qscale = new_err;
      BCX_STAMP(st, 9);
      return;
    }
    // giga.py:21-30 on the new iterate
    const double t = tdot / new_nw;
    double c2[1] = {0.0};
    for (int j = tid; j < d; j += blockDim.x) {
      const double c = v.bn[j] - t * (v.xw[j] / new_nw);
      v.tx[j] = c;
      c2[0] += c * c;
    }
    block_allsum<1>(c2, scratch);
    const double cn = sqrt(c2[0]);
    if (!(cn < tol)) {
      for (int j = tid; j < d; j += blockDim.x) {
        store_query(a, 0, j, v.tx[j] / cn);
        store_query(a, 1, j, v.xw[j] / new_nw);
      }
      if (tid == 0) st->qscale = 1.0;
      BCX_STAMP(st, 9);
      return;
    }
  }
  // slow path: end of call, refresh due, failed step or failing select -- generic state machine
  __syncthreads();
  prepare_next(a, scratch);
}
