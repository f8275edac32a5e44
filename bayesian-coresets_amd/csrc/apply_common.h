// apply_common.h -- pieces of the replicated O(d) state machine shared by the GIGA/FW apply
// kernel (resolve.hip) and the OMP / optimize kernels (nnls.hip).
#pragma once
#include "bcx_internal.h"
#include "dev_util.h"

struct ApplyArgs {
  DevState* st;
  const double* recs;   // world_size records of (d + 4) doubles
  int world;
  int d, ld, ld64;
  int alg;
  int store_f64;
  int refresh_every;
  const double* b;
  const double* bn;
  double* xw;
  double* q64;
  void* qst;
  double* tmp;          // >= 2*d scratch
  int64_t cap;
  int64_t* act_idx;
  double* act_w;
  double* act_rows;
  double* act_norm;
  int64_t* tr_sel;
  double* tr_err;
  int32_t* tr_status;
};

static __device__ __forceinline__ void store_query(const ApplyArgs& a, int which, int j, double v) {
  a.q64[(size_t)which * a.ld64 + j] = v;
  if (a.store_f64) ((double*)a.qst)[(size_t)which * a.ld + j] = v;
  else ((float*)a.qst)[(size_t)which * a.ld + j] = (float)v;
}

// Query for the next select.  Returns false when the reference's _select would raise
// (giga.py:28-29).  All threads must call; result is uniform.
static __device__ bool prepare_query(const ApplyArgs& a, double* scratch) {
  DevState* st = a.st;
  const int d = a.d;
  if (a.alg != BCX_ALG_GIGA) {
    // residual = b - A w    frankwolfe.py:16 / orthopursuit.py:18
    for (int j = threadIdx.x; j < d; j += blockDim.x) store_query(a, 0, j, a.b[j] - a.xw[j]);
    if (threadIdx.x == 0) st->qscale = st->err;
    return true;
  }
  const double nw = st->nw;
  double v[1] = {0.0};
  for (int j = threadIdx.x; j < d; j += blockDim.x) v[0] += a.bn[j] * (a.xw[j] / nw);   // giga.py:26
  block_allsum<1>(v, scratch);
  const double t = v[0];
  double c2[1] = {0.0};
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    const double xh = a.xw[j] / nw;
    const double c = a.bn[j] - t * xh;
    a.tmp[j] = c;
    c2[0] += c * c;
  }
  block_allsum<1>(c2, scratch);
  const double cn = sqrt(c2[0]);
  if (cn < st->tol) return false;                                                         // giga.py:28
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    store_query(a, 0, j, a.tmp[j] / cn);
    store_query(a, 1, j, a.xw[j] / nw);
  }
  if (threadIdx.x == 0) st->qscale = 1.0;
  return true;
}

// Recompute xw = sum_j w_j A[:,j] over the slots (snnls.py:29 recomputes A.dot(w) on every call;
// the engine does so periodically to keep the incrementally updated xw at fresh-sum accuracy),
// then err and nw.  All threads.
static __device__ void refresh_state(const ApplyArgs& a, double* scratch, bool recompute_xw) {
  DevState* st = a.st;
  const int d = a.d, k = st->k;
  double v[2] = {0.0, 0.0};
  for (int j = threadIdx.x; j < d; j += blockDim.x) {
    double x;
    if (recompute_xw) {
      // fixed slot order; 8 independent loads per step so the L2 latencies overlap
      x = 0.0;
      int s = 0;
      for (; s + 8 <= k; s += 8) {
        double r[8], w[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) { w[t] = a.act_w[s + t]; r[t] = a.act_rows[(size_t)(s + t) * d + j]; }
#pragma unroll
        for (int t = 0; t < 8; ++t) x += w[t] * r[t];
      }
      for (; s < k; ++s) x += a.act_w[s] * a.act_rows[(size_t)s * d + j];
      a.xw[j] = x;
    } else {
      x = a.xw[j];
    }
    const double r = x - a.b[j];
    v[0] += r * r;
    v[1] += x * x;
  }
  block_allsum<2>(v, scratch);
  if (threadIdx.x == 0) {
    st->err = sqrt(v[0]);
    const double nw = sqrt(v[1]);
    st->nw = (nw == 0.0) ? 1.0 : nw;
    if (recompute_xw) st->since_refresh = 0;
  }
  __syncthreads();
}

// Loop tail shared by begin/apply: consume failed-select iterations, stop at the end of the call.
// (snnls.py:41,63-74).  All threads; returns with st->active decided.
static __device__ void prepare_next(const ApplyArgs& a, double* scratch) {
  DevState* st = a.st;
  __shared__ int go;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) {
      go = 1;
      if (st->it >= st->itrs) { st->active = 0; st->halt = HALT_DONE; go = 0; }
    }
    __syncthreads();
    if (!go) return;
    if (a.refresh_every > 0 && st->since_refresh >= a.refresh_every) refresh_state(a, scratch, true);
    const bool ok = prepare_query(a, scratch);
    if (ok) return;
    __syncthreads();
    if (threadIdx.x == 0) {
      const int64_t it = st->it;
      a.tr_sel[it] = -1; a.tr_err[it] = st->err; a.tr_status[it] = BCX_IT_FAIL_SELECT;
      st->it = it + 1;
      if (st->retried) { st->limit = 1; st->active = 0; st->halt = HALT_LIMIT; go = 0; }
      else st->retried = 1;
    }
    __syncthreads();
    if (!go) return;
  }
}


static inline void fill_apply_args(bcx_solver* s, ApplyArgs& a, const double* recs) {
  a.st = s->st;
  a.recs = recs;
  a.world = s->cfg.world_size;
  a.d = s->cfg.d; a.ld = s->ld; a.ld64 = s->ld64;
  a.alg = s->cfg.alg;
  a.store_f64 = s->cfg.store_dtype == BCX_F64;
  a.refresh_every = s->cfg.refresh_every;
  a.b = s->b; a.bn = s->bn; a.xw = s->xw; a.q64 = s->q64; a.qst = s->qst; a.tmp = s->tmp;
  a.cap = s->cap;
  a.act_idx = s->act_idx; a.act_w = s->act_w; a.act_rows = s->act_rows; a.act_norm = s->act_norm;
  a.tr_sel = s->tr_sel; a.tr_err = s->tr_err; a.tr_status = s->tr_status;
}

