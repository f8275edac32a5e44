// resolve.hip -- the O(d) fp64 part of a greedy iteration, single-workgroup kernels:
//   resolve_kernel : scan partials -> candidate rows -> exact fp64 re-score -> this shard's record
//   apply_kernel   : winner over all shards' records, reweight, monotone check / revert /
//                    retry / latch (snnls.py:41-74), next query vector
//   begin_kernel   : start of a build() call (snnls.py:31-40) + first query
// All state is replicated: every shard runs the same code on the same gathered records, so
// xw, weights and the trace stay bit-identical across shards.
#include "bcx_internal.h"
#include "dev_util.h"
#include "apply_common.h"

// ------------------------------------------------------------------------------------------
// resolve
// ------------------------------------------------------------------------------------------
struct ResolveArgs {
  const ScanPartial* partials;
  int n_partials;
  DevState* st;
  const void* An;
  int store_f64;
  int ld;
  const double* A64;
  int ld64;
  const double* norms;
  const double* q64;
  int d;
  int alg;
  int64_t n_local;
  int64_t row_offset;
  int exact;      // take the arg-max of the partials as is (fp64 scan, or fallback without raw rows)
  double* rec;    // out: d + 4 doubles
};

__device__ __forceinline__ double giga_score64(double s0, double s1) {
  const bool ok = (s1 > -1.0 + 1e-14) && (1.0 - s1 * s1 > 0.0);   // giga.py:33
  const double den = ok ? sqrt(1.0 - s1 * s1) : INFINITY;          // giga.py:35-36
  return s0 / den;                                                 // giga.py:38
}

__device__ __forceinline__ double row_elem(const ResolveArgs& a, int64_t i, int j, double nrm) {
  // normalised element An[i][j] in fp64: raw/norm when the raw rows are resident (giga.py:13)
  if (a.A64) return a.A64[i * (int64_t)a.ld64 + j] / nrm;
  if (a.store_f64) return ((const double*)a.An)[i * (int64_t)a.ld + j];
  return (double)((const float*)a.An)[i * (int64_t)a.ld + j];
}

__global__ __launch_bounds__(256) void resolve_kernel(ResolveArgs a) {
  if (!a.st->active) return;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ int cand[BCX_MAX_CAND];
  __shared__ double cscore[BCX_MAX_CAND];
  __shared__ int ncand, overflow, winner;
  __shared__ double wscore;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  if (tid == 0) { ncand = 0; overflow = 0; winner = -1; wscore = -INFINITY; }
  __syncthreads();
  const bool exact = a.exact || a.st->exact_mode;
  if (exact) {
    // arg-max of the per-workgroup maxima, lowest index on ties
    double bu = -INFINITY; int bi = 0x7fffffff;
    for (int p = tid; p < a.n_partials; p += blockDim.x) {
      const ScanPartial sp = a.partials[p];
      if (sp.U1 > bu || (sp.U1 == bu && sp.i1 < bi)) { bu = sp.U1; bi = sp.i1; }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double ou = __shfl_xor(bu, off, BCX_WAVE); const int oi = __shfl_xor(bi, off, BCX_WAVE);
      if (ou > bu || (ou == bu && oi < bi)) { bu = ou; bi = oi; }
    }
    __shared__ double wu[4]; __shared__ int wi[4];
    if (lane == 0) { wu[wave] = bu; wi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 0; w < nwaves; ++w)
        if (wu[w] > bu || (wu[w] == bu && wi[w] < bi)) { bu = wu[w]; bi = wi[w]; }
      if (bi != 0x7fffffff && bu > -INFINITY) { cand[0] = bi; ncand = 1; }
    }
    __syncthreads();
  } else {
    double lmax = -INFINITY;
    for (int p = tid; p < a.n_partials; p += blockDim.x) lmax = fmax(lmax, a.partials[p].L);
    const double Lstar = block_allmax(lmax, scratch);
    for (int p = tid; p < a.n_partials; p += blockDim.x) {
      const ScanPartial sp = a.partials[p];
      if (sp.U1 > -INFINITY && sp.U1 >= Lstar) {
        const int slot = atomicAdd(&ncand, 1);
        if (slot < BCX_MAX_CAND) cand[slot] = sp.i1;
      }
      if (sp.U2 > -INFINITY && sp.U2 >= Lstar) {
        const int slot = atomicAdd(&ncand, 1);
        if (slot < BCX_MAX_CAND) cand[slot] = sp.i2;
      }
      if (sp.U3 > -INFINITY && sp.U3 >= Lstar) overflow = 1;
    }
    __syncthreads();
    if (ncand > BCX_MAX_CAND) overflow = 1;
    __syncthreads();
  }
  double* rec = a.rec;
  if (overflow) {
    if (tid == 0) { rec[0] = -INFINITY; rec[1] = -1.0; rec[2] = 0.0; rec[3] = BCX_REC_OVERFLOW; }
    return;
  }
  const int nc = ncand;
  if (nc == 0) {
    if (tid == 0) { rec[0] = -INFINITY; rec[1] = -1.0; rec[2] = 0.0; rec[3] = 0.0; }
    return;
  }
  // exact fp64 score of every candidate: one wave per candidate
  const double* q0 = a.q64;
  const double* q1 = a.q64 + a.ld64;
  for (int c = wave; c < nc; c += nwaves) {
    const int64_t i = cand[c];
    const double nrm = a.norms[i];
    double s0 = 0.0, s1 = 0.0;
    for (int j = lane; j < a.d; j += 64) {
      const double v = row_elem(a, i, j, nrm);
      s0 += v * q0[j];
      if (a.alg == BCX_ALG_GIGA) s1 += v * q1[j];
    }
    s0 = wave_allsum(s0);
    if (a.alg == BCX_ALG_GIGA) s1 = wave_allsum(s1);
    if (lane == 0) cscore[c] = (a.alg == BCX_ALG_GIGA) ? giga_score64(s0, s1) : s0;
  }
  __syncthreads();
  if (tid == 0) {
    int best = 0;
    for (int c = 1; c < nc; ++c)
      if (cscore[c] > cscore[best] || (cscore[c] == cscore[best] && cand[c] < cand[best])) best = c;
    // NaN scores never win a '>' comparison; if candidate 0 is NaN and another is not, prefer the other
    if (cscore[best] != cscore[best])
      for (int c = 0; c < nc; ++c) if (cscore[c] == cscore[c]) { best = c; break; }
    winner = cand[best];
    wscore = cscore[best];
  }
  __syncthreads();
  const int64_t wrow = winner;
  const double nrm = a.norms[wrow];
  if (tid == 0) {
    rec[0] = wscore;
    rec[1] = (double)(a.row_offset + wrow);
    rec[2] = nrm;
    rec[3] = BCX_REC_VALID;
  }
  for (int j = tid; j < a.d; j += blockDim.x) {
    double raw;
    if (a.A64) raw = a.A64[wrow * (int64_t)a.ld64 + j];
    else if (a.store_f64) raw = ((const double*)a.An)[wrow * (int64_t)a.ld + j] * nrm;
    else raw = (double)((const float*)a.An)[wrow * (int64_t)a.ld + j] * nrm;
    rec[BCX_REC_HDR + j] = raw;
  }
}

int bcx_launch_resolve(bcx_solver* s, double* send_dev, int exact) {
  ResolveArgs a;
  a.partials = s->partials;
  a.n_partials = s->n_partials;
  a.st = s->st;
  a.An = s->An;
  a.store_f64 = s->cfg.store_dtype == BCX_F64;
  a.ld = s->ld;
  a.A64 = s->A64;
  a.ld64 = s->ld64;
  a.norms = s->norms;
  a.q64 = s->q64;
  a.d = s->cfg.d;
  a.alg = s->cfg.alg;
  a.n_local = s->cfg.n_local;
  a.row_offset = s->cfg.row_offset;
  a.exact = exact || a.store_f64;
  a.rec = send_dev;
  hipLaunchKernelGGL(resolve_kernel, dim3(1), dim3(256), 0, s->stream, a);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

// ------------------------------------------------------------------------------------------
// apply (GIGA / Frank-Wolfe); OMP lives in nnls.hip and reuses the helpers below
// ------------------------------------------------------------------------------------------
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
  if (threadIdx.x == 0 && st->halt == HALT_NEED_EXACT) { st->active = 1; st->halt = HALT_NONE; st->exact_mode = 1; }
}

template <int ALG>
__global__ __launch_bounds__(BCX_APPLY_THREADS) void apply_kernel(ApplyArgs a) {
  DevState* st = a.st;
  if (!st->active) return;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ int s_win, s_overflow, s_slot, s_npos, s_status;
  __shared__ double s_alpha, s_beta;
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
    s_win = win; s_overflow = ovf; s_slot = 0x7fffffff; s_npos = 0; s_status = BCX_IT_OK;
  }
  __syncthreads();
  if (s_overflow) {
    if (tid == 0) { st->active = 0; st->halt = HALT_NEED_EXACT; }
    return;
  }
  if (s_win < 0) {  // no data anywhere: nothing to select (snnls.py:36-38 is handled by the host)
    if (tid == 0) { st->active = 0; st->halt = HALT_DONE; }
    return;
  }
  const double* rec = a.recs + (size_t)s_win * recw;
  const int64_t f = (int64_t)rec[1];
  const double nf = rec[2];
  const double* xf = rec + BCX_REC_HDR;
  const int k = st->k;
  // slot of f in the sparse weight list, and size() > 0  (snnls.py:44)
  int npos = 0;
  for (int s = tid; s < k; s += blockDim.x) {
    if (a.act_idx[s] == f) atomicMin(&s_slot, s);
    if (a.act_w[s] > 0.0) ++npos;
  }
  if (npos) atomicAdd(&s_npos, npos);
  __syncthreads();
  const bool checked = s_npos > 0;
  const int slot = s_slot == 0x7fffffff ? -1 : s_slot;
  const double wf_old = slot >= 0 ? a.act_w[slot] : 0.0;
  const double nw = st->nw;

  if (ALG == BCX_ALG_GIGA) {
    // giga.py:42-61
    double v[3] = {0.0, 0.0, 0.0};
    for (int j = tid; j < d; j += blockDim.x) {
      const double xh = a.xw[j] / nw, fh = xf[j] / nf, bj = a.bn[j];
      v[0] += bj * fh; v[1] += bj * xh; v[2] += xh * fh;
    }
    block_allsum<3>(v, scratch);
    const double gA = v[0] - v[1] * v[2];
    const double gB = v[1] - v[0] * v[2];
    if (gA <= 0.0 || gB < 0.0) {
      if (tid == 0) s_status = BCX_IT_FAIL_REWEIGHT;
    } else {
      const double ca = gB / (gA + gB) / nw;
      const double cb = gA / (gA + gB) / nf;
      double u0[1] = {0.0}, u1[1] = {0.0};
      for (int j = tid; j < d; j += blockDim.x) {
        const double x = ca * a.xw[j] + cb * xf[j];
        a.tmp[d + j] = x;
        u0[0] += x * x;
      }
      block_allsum<1>(u0, scratch);
      const double nx = sqrt(u0[0]);
      for (int j = tid; j < d; j += blockDim.x) u1[0] += (a.tmp[d + j] / nx) * a.bn[j];
      block_allsum<1>(u1, scratch);
      const double scale = st->bnorm / nx * u1[0];
      if (tid == 0) { s_alpha = ca * scale; s_beta = cb * scale; }
    }
  } else {
    // frankwolfe.py:19-37
    if (!checked) {
      if (tid == 0) { s_alpha = 0.0; s_beta = st->sigma / nf; }
    } else {
      const double sc = st->sigma / nf;
      double v[2] = {0.0, 0.0};
      for (int j = tid; j < d; j += blockDim.x) {
        const double vv = sc * xf[j] - a.xw[j];
        v[0] += vv * (a.b[j] - a.xw[j]);
        v[1] += vv * vv;
      }
      block_allsum<2>(v, scratch);
      const double gnum = v[0], gden = v[1];
      if (gnum < 0.0 || gden == 0.0 || gnum > gden) {
        if (tid == 0) s_status = BCX_IT_FAIL_REWEIGHT;
      } else if (tid == 0) {
        s_alpha = 1.0 - gnum / gden;
        s_beta = sc * gnum / gden;
      }
    }
  }
  __syncthreads();
  double new_err = st->err, new_nw2 = 0.0, wf_new = 0.0;
  if (s_status == BCX_IT_OK) {
    // w <- alpha*w ; w[f] <- max(0, w[f] + beta)   giga.py:63-64 / frankwolfe.py:39-40
    const double alpha = s_alpha, beta = s_beta;
    const double wf_scaled = alpha * wf_old;
    wf_new = fmax(0.0, wf_scaled + beta);
    const double delta = wf_new - wf_scaled;
    double v[2] = {0.0, 0.0};
    for (int j = tid; j < d; j += blockDim.x) {
      const double x = alpha * a.xw[j] + delta * xf[j];   // = A w'
      a.tmp[j] = x;
      const double r = x - a.b[j];
      v[0] += r * r; v[1] += x * x;
    }
    block_allsum<2>(v, scratch);
    new_err = sqrt(v[0]); new_nw2 = v[1];
    if (checked && new_err > st->err) {                    // snnls.py:58
      if (tid == 0) s_status = BCX_IT_FAIL_MONOTONE;
    }
  }
  __syncthreads();
  const int status = s_status;
  if (status == BCX_IT_OK) {
    const double alpha = s_alpha;
    for (int s = tid; s < k; s += blockDim.x)
      if (s != slot) a.act_w[s] = alpha * a.act_w[s];
    for (int j = tid; j < d; j += blockDim.x) a.xw[j] = a.tmp[j];
    const int dst = slot >= 0 ? slot : k;
    if (slot < 0)
      for (int j = tid; j < d; j += blockDim.x) a.act_rows[(size_t)dst * d + j] = xf[j];
    if (tid == 0) {
      a.act_w[dst] = wf_new;
      if (slot < 0) { a.act_idx[dst] = f; a.act_norm[dst] = nf; st->k = k + 1; }
      st->err = new_err;
      const double nwn = sqrt(new_nw2);
      st->nw = nwn == 0.0 ? 1.0 : nwn;
      st->since_refresh += 1;
      if (checked) st->retried = 0;                        // snnls.py:62
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int64_t it = st->it;
    a.tr_sel[it] = f; a.tr_err[it] = st->err; a.tr_status[it] = status;
    st->it = it + 1;
    st->exact_mode = 0;
    if (status != BCX_IT_OK) {                             // snnls.py:63-72
      if (st->retried) { st->limit = 1; st->active = 0; st->halt = HALT_LIMIT; }
      else st->retried = 1;
    }
  }
  __syncthreads();
  if (!st->active) return;
  prepare_next(a, scratch);
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
  if (s->cfg.alg == BCX_ALG_GIGA)
    hipLaunchKernelGGL((apply_kernel<BCX_ALG_GIGA>), dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, a);
  else
    hipLaunchKernelGGL((apply_kernel<BCX_ALG_FW>), dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, a);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

// error() outside a build: refresh xw from the slots and recompute err (snnls.py:28-29)
__global__ __launch_bounds__(BCX_APPLY_THREADS) void error_refresh_kernel(ApplyArgs a) {
  __shared__ double scratch[BCX_SCRATCH];
  refresh_state(a, scratch, a.st->k > 0);
}

int bcx_launch_error_refresh(bcx_solver* s) {
  ApplyArgs a;
  fill_apply_args(s, a, nullptr);
  hipLaunchKernelGGL(error_refresh_kernel, dim3(1), dim3(BCX_APPLY_THREADS), 0, s->stream, a);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}
