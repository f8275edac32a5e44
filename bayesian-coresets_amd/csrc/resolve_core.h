// resolve_core.h -- scan partials -> candidate rows -> exact fp64 re-score -> winner: the first phase of every tail of a
// greedy iteration (resolve_kernel / tail_kernel / tail_exchange_kernel in resolve.hip, the fused OMP step in omp_lh.hip).
#pragma once
#include <hip/hip_fp16.h>
#include "bcx_internal.h"
#include "dev_util.h"

struct ResolveArgs {
  PartialView pv;
  int n_partials;
  DevState* st;
  const void* An;
  int store_f64;
  int store_f16;
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
  int need_score; // 0: a single candidate wins unscored (single shard, GIGA / FW: nobody reads its exact score)
  double* rec;    // out: d + 4 doubles (may be null inside tail_kernel)
};

// result of resolve, in LDS
struct Winner {
  double score, norm, flags;
  int64_t gidx;
  int lrow;
};

#define PP_MAX 8   // partials per thread (n_partials <= 2048, 256 threads)

static __device__ __forceinline__ double giga_score64(double s0, double s1) {
  const bool ok = (s1 > -1.0 + 1e-14) && (1.0 - s1 * s1 > 0.0);   // giga.py:33
  const double den = ok ? sqrt(1.0 - s1 * s1) : INFINITY;          // giga.py:35-36
  return s0 / den;                                                 // giga.py:38
}

static __device__ __forceinline__ double raw_elem(const ResolveArgs& a, int64_t i, int j, double nrm) {
  if (a.A64) return a.A64[i * (int64_t)a.ld64 + j];
  if (a.store_f64) return ((const double*)a.An)[i * (int64_t)a.ld + j] * nrm;
  if (a.store_f16) return (double)__half2float(((const __half*)a.An)[i * (int64_t)a.ld + j]) * nrm;
  return (double)((const float*)a.An)[i * (int64_t)a.ld + j] * nrm;
}

// Partials -> candidates -> exact scores -> winner (LDS `win`); raw winner row to xf_lds (optional)
// and the record to a.rec (optional).  All threads of a workgroup of >= 256 threads; workgroups of a grid that all call
// it (the fused OMP step, omp_lh.hip) compute the same winner -- there is no state it changes besides the diagnostics.
// PP = partials per thread: the first 2048 / PP threads of the workgroup hold them (PP = 8: 256 threads).
template <int PP = PP_MAX>
static __device__ __forceinline__ void resolve_core(const ResolveArgs& a, Winner* win, double* xf_lds, double* scratch) {
  __shared__ int cand[BCX_MAX_CAND];        // local row
  __shared__ double cscore[BCX_MAX_CAND];
  __shared__ int ncand, overflow, minrow;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int np = a.n_partials;
  const bool exact = a.exact || a.st->exact_mode;
  // one round trip: this thread's partial bounds
  double u1[PP], u2[PP], u3[PP], lo[PP];
  int r1[PP], r2[PP];               // the rows behind U1 / U2 come along: no second round trip
#pragma unroll
  for (int t = 0; t < PP; ++t) {
    const int p = tid + t * (BCX_MAX_PARTIALS / PP);
    const bool ok = p < np && tid < BCX_MAX_PARTIALS / PP;   // (wider workgroups: the first 2048 / PP threads hold the partials)
    u1[t] = ok ? a.pv.U1[p] : -INFINITY;
    u2[t] = ok ? a.pv.U2[p] : -INFINITY;
    u3[t] = ok ? a.pv.U3[p] : -INFINITY;
    lo[t] = ok ? a.pv.L[p] : -INFINITY;
    r1[t] = ok ? a.pv.i1[p] : 0;
    r2[t] = ok ? a.pv.i2[p] : 0;
  }
  if (tid == 0) { ncand = 0; overflow = 0; minrow = 0x7fffffff; }
  double lmax = -INFINITY;
#pragma unroll
  for (int t = 0; t < PP; ++t) lmax = fmax(lmax, exact ? u1[t] : lo[t]);
  const double Lstar = block_allmax(lmax, scratch);   // (exact: the maximum score itself)
  BCX_STAMP(a.st, 1);
#pragma unroll
  for (int t = 0; t < PP; ++t) {
    if (u1[t] > -INFINITY && u1[t] >= Lstar) {
      const int slot = atomicAdd(&ncand, 1);
      if (slot < BCX_MAX_CAND) cand[slot] = r1[t];
    }
    if (!exact) {
      if (u2[t] > -INFINITY && u2[t] >= Lstar) {
        const int slot = atomicAdd(&ncand, 1);
        if (slot < BCX_MAX_CAND) cand[slot] = r2[t];
      }
      if (u3[t] > -INFINITY && u3[t] >= Lstar) overflow = 1;
    }
  }
  __syncthreads();
  int nc = ncand;
  const bool storm = nc > BCX_MAX_CAND;
  if (storm && !exact) overflow = 1;
  if (storm && exact) {
    // exact scores tie across more than 64 workgroups: the lowest row among the maxima wins
#pragma unroll
    for (int t = 0; t < PP; ++t) {
      if (u1[t] > -INFINITY && u1[t] == Lstar) atomicMin(&minrow, r1[t]);
    }
    __syncthreads();
    if (tid == 0) cand[0] = minrow;
    nc = 1;
    __syncthreads();
  }   // (a storm without exact scores overflows: the candidate list is not read)
  if (overflow || nc == 0) {
    if (tid == 0) {
      win->score = -INFINITY; win->gidx = -1; win->norm = 0.0; win->lrow = -1;
      win->flags = overflow ? BCX_REC_OVERFLOW : 0.0;
      if (a.rec) { a.rec[0] = -INFINITY; a.rec[1] = -1.0; a.rec[2] = 0.0; a.rec[3] = win->flags; }
    }
    __syncthreads();
    return;
  }
  if (tid == 0 && blockIdx.x == 0) {   // diagnostics: fire-and-forget atomics (a read-modify-write would stall wave 0 for a memory round trip)
    atomicAdd((unsigned long long*)&a.st->n_cand, (unsigned long long)nc);
    atomicAdd((unsigned long long*)&a.st->n_resolved, 1ull);
  }
  BCX_STAMP(a.st, 2);
  // exact fp64 score of every candidate, one wave per candidate; norm, row and query loads are independent
  const double* q0 = a.q64;
  const double* q1 = a.q64 + a.ld64;
  const bool dual = a.alg == BCX_ALG_GIGA;
  // No re-score when (a) the scan was exact: every candidate's score IS Lstar, the maximum of the scan's own fp64
  // scores -- comparing shards by that value (and rows by index) is what makes the pick independent of the shard
  // count; a re-score in this kernel's summation order could split rows the scan tied and name a different winner on
  // one shard than on four (found by the 4-rank exact-fallback test on rows duplicated up to scaling) -- or (b) a
  // single candidate on a single shard, whose score nobody reads (the usual case: 1.00-1.01 candidates per iteration).
  const bool unscored = exact || (!a.need_score && nc == 1);
  if (unscored && tid < nc) cscore[tid] = exact ? Lstar : 0.0;
  // A single candidate that has to be scored (the OMP step compares its score with the negative direction, a row shard
  // sends it to its peers): the wave that scores it leaves the raw row in LDS / in the record as it reads it, and the
  // winner's row is not fetched a second time -- one dependent round trip less (the usual case: 1.00-1.01 candidates).
  const bool stage1 = !unscored && nc == 1;
  __shared__ double s_nrm;
  for (int c = wave; c < nc && !unscored; c += nwaves) {
    const int64_t i = cand[c];
    const double nrm = a.norms[i];
    double s0 = 0.0, s1 = 0.0;
    // a lane's elements j = lane, lane + 64, ... are summed in that order (what makes rows that tie mathematically tie bit
    // for bit, for any number of candidates); the loads of eight of them are in flight together
    for (int j0 = 0; j0 < a.d; j0 += 512) {
      double rv[8], qa[8], qb[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int j = j0 + 64 * t + lane;
        const bool ok = j < a.d;
        // An = A / Anorms element by element (giga.py:13): exactly +-1 for d = 1, so mathematically tied
        // rows tie bit-for-bit as they do in the reference; stored rows are already normalised
        rv[t] = ok ? raw_elem(a, i, j, 1.0) : 0.0;
        qa[t] = ok ? q0[j] : 0.0;
        qb[t] = (ok && dual) ? q1[j] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int j = j0 + 64 * t + lane;
        if (j < a.d) {
          double v = rv[t];
          if (stage1) {
            const double raw = a.A64 ? v : v * nrm;
            if (xf_lds) xf_lds[j] = raw;
            if (a.rec) a.rec[BCX_REC_HDR + j] = raw;
          }
          if (a.A64) v /= nrm;
          s0 += v * qa[t];
          if (dual) s1 += v * qb[t];
        }
      }
    }
    s0 = wave_allsum(s0);
    if (dual) s1 = wave_allsum(s1);
    if (lane == 0) { cscore[c] = dual ? giga_score64(s0, s1) : s0; if (stage1) s_nrm = nrm; }
  }
  __syncthreads();
  BCX_STAMP(a.st, 3);
  if (tid == 0) {
    int best = 0;
    for (int c = 1; c < nc; ++c)
      if (cscore[c] > cscore[best] || (cscore[c] == cscore[best] && cand[c] < cand[best])) best = c;
    if (cscore[best] != cscore[best])   // NaN never wins '>' : prefer any finite candidate
      for (int c = 0; c < nc; ++c) if (cscore[c] == cscore[c]) { best = c; break; }
    win->lrow = cand[best];
    win->score = cscore[best];
    win->gidx = a.row_offset + cand[best];
    win->flags = BCX_REC_VALID;
  }
  __syncthreads();
  const int64_t wrow = win->lrow;
  const double nrm = stage1 ? s_nrm : a.norms[wrow];
  if (tid == 0) {
    win->norm = nrm;
    if (a.rec) { a.rec[0] = win->score; a.rec[1] = (double)win->gidx; a.rec[2] = nrm; a.rec[3] = BCX_REC_VALID; }
  }
  if (!stage1) {
    for (int j = tid; j < a.d; j += blockDim.x) {
      const double raw = raw_elem(a, wrow, j, nrm);
      if (xf_lds) xf_lds[j] = raw;
      if (a.rec) a.rec[BCX_REC_HDR + j] = raw;
    }
  }
  __syncthreads();
  BCX_STAMP(a.st, 4);
}


void bcx_fill_resolve_args(bcx_solver* s, ResolveArgs& a, double* send_dev, int exact);   // resolve.hip (host)
