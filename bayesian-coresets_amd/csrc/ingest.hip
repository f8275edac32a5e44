// ingest.hip -- constructor work of the reference solvers, one fused pass per row:
//   Anorms = sqrt((A**2).sum(axis=0)); An = A / Anorms      giga.py:10-13, frankwolfe.py:10-13, orthopursuit.py:12-15
//   b = vecs.sum(axis=0)                                    hilbert.py:24
//   sum(Anorms)                                             frankwolfe.py:22,25
// HBM-bound: reads N*d source elements once, writes N*d stored elements once.
#include <hip/hip_fp16.h>
#include "bcx_internal.h"
#include "dev_util.h"

template <typename TD> __device__ __forceinline__ TD to_store(double v) { return (TD)v; }
// sum of one 16-byte piece, in the association the projection's former centring pass used ((x + y) per piece)
template <int EPS> __device__ __forceinline__ double piece_sum(const double* v) {
  if constexpr (EPS == 2) return v[0] + v[1];
  else return (v[0] + v[1]) + (v[2] + v[3]);
}
template <> __device__ __forceinline__ __half to_store<__half>(double v) { return __float2half_rn((float)v); }

// One workgroup per chunk of BCX_CHUNK_ROWS rows; one wave per row, lanes stride the columns
// (coalesced 512-byte segments of the fp64 source), up to 16 waves per workgroup.  Column sums are
// accumulated per wave in LDS (each lane owns its columns -> no conflicts, fixed order) and
// combined wave 0..nw-1; nw depends only on d, so the summation tree is reproducible.
template <typename TS, typename TD>
__global__ __launch_bounds__(1024) void ingest_kernel(const TS* src, int64_t ld_src, int64_t row_begin,
                                                      int64_t rows, int d, TD* __restrict__ An, int ld, int ld64,
                                                      double* A64, double* __restrict__ norms,
                                                      double* __restrict__ chunk_sums, DevState* st, int center, int acc_global) {
  extern __shared__ double lds[];  // nw * d column accumulators + nw norm accumulators
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int64_t chunk = blockIdx.x;
  // acc_global (rows too long for d doubles of LDS; one wave per workgroup): the chunk's row of chunk_sums IS the accumulator
  double* colacc = acc_global ? chunk_sums + (row_begin / BCX_CHUNK_ROWS + chunk) * (int64_t)(d + 1) : lds + (size_t)wave * d;
  for (int c = lane; c < d; c += 64) colacc[c] = 0.0;
  double normacc = 0.0;
  const int64_t r0 = chunk * BCX_CHUNK_ROWS;
  const int64_t r1 = (r0 + BCX_CHUNK_ROWS < rows) ? r0 + BCX_CHUNK_ROWS : rows;
  for (int64_t r = r0 + wave; r < r1; r += nw) {
    const TS* x = src + r * ld_src;
    // squares are summed in the order of the vectorised kernel below (a lane owns the 16-byte pieces
    // lane, lane + 64, ... of the row), so both kernels give the same norm bit for bit
    constexpr int EPS = 16 / (int)sizeof(TS);
    // center: subtract the row mean first (projector.py:21 folded into the constructor pass); the mean is summed in
    // the piece order of the vectorised kernel, so both kernels centre alike
    double mean = 0.0;
    if (center) {
      double acc = 0.0;
      for (int c0 = lane * EPS; c0 < d; c0 += 64 * EPS) {
        double pv[EPS];
#pragma unroll
        for (int e = 0; e < EPS; ++e) pv[e] = c0 + e < d ? (double)x[c0 + e] : 0.0;
        acc += piece_sum<EPS>(pv);
      }
      mean = wave_allsum(acc) / (double)d;
    }
    double ss = 0.0;
    for (int c0 = lane * EPS; c0 < d; c0 += 64 * EPS) {
#pragma unroll
      for (int e = 0; e < EPS; ++e) {
        if (c0 + e < d) { const double v = (double)x[c0 + e] - mean; ss = fma(v, v, ss); }   // (explicit: both kernels contract alike)
      }
    }
    ss = wave_allsum(ss);
    const double nrm = sqrt(ss);
    const int64_t lr = row_begin + r;
    if (lane == 0) {
      norms[lr] = nrm;
      if (!(nrm > 0.0)) {
        // remember the smallest offending local row (+1); 0 means "none"
        const int32_t want = (int32_t)(lr + 1);
        int32_t cur = atomicCAS(&st->zero_row, 0, want);
        while (cur != 0 && cur > want) {
          const int32_t prev = atomicCAS(&st->zero_row, cur, want);
          if (prev == cur) break;
          cur = prev;
        }
      }
    }
    normacc += nrm;
    TD* y = An + lr * (int64_t)ld;
    double* raw = A64 ? A64 + lr * (int64_t)ld64 : nullptr;
    const bool copy_raw = raw && ((const void*)raw != (const void*)x || center);
    for (int c = lane; c < d; c += 64) {
      double v = (double)x[c] - mean;  // second touch hits L1/L2
      colacc[c] += v;
      y[c] = to_store<TD>(v / nrm);
      if (copy_raw) raw[c] = v;
    }
  }
  __syncthreads();
  const int64_t gchunk = row_begin / BCX_CHUNK_ROWS + chunk;
  double* out = chunk_sums + gchunk * (int64_t)(d + 1);
  if (acc_global) {                       // (one wave: the sums are in place)
    if (threadIdx.x == 0) out[d] = normacc;
    return;
  }
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    double acc = lds[c];
    for (int w = 1; w < nw; ++w) acc += lds[(size_t)w * d + c];
    out[c] = acc;
  }
  double* nacc = lds + (size_t)nw * d;
  __syncthreads();
  if (lane == 0) nacc[wave] = normacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = nacc[0];
    for (int w = 1; w < nw; ++w) acc += nacc[w];
    out[d] = acc;
  }
}

// ---- vectorised form: 16-byte pieces, the row stays in registers between the norm and the store, column sums
// accumulate in registers (a lane always owns the same columns) and meet in LDS once per chunk.  Same row ->
// wave assignment and the same wave 0..nw-1 combination as the scalar kernel above, so the chunk sums (hence
// b) are bit-identical between the two.  Needs d % EPS == 0 and 16-byte aligned source rows.
template <typename TS> struct SrcVec;
template <> struct SrcVec<double> { typedef double2 V; static constexpr int EPS = 2; };
template <> struct SrcVec<float>  { typedef float4 V;  static constexpr int EPS = 4; };
__device__ __forceinline__ void unpack(const double2& v, double* o) { o[0] = v.x; o[1] = v.y; }
__device__ __forceinline__ void unpack(const float4& v, double* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
template <typename TD, int EPS> __device__ __forceinline__ void store_piece(TD* dst, const double* v) {
  if constexpr (sizeof(TD) == 8) {
#pragma unroll
    for (int e = 0; e < EPS; e += 2) *(double2*)(dst + e) = make_double2(v[e], v[e + 1]);
  } else if constexpr (sizeof(TD) == 4) {
    if constexpr (EPS == 2) *(float2*)dst = make_float2((float)v[0], (float)v[1]);
    else *(float4*)dst = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  } else {
#pragma unroll
    for (int e = 0; e < EPS; e += 2)
      *(__half2*)(dst + e) = __halves2half2(__float2half_rn((float)v[e]), __float2half_rn((float)v[e + 1]));
  }
}

// Workgroup size bound by the row length: a lane holds CHI * EPS values of the row AND as many column-sum accumulators,
// both fp64.  Up to 16 values per lane (d <= 1024) that fits the 128 registers of a 1024-thread workgroup; with 32
// (d <= 2048) or 64 (d <= 4096) it does not -- the kernel spilled up to 644 registers there -- so those instantiations are
// built for 512 / 256 threads (the launcher's wave count for such rows is bounded by the LDS accumulators anyway).
#define INGEST_VEC_THREADS(CHI, EPS) ((CHI) * (EPS) >= 64 ? 256 : (CHI) * (EPS) >= 32 ? 512 : 1024)
template <typename TS, typename TD, int CHI>
__global__ __launch_bounds__(INGEST_VEC_THREADS(CHI, SrcVec<TS>::EPS)) void ingest_vec_kernel(const TS* src, int64_t ld_src, int64_t row_begin,
                                                          int64_t rows, int d, TD* __restrict__ An, int ld, int ld64,
                                                          double* A64, double* __restrict__ norms,
                                                          double* __restrict__ chunk_sums, DevState* st, int center, int /*acc_global: scalar kernel only*/) {
  typedef typename SrcVec<TS>::V V;
  constexpr int EPS = SrcVec<TS>::EPS;
  extern __shared__ double lds[];  // nw * d column sums + nw norm sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int npieces = d / EPS;
  double ca[CHI][EPS];
#pragma unroll
  for (int t = 0; t < CHI; ++t)
#pragma unroll
    for (int e = 0; e < EPS; ++e) ca[t][e] = 0.0;
  double normacc = 0.0;
  const int64_t chunk = blockIdx.x;
  const int64_t r0 = chunk * BCX_CHUNK_ROWS;
  const int64_t r1 = (r0 + BCX_CHUNK_ROWS < rows) ? r0 + BCX_CHUNK_ROWS : rows;
  for (int64_t r = r0 + wave; r < r1; r += nw) {
    const V* x = (const V*)(src + r * ld_src);
    V xv[CHI];
#pragma unroll
    for (int t = 0; t < CHI; ++t) {
      const int pc = lane + 64 * t;
      xv[t] = x[pc < npieces ? pc : 0];
    }
    double v[CHI][EPS];
#pragma unroll
    for (int t = 0; t < CHI; ++t) unpack(xv[t], v[t]);
    if (center) {
      // row mean subtracted here (projector.py:21): the row is in registers anyway, one more wave sum -- the separate
      // centring pass over the N x S matrix (read + write) is gone.  Same lane -> column ownership and association as
      // that pass had, so the centred values are the same bits.
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < CHI; ++t)
        if (lane + 64 * t < npieces) acc += piece_sum<EPS>(v[t]);
      const double mean = wave_allsum(acc) / (double)d;
#pragma unroll
      for (int t = 0; t < CHI; ++t)
#pragma unroll
        for (int e = 0; e < EPS; ++e) v[t][e] -= mean;
    }
    double ss = 0.0;
#pragma unroll
    for (int t = 0; t < CHI; ++t) {
      if (lane + 64 * t < npieces) {
#pragma unroll
        for (int e = 0; e < EPS; ++e) ss = fma(v[t][e], v[t][e], ss);
      }
    }
    ss = wave_allsum(ss);
    const double nrm = sqrt(ss);
    const int64_t lr = row_begin + r;
    if (lane == 0) {
      norms[lr] = nrm;
      if (!(nrm > 0.0)) {
        const int32_t want = (int32_t)(lr + 1);
        int32_t cur = atomicCAS(&st->zero_row, 0, want);
        while (cur != 0 && cur > want) {
          const int32_t prev = atomicCAS(&st->zero_row, cur, want);
          if (prev == cur) break;
          cur = prev;
        }
      }
    }
    normacc += nrm;
    TD* y = An + lr * (int64_t)ld;
    double* raw = A64 ? A64 + lr * (int64_t)ld64 : nullptr;
    const bool copy_raw = raw && ((const void*)raw != (const void*)x || center);
#pragma unroll
    for (int t = 0; t < CHI; ++t) {
      const int pc = lane + 64 * t;
      if (pc < npieces) {
        double q[EPS];
#pragma unroll
        for (int e = 0; e < EPS; ++e) { ca[t][e] += v[t][e]; q[e] = v[t][e] / nrm; }
        store_piece<TD, EPS>(y + pc * EPS, q);
        if (copy_raw) store_piece<double, EPS>(raw + pc * EPS, v[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < CHI; ++t) {
    const int pc = lane + 64 * t;
    if (pc < npieces) {
#pragma unroll
      for (int e = 0; e < EPS; ++e) lds[(size_t)wave * d + pc * EPS + e] = ca[t][e];
    }
  }
  __syncthreads();
  const int64_t gchunk = row_begin / BCX_CHUNK_ROWS + chunk;
  double* out = chunk_sums + gchunk * (int64_t)(d + 1);
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    double acc = lds[c];
    for (int w = 1; w < nw; ++w) acc += lds[(size_t)w * d + c];
    out[c] = acc;
  }
  double* nacc = lds + (size_t)nw * d;
  __syncthreads();
  if (lane == 0) nacc[wave] = normacc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double acc = nacc[0];
    for (int w = 1; w < nw; ++w) acc += nacc[w];
    out[d] = acc;
  }
}

int bcx_launch_ingest(bcx_solver* s, const void* src, int src_dtype, int64_t ld_src, int64_t row_begin, int64_t rows,
                      int center) {
  const int d = s->cfg.d;
  const int64_t nblk = (rows + BCX_CHUNK_ROWS - 1) / BCX_CHUNK_ROWS;
  int nw = (int)((144 * 1024) / ((size_t)d * 8));   // column accumulators must fit the 160 KiB LDS
  if (nw > 16) nw = 16;
  // long rows: the vector kernel's register budget (INGEST_VEC_THREADS).  The rule depends on d alone, and the scalar
  // kernel follows it too, so both forms add the chunk sums in the same order (bit-identical b) for every row length.
  if (d > 2048 && nw > 4) nw = 4;
  else if (d > 1024 && nw > 8) nw = 8;
  // rows of more than 18432 values: not even one wave's d column accumulators fit the LDS -- one wave per chunk adds into
  // the chunk's row of chunk_sums directly (its lanes own their columns: no conflicts, the same fixed order)
  const int acc_global = nw < 1 ? 1 : 0;
  if (nw < 1) nw = 1;
  const size_t shmem = acc_global ? 64 : ((size_t)nw * d + nw) * sizeof(double);
  dim3 grid((unsigned)nblk), block(64 * nw);
  const int esz = src_dtype == BCX_F64 ? 8 : 4, eps = 16 / esz;
  const bool vec_ok = d % eps == 0 && ld_src % eps == 0 && ((uintptr_t)src % 16 == 0) && d / eps <= 64 * 16 &&
                      !bcx_dev_env("BCX_INGEST_SCALAR");
  int chi = 1;
  while (vec_ok && chi * 64 < d / eps) chi <<= 1;
#define LAUNCH_K(KFN)                                                                                         \
  do {                                                                                                        \
    auto kfn = KFN;                                                                                           \
    if (shmem > 48 * 1024)                                                                                    \
      BCX_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));  \
    hipLaunchKernelGGL(kfn, grid, block, shmem, s->stream, (const TS_*)src, ld_src, row_begin, rows, d,       \
                       (TD_*)s->An, s->ld, s->ld64, s->A64, s->norms, s->chunk_sums, s->st, center, acc_global); \
  } while (0)
#define LAUNCH(TS, TD)                                                                                        \
  do {                                                                                                        \
    typedef TS TS_; typedef TD TD_;                                                                           \
    if (!vec_ok) LAUNCH_K((ingest_kernel<TS, TD>));                                                           \
    else if (chi == 1) LAUNCH_K((ingest_vec_kernel<TS, TD, 1>));                                              \
    else if (chi == 2) LAUNCH_K((ingest_vec_kernel<TS, TD, 2>));                                              \
    else if (chi == 4) LAUNCH_K((ingest_vec_kernel<TS, TD, 4>));                                              \
    else if (chi == 8) LAUNCH_K((ingest_vec_kernel<TS, TD, 8>));                                              \
    else LAUNCH_K((ingest_vec_kernel<TS, TD, 16>));                                                           \
  } while (0)
  const int sd = s->cfg.store_dtype;
  if (src_dtype == BCX_F64) {
    if (sd == BCX_F32) LAUNCH(double, float); else if (sd == BCX_F16) LAUNCH(double, __half); else LAUNCH(double, double);
  } else {
    if (sd == BCX_F32) LAUNCH(float, float); else if (sd == BCX_F16) LAUNCH(float, __half); else LAUNCH(float, double);
  }
#undef LAUNCH
#undef LAUNCH_K
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}

// Finish construction: b and sum(Anorms) from the chunk sums in global chunk order
// (or b from the caller), ||b||, bn = b/||b|| (giga.py:15-18), and the zero state
// xw = 0, err = ||b|| (snnls.py:15,29).
// Column sums of the chunk sums, level 1: workgroup g adds chunks [g*FIN_GROUP, (g+1)*FIN_GROUP) in order.
// The grouping depends only on the GLOBAL chunk index, so b stays bit-identical for any shard count.
#define FIN_GROUP 128
__global__ __launch_bounds__(256) void finalize_partial_kernel(int d, const double* __restrict__ sums, int64_t n_sums,
                                                               double* __restrict__ part) {
  const int64_t j0 = (int64_t)blockIdx.x * FIN_GROUP;
  const int64_t j1 = j0 + FIN_GROUP < n_sums ? j0 + FIN_GROUP : n_sums;
  for (int c = threadIdx.x; c <= d; c += blockDim.x) {
    double acc = 0.0;
    int64_t j = j0;
    for (; j + 8 <= j1; j += 8) {
      double m[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) m[t] = sums[(j + t) * (int64_t)(d + 1) + c];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += m[t];
    }
    for (; j < j1; ++j) acc += sums[j * (int64_t)(d + 1) + c];
    part[(int64_t)blockIdx.x * (d + 1) + c] = acc;
  }
}

__global__ __launch_bounds__(256) void finalize_kernel(int d, int have_b, const double* __restrict__ sums,
                                                       int64_t n_sums, double* __restrict__ b, double* __restrict__ bn,
                                                       double* __restrict__ xw, DevState* st) {
  __shared__ double scratch[BCX_SCRATCH];
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    if (!have_b) {
      double acc = 0.0;
      for (int64_t j = 0; j < n_sums; ++j) acc += sums[j * (int64_t)(d + 1) + c];
      b[c] = acc;
    }
    xw[c] = 0.0;
  }
  double sig = 0.0;
  if (threadIdx.x == 0)
    for (int64_t j = 0; j < n_sums; ++j) sig += sums[j * (int64_t)(d + 1) + d];
  __syncthreads();
  double v[1] = {0.0};
  for (int c = threadIdx.x; c < d; c += blockDim.x) v[0] += b[c] * b[c];
  block_allsum<1>(v, scratch);
  const double bnorm = sqrt(v[0]);
  for (int c = threadIdx.x; c < d; c += blockDim.x) bn[c] = b[c] / bnorm;
  if (threadIdx.x == 0) {
    st->bnorm = bnorm;
    st->sigma = sig;
    st->err = bnorm;
    st->nw = 1.0;
    st->k = 0;
    st->limit = 0;
    st->retried = 0;
    st->active = 0;
    st->halt = HALT_NONE;
    st->since_refresh = 0;
    st->qscale = 1.0;
    st->np = 0;
    st->hvalid = 1;
    st->omp_ill = 0;
    st->omp_mode = OMP_IDLE;
  }
}

int bcx_launch_finalize(bcx_solver* s, int have_b, const double* gathered, int64_t n_gathered) {
  const double* sums = gathered ? gathered : s->chunk_sums;
  int64_t n = gathered ? n_gathered : s->n_chunks;
  const int d = s->cfg.d;
  if (n > 2 * FIN_GROUP) {
    // two levels: groups of FIN_GROUP chunks in parallel, then the (few) group sums in order
    const int64_t ng = (n + FIN_GROUP - 1) / FIN_GROUP;
    if (s->fin_cap < ng) {
      if (s->fin_part) (void)hipFree(s->fin_part);
      s->fin_part = nullptr;
      BCX_HIP(hipMalloc((void**)&s->fin_part, (size_t)ng * (d + 1) * sizeof(double)));
      s->fin_cap = ng;
    }
    hipLaunchKernelGGL(finalize_partial_kernel, dim3((unsigned)ng), dim3(256), 0, s->stream, d, sums, n, s->fin_part);
    sums = s->fin_part;
    n = ng;
  }
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, s->stream, d, have_b, sums, n, s->b, s->bn, s->xw, s->st);
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}
