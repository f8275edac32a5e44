// proj.hip -- device-native log-likelihood projection:
//   vecs[n][s] = loglik(z_n, theta_s) - mean_s' loglik(z_n, theta_s')        projector.py:19-21
// for the three example likelihoods of the reference
//   logistic regression   examples/common/model_lr.py:25-32
//   Poisson (softplus)    examples/common/model_poiss.py:25-38
//   Gaussian linear regr. examples/common/model_linreg.py:4-10
// as one fused kernel: Z . Theta^T on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, one wave per
// 16-row tile, all S columns in 16-column tiles) with the likelihood as the epilogue, and three
// consumers that never need the N x S matrix twice:
//   WRITE   : store the uncentred values + per-row sums (a second elementwise pass subtracts the mean)
//   COLSUM  : only the column sums  sum_n vecs[n][s]        (SparseVI gradient, sparsevi.py:70-74)
//   SELECT  : per row  corr_n = vecs[n].resid / ||vecs[n]|| / S  and its arg-max   (sparsevi.py:44-55)
// Arithmetic is fp64 throughout (the selection compares correlations to ~1e-7).
#include <algorithm>
#include <string>
#include "bcx_internal.h"
#include "dev_util.h"

enum { FAM_LOGISTIC = 0, FAM_POISSON = 1, FAM_LINREG = 2 };
enum { PMODE_WRITE = 0, PMODE_COLSUM = 1, PMODE_SELECT = 2 };

typedef double pv4d __attribute__((ext_vector_type(4)));

struct ProjArgs {
  const double* Z;      // N x ldz: features in columns [0, D), response (if any) in column ycol
  const double* theta;  // S x ldt
  int64_t N;
  int64_t ldz;
  int ldt, D, S, ycol;
  double param;         // linreg: sigma^2
  double* out;          // WRITE: N x ldo
  int64_t ldo;
  double* rowsum;       // WRITE: N
  double* colpart;      // COLSUM: gridDim.x x S partial column sums
  const double* resid;  // SELECT: S
  double resid_sum;     // SELECT: sum_s resid[s]
  double* best_val;     // SELECT: gridDim.x
  int64_t* best_idx;    // SELECT: gridDim.x
};

template <int FAM> __device__ __forceinline__ double loglik(double m, double y, double param, double c0) {
  if (FAM == FAM_LOGISTIC) {
    const double t = -m;                                   // model_lr.py:28
    return t < 100.0 ? -log1p(exp(t)) : -t;                // model_lr.py:29-31
  } else if (FAM == FAM_POISSON) {
    double s = m;                                          // model_poiss.py:25-30
    if (s > -100.0) s = log(fmax(s, 0.0) + log1p(exp(-fabs(s))));
    return y * s - c0 - exp(s);                            // model_poiss.py:38  (c0 = gammaln(y+1))
  } else {
    return c0 - (y * y - 2.0 * m * y + m * m) / (2.0 * param);   // model_linreg.py:10 (c0 = -0.5 log(2 pi sigsq))
  }
}

// sum over the 16 lanes of a DPP row (lanes that share l >> 4)
__device__ __forceinline__ double row16_sum(double v) {
  v += bcx_dpp_f64<0xB1>(v);
  v += bcx_dpp_f64<0x4E>(v);
  v += bcx_dpp_f64<0x141>(v);
  v += bcx_dpp_f64<0x140>(v);
  return v;
}

struct __attribute__((aligned(8))) pd2 { double x, y; };   // two consecutive k values, 8-byte aligned

// Register-blocked: a wave owns 32 rows x 64 columns of the product at a time (2 x 4 MFMA tiles).  The four
// k-slots of an MFMA step are fed from a permuted k order -- lane group lk supplies k = 8t + 2 lk (+1) to
// steps 2t (2t+1) -- so every lane fetches its A and B operands for two steps with ONE 16-byte load:
// 6 loads per 16 MFMAs instead of 2 loads per MFMA.
template <int FAM, int MODE>
__global__ __launch_bounds__(256) void proj_kernel(ProjArgs p) {
  extern __shared__ double lds[];            // COLSUM: 4 x S column accumulators
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int S = p.S, D = p.D;
  const int ngroup_c = (S + 63) / 64;
  double* colacc = lds + (size_t)wave * S;
  if (MODE == PMODE_COLSUM) {
    for (int c = lane; c < S; c += 64) colacc[c] = 0.0;
  }
  double bestv = -INFINITY;
  int64_t besti = 0x7fffffffffffffffLL;
  const double clin = (FAM == FAM_LINREG) ? -0.5 * log(2.0 * 3.14159265358979323846 * p.param) : 0.0;
  const int64_t nblk_r = (p.N + 31) / 32;
  for (int64_t br = (int64_t)blockIdx.x * 4 + wave; br < nblk_r; br += (int64_t)gridDim.x * 4) {
    const int64_t r0 = br * 32;
    const double* zrow[2];
    bool avalid[2];
#pragma unroll
    for (int tr = 0; tr < 2; ++tr) {
      const int64_t arow = r0 + 16 * tr + li;
      avalid[tr] = arow < p.N;
      zrow[tr] = p.Z + (avalid[tr] ? arow : 0) * p.ldz;
    }
    // responses / per-row constants of the 8 rows this lane's accumulator registers belong to
    double yv[2][4], c0[2][4];
#pragma unroll
    for (int tr = 0; tr < 2; ++tr)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = r0 + 16 * tr + lk + 4 * r;
        const double y = (p.ycol >= 0 && row < p.N) ? p.Z[row * p.ldz + p.ycol] : 0.0;
        yv[tr][r] = y;
        c0[tr][r] = (FAM == FAM_POISSON) ? lgamma(y + 1.0) : clin;
      }
    double rs[2][4] = {}, rq[2][4] = {}, rd[2][4] = {};
    for (int cg = 0; cg < ngroup_c; ++cg) {
      const double* trow[4];
      bool bvalid[4];
#pragma unroll
      for (int tc = 0; tc < 4; ++tc) {
        const int bcol = cg * 64 + 16 * tc + li;
        bvalid[tc] = bcol < S;
        trow[tc] = p.theta + (size_t)(bvalid[tc] ? bcol : 0) * p.ldt;
      }
      pv4d acc[2][4];
#pragma unroll
      for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) acc[tr][tc] = (pv4d){0.0, 0.0, 0.0, 0.0};
      for (int k0 = 0; k0 < D; k0 += 8) {
        const int k = k0 + 2 * lk;
        pd2 av[2], bv[4];
        if (k + 1 < D) {
#pragma unroll
          for (int tr = 0; tr < 2; ++tr) av[tr] = *(const pd2*)(zrow[tr] + k);
#pragma unroll
          for (int tc = 0; tc < 4; ++tc) bv[tc] = *(const pd2*)(trow[tc] + k);
        } else {
#pragma unroll
          for (int tr = 0; tr < 2; ++tr) { av[tr].x = k < D ? zrow[tr][k] : 0.0; av[tr].y = 0.0; }
#pragma unroll
          for (int tc = 0; tc < 4; ++tc) { bv[tc].x = k < D ? trow[tc][k] : 0.0; bv[tc].y = 0.0; }
        }
#pragma unroll
        for (int tr = 0; tr < 2; ++tr) if (!avalid[tr]) { av[tr].x = 0.0; av[tr].y = 0.0; }
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) if (!bvalid[tc]) { bv[tc].x = 0.0; bv[tc].y = 0.0; }
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
          for (int tc = 0; tc < 4; ++tc) {
            acc[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tr].x, bv[tc].x, acc[tr][tc], 0, 0, 0);
            acc[tr][tc] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[tr].y, bv[tc].y, acc[tr][tc], 0, 0, 0);
          }
      }
      // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
      for (int tc = 0; tc < 4; ++tc) {
        const int col = cg * 64 + 16 * tc + li;
        const bool cvalid = col < S;
        const double rsd = (MODE == PMODE_SELECT && cvalid) ? p.resid[col] : 0.0;
        double csum = 0.0;
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int64_t row = r0 + 16 * tr + lk + 4 * r;
            const bool ok = cvalid && row < p.N;
            const double ll = ok ? loglik<FAM>(acc[tr][tc][r], yv[tr][r], p.param, c0[tr][r]) : 0.0;
            if (MODE == PMODE_WRITE) {
              if (ok) p.out[row * p.ldo + col] = ll;
              rs[tr][r] += ll;
            } else if (MODE == PMODE_COLSUM) {
              csum += ll;
            } else {
              rs[tr][r] += ll; rq[tr][r] += ll * ll; rd[tr][r] += ll * rsd;
            }
          }
        if (MODE == PMODE_COLSUM) {
          // add the 4 lane groups that hold the same column (xor 16, xor 32)
          csum += bcx_xor16_f64(csum);
          csum += bcx_xor32_f64(csum);
          if (lk == 0 && cvalid) colacc[col] += csum;
        }
      }
    }
    if (MODE == PMODE_WRITE) {
#pragma unroll
      for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double t = row16_sum(rs[tr][r]);
          const int64_t row = r0 + 16 * tr + lk + 4 * r;
          if (li == 0 && row < p.N) p.rowsum[row] = t;
        }
    }
    if (MODE == PMODE_SELECT) {
#pragma unroll
      for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double s1 = row16_sum(rs[tr][r]), s2 = row16_sum(rq[tr][r]), sd = row16_sum(rd[tr][r]);
          const int64_t row = r0 + 16 * tr + lk + 4 * r;
          const double mean = s1 / (double)S;
          const double dot = sd - mean * p.resid_sum;                  // (ll - mean) . resid
          const double nrm2 = s2 - (double)S * mean * mean;            // ||ll - mean||^2
          const double corr = dot / sqrt(nrm2) / (double)S;            // sparsevi.py:51
          if (row < p.N && (corr > bestv || (corr == bestv && row < besti))) { bestv = corr; besti = row; }
        }
    }
  }
  if (MODE == PMODE_COLSUM) {
    __syncthreads();
    double* outp = p.colpart + (size_t)blockIdx.x * S;
    for (int c = threadIdx.x; c < S; c += blockDim.x)
      outp[c] = ((lds[c] + lds[(size_t)S + c]) + lds[2 * (size_t)S + c]) + lds[3 * (size_t)S + c];
  }
  if (MODE == PMODE_SELECT) {
    // arg-max over the workgroup: (value desc, row asc)
    __shared__ double sv[4];
    __shared__ long long si[4];
    for (int off = 32; off >= 1; off >>= 1) {
      const double ov = __shfl_xor(bestv, off, BCX_WAVE);
      const long long oi = __shfl_xor((long long)besti, off, BCX_WAVE);
      if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
    }
    if (lane == 0) { sv[wave] = bestv; si[wave] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w)
        if (sv[w] > bestv || (sv[w] == bestv && si[w] < besti)) { bestv = sv[w]; besti = si[w]; }
      p.best_val[blockIdx.x] = bestv;
      p.best_idx[blockIdx.x] = besti;
    }
  }
}

// out[n][s] -= rowsum[n] / S    (lls -= lls.mean(axis=1)[:, None], projector.py:21)
__global__ __launch_bounds__(256) void center_kernel(double* out, int64_t ldo, const double* rowsum, int64_t N, int S) {
  const int64_t total = N * (int64_t)S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / S;
    const int s = (int)(i - n * S);
    out[n * ldo + s] -= rowsum[n] / (double)S;
  }
}

// colsum[s] = sum over the workgroup partials in a fixed order: one workgroup per 64 columns, four
// partial-index segments per column combined 0..3 (8 independent loads in flight per thread).
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const double* part, int nparts, int S, double* colsum) {
  __shared__ double seg[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
  double acc = 0.0;
  if (c < S) {
    int b = sg;
    for (; b + 28 < nparts; b += 32) {
      double v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = part[(size_t)(b + 4 * t) * S + c];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += v[t];
    }
    for (; b < nparts; b += 4) acc += part[(size_t)b * S + c];
  }
  seg[sg][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sg == 0 && c < S) colsum[c] = ((seg[0][threadIdx.x] + seg[1][threadIdx.x]) + seg[2][threadIdx.x]) + seg[3][threadIdx.x];
}
// the centring correction  sum_n (ll[n][s] - mean_n) = raw[s] - (1/S) sum_s' raw[s']
__global__ __launch_bounds__(256) void colsum_center_kernel(int S, double* colsum) {
  __shared__ double scratch[BCX_SCRATCH];
  double tot[1] = {0.0};
  for (int c = threadIdx.x; c < S; c += blockDim.x) tot[0] += colsum[c];
  block_allsum<1>(tot, scratch);
  const double corr = tot[0] / (double)S;
  for (int c = threadIdx.x; c < S; c += blockDim.x) colsum[c] -= corr;
}

__global__ __launch_bounds__(256) void select_final_kernel(const double* bv, const int64_t* bi, int nparts, double* out_val, int64_t* out_idx) {
  __shared__ double sv[256];
  __shared__ long long si[256];
  double v = -INFINITY; long long i = 0x7fffffffffffffffLL;
  for (int b = threadIdx.x; b < nparts; b += blockDim.x)
    if (bv[b] > v || (bv[b] == v && bi[b] < i)) { v = bv[b]; i = bi[b]; }
  sv[threadIdx.x] = v; si[threadIdx.x] = i;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 1; t < 256; ++t)
      if (sv[t] > v || (sv[t] == v && si[t] < i)) { v = sv[t]; i = si[t]; }
    *out_val = v; *out_idx = i;
  }
}

// ---- C ABI --------------------------------------------------------------------------------------
static thread_local std::string g_proj_err;
extern "C" const char* bcx_project_last_error(void) { return g_proj_err.c_str(); }

#define PROJ_HIP(call)                                                            \
  do {                                                                            \
    hipError_t _e = (call);                                                       \
    if (_e != hipSuccess) {                                                       \
      g_proj_err = std::string(#call) + ": " + hipGetErrorString(_e);             \
      return BCX_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

static int proj_grid(int64_t N) {
  int64_t tiles = (N + 31) / 32;
  int64_t wg = (tiles + 3) / 4;
  if (wg > 2048) wg = 2048;
  if (wg < 1) wg = 1;
  return (int)wg;
}

template <int MODE> static int launch_family(int family, dim3 grid, size_t shmem, hipStream_t st, const ProjArgs& p) {
  if (shmem > 48 * 1024) {
    PROJ_HIP(hipFuncSetAttribute((const void*)proj_kernel<FAM_LOGISTIC, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    PROJ_HIP(hipFuncSetAttribute((const void*)proj_kernel<FAM_POISSON, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    PROJ_HIP(hipFuncSetAttribute((const void*)proj_kernel<FAM_LINREG, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  }
  switch (family) {
    case FAM_LOGISTIC: hipLaunchKernelGGL((proj_kernel<FAM_LOGISTIC, MODE>), grid, dim3(256), shmem, st, p); break;
    case FAM_POISSON: hipLaunchKernelGGL((proj_kernel<FAM_POISSON, MODE>), grid, dim3(256), shmem, st, p); break;
    case FAM_LINREG: hipLaunchKernelGGL((proj_kernel<FAM_LINREG, MODE>), grid, dim3(256), shmem, st, p); break;
    default: g_proj_err = "unknown likelihood family"; return BCX_ERR_ARG;
  }
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}

static int fill(ProjArgs& p, int family, const void* Z, int64_t N, int64_t ldz, int D, int ycol, const void* theta,
                int S, int ldt, double param) {
  if (!Z || !theta || N < 0 || D < 1 || S < 1 || ldz < D || ldt < D || S > 4096) {
    g_proj_err = "bcx_project: bad arguments";
    return BCX_ERR_ARG;
  }
  if (family != FAM_LOGISTIC && (ycol < 0 || ycol >= ldz)) { g_proj_err = "bcx_project: response column required"; return BCX_ERR_ARG; }
  p.Z = (const double*)Z; p.theta = (const double*)theta; p.N = N; p.ldz = ldz; p.ldt = ldt; p.D = D; p.S = S;
  p.ycol = family == FAM_LOGISTIC ? -1 : ycol; p.param = param;
  p.out = nullptr; p.ldo = 0; p.rowsum = nullptr; p.colpart = nullptr; p.resid = nullptr; p.resid_sum = 0.0;
  p.best_val = nullptr; p.best_idx = nullptr;
  return BCX_OK;
}

// vecs (N x S, centred) into out_dev; rowsum_dev is N doubles of scratch.
extern "C" int bcx_project_write(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                 int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                 void* out_dev, int64_t ldo, void* rowsum_dev) {
  ProjArgs p;
  int rc = fill(p, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param);
  if (rc) return rc;
  if (!out_dev || !rowsum_dev || ldo < S) { g_proj_err = "bcx_project_write: bad output"; return BCX_ERR_ARG; }
  if (N == 0) return BCX_OK;
  p.out = (double*)out_dev; p.ldo = ldo; p.rowsum = (double*)rowsum_dev;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = launch_family<PMODE_WRITE>(family, dim3(proj_grid(N)), 0, st, p))) return rc;
  const int64_t total = N * (int64_t)S;
  const int g = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(center_kernel, dim3(g), dim3(256), 0, st, p.out, ldo, p.rowsum, N, S);
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}

// colsum_dev[s] = sum_n vecs[n][s] without materialising vecs.  work_dev: 2048 * S doubles.
extern "C" int bcx_project_colsum(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                  int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                  void* colsum_dev, void* work_dev) {
  ProjArgs p;
  int rc = fill(p, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param);
  if (rc) return rc;
  if (!colsum_dev || !work_dev) { g_proj_err = "bcx_project_colsum: bad output"; return BCX_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int grid = proj_grid(N);
  p.colpart = (double*)work_dev;
  if ((rc = launch_family<PMODE_COLSUM>(family, dim3(grid), 4 * (size_t)S * sizeof(double), st, p))) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((S + 63) / 64), dim3(256), 0, st, p.colpart, grid, S, (double*)colsum_dev);
  hipLaunchKernelGGL(colsum_center_kernel, dim3(1), dim3(256), 0, st, S, (double*)colsum_dev);
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}

// arg-max_n of vecs[n].resid / ||vecs[n]|| / S  (first maximum), result to result_dev = {double value, int64 row}.
// work_dev: 2048 doubles + 2048 int64.
extern "C" int bcx_project_select(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                  int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                  const void* resid_dev, double resid_sum, void* result_dev, void* work_dev) {
  ProjArgs p;
  int rc = fill(p, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param);
  if (rc) return rc;
  if (!resid_dev || !result_dev || !work_dev) { g_proj_err = "bcx_project_select: bad output"; return BCX_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int grid = proj_grid(N);
  p.resid = (const double*)resid_dev; p.resid_sum = resid_sum;
  p.best_val = (double*)work_dev; p.best_idx = (int64_t*)((double*)work_dev + 2048);
  if ((rc = launch_family<PMODE_SELECT>(family, dim3(grid), 0, st, p))) return rc;
  hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(256), 0, st, p.best_val, p.best_idx, grid, (double*)result_dev,
                     (int64_t*)((double*)result_dev + 1));
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}
