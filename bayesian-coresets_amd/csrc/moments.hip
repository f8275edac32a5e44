// moments.hip -- closed-form column sums of the Gaussian linear-regression projection.
//
// SparseVI needs  colsum_s = sum_n vecs[n][s]  of a FRESH projection of the whole data set at every ADAM step
// (sparsevi.py:23-42, 69-76: 1 + opt_itrs projections per greedy step).  For the linear-regression likelihood
// (examples/common/model_linreg.py:4-10)
//     ll[n][s] = c - (y_n - x_n.theta_s)^2 / (2 sigsq),     vecs = ll - rowmean(ll)             (projector.py:19-21)
// the sum over n is a quadratic form in the one-time second moments of the data  M = Z^T Z,  Z = [X, y]  ((D+1) x (D+1)):
//     sum_n (y_n - x_n.theta)^2 = yy - 2 theta^T g + theta^T G theta =: Q(theta),     G = X^T X, g = X^T y
//     colsum_s = -(Q(theta_s) - mean_t Q(theta_t)) / (2 sigsq)
// Evaluated on the differences delta_s = theta_s - thetabar so that the common part cancels exactly:
//     Q(thetabar + delta) = Q(thetabar) + 2 delta^T (G thetabar - g) + delta^T G delta
//     T_s = 2 delta_s^T v + delta_s^T G delta_s,   v = G thetabar - g,       colsum_s = -(T_s - mean_t T_t) / (2 sigsq)
// That is O(S D^2) per call instead of the 2 N D S flops of a projection (configs[4]: 2.3e7 against 7.7e11).
//
//   moments_kernel         M partials: fp64 MFMA (v_mfma_f64_16x16x4_f64), one workgroup per (64 x 64 block pair of the
//                          upper triangle, slice of rows); operands staged through LDS, next chunk prefetched in registers
//   moments_reduce_kernel  partials summed in slice order (fixed association: reproducible), both triangles written
//   moments_mean_kernel    thetabar
//   moments_quad_kernel    T_s: Delta G on the fp64 matrix cores (a workgroup per 16 x 16 tile, its waves sharing the inner index),
//                          workgroup 0 sums the tile partials as they appear, centres and scales
#include <algorithm>
#include <string>
#include "bcx_internal.h"
#include "dev_util.h"

#include "moments_quad.h"

#define MOM_BLK 64          // columns per block
#define MOM_ROWS 32         // rows staged per chunk (4 waves x 2 k-steps x 4 rows)
#define MOM_LDS_LD 80       // doubles per staged row: rows 0/1 and 2/3 of a k-step land on disjoint bank halves
#define MOM_MAX_COLS 1024   // (D + 1) <= 1024: 16 blocks, 136 block pairs

// part[(slice * npairs + pair) * 4096 + i * 64 + j]
__global__ __launch_bounds__(256, 2) void moments_kernel(const double* __restrict__ Z, int64_t N, int64_t ldz, int C,
                                                         int nblk, int nslices, int64_t rows_per_slice,
                                                         double* __restrict__ part) {
  __shared__ double sA[MOM_ROWS * MOM_LDS_LD];
  __shared__ double sB[MOM_ROWS * MOM_LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int npairs = nblk * (nblk + 1) / 2;
  const int pair = blockIdx.x % npairs, slice = blockIdx.x / npairs;
  int t = pair, I = 0;
  while (t >= nblk - I) { t -= nblk - I; ++I; }
  const int J = I + t;
  const bool diag = I == J;
  const int64_t r_begin = (int64_t)slice * rows_per_slice;
  const int64_t r_end = r_begin + rows_per_slice < N ? r_begin + rows_per_slice : N;

  // staging: 32 rows x 64 columns per block = 2048 doubles, 8 per thread: thread -> (row = q*4 + tid/64, col = tid%64)
  const int scol = tid & 63, srow0 = tid >> 6;
  const bool ca = I * MOM_BLK + scol < C, cb = J * MOM_BLK + scol < C;
  const double* za = Z + (I * MOM_BLK + (ca ? scol : 0));
  const double* zb = Z + (J * MOM_BLK + (cb ? scol : 0));
  double ra[8], rb[8];
  auto fetch = [&](int64_t r0) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int64_t r = r0 + q * 4 + srow0;
      const bool ok = r < r_end;
      const int64_t rr = ok ? r : r_begin;
      ra[q] = (ok && ca) ? za[rr * ldz] : 0.0;
      if (!diag) rb[q] = (ok && cb) ? zb[rr * ldz] : 0.0;
    }
  };
  mv4d acc[4][4];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int v = 0; v < 4; ++v) acc[u][v] = (mv4d){0.0, 0.0, 0.0, 0.0};

  if (r_begin < r_end) fetch(r_begin);
  for (int64_t r0 = r_begin; r0 < r_end; r0 += MOM_ROWS) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      sA[(q * 4 + srow0) * MOM_LDS_LD + scol] = ra[q];
      if (!diag) sB[(q * 4 + srow0) * MOM_LDS_LD + scol] = rb[q];
    }
    __syncthreads();
    if (r0 + MOM_ROWS < r_end) fetch(r0 + MOM_ROWS);      // next chunk in flight while this one is multiplied
    const double* pB = diag ? sA : sB;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int row = wave * 8 + ks * 4 + lk;
      double av[4], bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = sA[row * MOM_LDS_LD + 16 * u + li];
        bv[u] = pB[row * MOM_LDS_LD + 16 * u + li];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[v], acc[u][v], 0, 0, 0);
    }
    __syncthreads();
  }
  // the four waves hold partial sums over disjoint rows: add them in wave order through LDS
  __shared__ double sR[MOM_BLK * MOM_BLK];
  double* red = sR;
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int v = 0; v < 4; ++v)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
            const int i = 16 * u + lk + 4 * r, j = 16 * v + li;
            if (w == 0) red[i * MOM_BLK + j] = acc[u][v][r];
            else red[i * MOM_BLK + j] += acc[u][v][r];
          }
    }
    __syncthreads();
  }
  double* dst = part + ((size_t)slice * npairs + pair) * (MOM_BLK * MOM_BLK);
  for (int e = tid; e < MOM_BLK * MOM_BLK; e += 256) dst[e] = red[e];
}

// M[i][j] = sum over slices (in slice order) of the block partials; both triangles.
__global__ __launch_bounds__(256) void moments_reduce_kernel(const double* __restrict__ part, int nblk, int nslices, int C,
                                                             double* __restrict__ M, int64_t ldm) {
  const int npairs = nblk * (nblk + 1) / 2;
  const int pair = blockIdx.x;
  int t = pair, I = 0;
  while (t >= nblk - I) { t -= nblk - I; ++I; }
  const int J = I + t;
  // (four entries per thread at a time: their loads of a slice are independent; each entry's slices are added in slice order)
  for (int e0 = threadIdx.x; e0 < MOM_BLK * MOM_BLK; e0 += 1024) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int sl = 0; sl < nslices; ++sl) {
      const double* ps = part + ((size_t)sl * npairs + pair) * (MOM_BLK * MOM_BLK) + e0;
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += ps[256 * q];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + 256 * q;
      const int i = I * MOM_BLK + e / MOM_BLK, j = J * MOM_BLK + e % MOM_BLK;
      if (i < C && j < C) {
        if (I != J) { M[(size_t)i * ldm + j] = s[q]; M[(size_t)j * ldm + i] = s[q]; }
        else if (i <= j) { M[(size_t)i * ldm + j] = s[q]; M[(size_t)j * ldm + i] = s[q]; }   // diagonal block: upper triangle decides
      }
    }
  }
}

// thetabar[c] = mean_s theta[s][c]: four partial sums over interleaved samples per column, combined in a fixed order.
__global__ __launch_bounds__(256) void moments_mean_kernel(const double* __restrict__ theta, int S, int ldt, int D,
                                                           double* __restrict__ tbar) {
  __shared__ double part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  double m = 0.0;
  if (c < D) {
#pragma unroll 8
    for (int u = q; u < S; u += 4) m += theta[(size_t)u * ldt + c];
  }
  part[q][threadIdx.x & 63] = m;
  __syncthreads();
  if (q == 0 && c < D) tbar[c] = (((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x]) / (double)S;
}

template <bool AL>
__global__ __launch_bounds__(256) void moments_quad_kernel(MqArgs q) { moments_quad_body<AL>(q, blockIdx.x, gridDim.x); }

void bcx_project_set_error(const std::string& msg);   // proj.hip: the message bcx_project_last_error() returns
#define MOM_HIP(call)                                                             \
  do {                                                                            \
    hipError_t _e = (call);                                                       \
    if (_e != hipSuccess) {                                                       \
      bcx_project_set_error(std::string(#call) + ": " + hipGetErrorString(_e));   \
      return BCX_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

static void moments_plan(int64_t N, int C, int* nblk, int* nslices, int64_t* rows_per_slice) {
  *nblk = (C + MOM_BLK - 1) / MOM_BLK;
  const int npairs = *nblk * (*nblk + 1) / 2;
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int64_t chunks = (N + MOM_ROWS - 1) / MOM_ROWS;
  // two workgroups per CU are resident; two rounds of them, and never fewer than 8 chunks per slice
  int64_t sl = std::max<int64_t>(1, (int64_t)(4 * cus) / npairs);
  sl = std::min<int64_t>(sl, std::max<int64_t>(1, chunks / 8));
  *rows_per_slice = std::max<int64_t>(MOM_ROWS, (chunks + sl - 1) / sl * MOM_ROWS);      // (N = 0: one empty slice, M = 0)
  *nslices = (int)std::max<int64_t>(1, (N + *rows_per_slice - 1) / *rows_per_slice);
}

// Bytes of scratch bcx_project_moments needs for an N x C matrix (the block partials of every row slice).
extern "C" int64_t bcx_project_moments_scratch_bytes(int64_t N, int32_t C) {
  if (N < 0 || C < 1 || C > MOM_MAX_COLS) return -1;
  int nblk, nslices; int64_t rps;
  moments_plan(N, C, &nblk, &nslices, &rps);
  return (int64_t)nslices * (nblk * (nblk + 1) / 2) * MOM_BLK * MOM_BLK * (int64_t)sizeof(double);
}

// M_dev (C x ldm doubles, both triangles) = Z^T Z over the N rows of Z_dev (N x ldz doubles, C = D + 1 columns used).
extern "C" int bcx_project_moments(void* stream, const void* Z_dev, int64_t N, int64_t ldz, int32_t C, void* M_dev, int64_t ldm,
                                   void* work_dev, int64_t work_bytes) {
  if (!Z_dev || !M_dev || !work_dev || N < 0 || C < 1 || C > MOM_MAX_COLS || ldz < C || ldm < C) {
    bcx_project_set_error("bcx_project_moments: bad arguments (1 <= columns <= 1024)");
    return BCX_ERR_ARG;
  }
  if (work_bytes < bcx_project_moments_scratch_bytes(N, C)) { bcx_project_set_error("bcx_project_moments: scratch too small"); return BCX_ERR_ARG; }
  int nblk, nslices; int64_t rps;
  moments_plan(N, C, &nblk, &nslices, &rps);
  const int npairs = nblk * (nblk + 1) / 2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(moments_kernel, dim3(npairs * nslices), dim3(256), 0, st, (const double*)Z_dev, N, ldz, (int)C, nblk, nslices,
                     rps, (double*)work_dev);
  hipLaunchKernelGGL(moments_reduce_kernel, dim3(npairs), dim3(256), 0, st, (const double*)work_dev, nblk, nslices, (int)C,
                     (double*)M_dev, ldm);
  MOM_HIP(hipGetLastError());
  return BCX_OK;
}

// ---- optimize(): G = V V^T over the k active rows (nnls.hip; snnls.py:82-97) ------------------------------------------------
// G[i][j] = rows[i] . rows[j] for i, j < k (rows: k rows of d doubles, row stride ld), both triangles, leading dimension ldg.
// One workgroup per 64 x 64 block of the upper triangle (and slice of the row length, see gram_plan); its four waves own
// the block's four 32 x 32 quadrants (2 x 2 tiles of v_mfma_f64_16x16x4_f64) -- nothing is combined across waves.  The
// row length advances in chunks of 32: both operand blocks (64 rows x 32 values, one of them on the diagonal) are staged
// through LDS transposed (k-major, so that an MFMA operand is a conflict-free 8-byte read per lane: rows 80 doubles apart
// put lane groups lk and lk + 1 on disjoint bank halves), the next chunk's 16-byte global loads are in flight while this
// one is multiplied.  With one slice the block goes straight to G (mirror image included); with more the slices leave
// partial blocks that moments_reduce_kernel adds in slice order.
template <bool DIRECT>
__global__ __launch_bounds__(256, 2) void gram_tile_kernel(const double* __restrict__ V, int k, int d, int64_t ld, int nblk,
                                                           int64_t len_per_slice, int pshift, double* __restrict__ part,
                                                           double* __restrict__ G, int64_t ldg) {
  __shared__ double sA[MOM_ROWS * MOM_LDS_LD];
  __shared__ double sB[MOM_ROWS * MOM_LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int npairs = nblk * (nblk + 1) / 2;
  // Workgroup -> block.  A 64 x 64 block re-reads 1 KiB of rows per 8 Ki flops, and the blocks of one launch share rows only
  // through the L2 of their XCD (4 MiB; the k x d matrix of a large support does not fit): in block-row order every
  // workgroup streams its two row panels from beyond L2.  The blocks can therefore be dealt out in P x P patches, one patch per
  // XCD at a time: workgroups b, b + 8, ... sit on the same XCD (round-robin dispatch: a speed assumption, not a correctness
  // one) and take the P^2 blocks of a patch, which share P + P row panels.  Patch positions below the diagonal or beyond
  // the matrix stay idle.  (It matters little -- gram_plan: the panels come out of the infinity cache fast enough.  Large
  // supports run at 41-43 TFLOP/s; the same loop on v_mfma_f64_4x4x4_4b_f64 with a replicated operand -- 10 LDS reads per
  // k-step instead of 4 -- measured the same to 1 %, so it is the staging through LDS, not the instruction form, that bounds it.)
  const int P = 1 << pshift, PP = P * P;                           // patch edge in blocks (gram_plan: 8 for large supports ... 1)
  const int nsb = (nblk + P - 1) >> pshift, nsu = nsb * (nsb + 1) / 2, per_slice = (nsu + 7) / 8 * PP;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int slice = slot / per_slice, s2 = slot - slice * per_slice;
  const int sb = (s2 / PP) * 8 + ((xcd + slice) & 7), tin = s2 % PP;   // (rotated by the slice: the last, partial group of patches lands on different XCDs)
  if (sb >= nsu) return;
  int t = sb, SI = 0;
  while (t >= nsb - SI) { t -= nsb - SI; ++SI; }
  const int I = SI * P + (tin >> pshift), J = (SI + t) * P + (tin & (P - 1));
  if (I >= nblk || J >= nblk || I > J) return;
  const int pair = I * nblk - I * (I - 1) / 2 + (J - I);           // block-row order: what moments_reduce_kernel indexes by
  const bool diag = I == J;
  const int64_t c_begin = (int64_t)slice * len_per_slice;
  const int64_t c_end = c_begin + len_per_slice < d ? c_begin + len_per_slice : d;
  // staging: thread -> (row of the block = tid / 4, values 2 (tid % 4) + 8 j (+1), j = 0..3 of the chunk): 16-byte pieces along
  // the stored rows, the four threads of a row read 64 contiguous bytes per j (8-byte loads when rows are not 16-byte aligned)
  const int srow = tid >> 2, sk0 = 2 * (tid & 3);
  const bool ra_ok = I * MOM_BLK + srow < k, rb_ok = J * MOM_BLK + srow < k;
  const double* za = V + (size_t)(I * MOM_BLK + (ra_ok ? srow : 0)) * ld;
  const double* zb = V + (size_t)(J * MOM_BLK + (rb_ok ? srow : 0)) * ld;
  const bool al16 = (ld & 1) == 0 && ((uintptr_t)V & 15) == 0;
  // (two register sets -- the loads of chunks c + 1 and c + 2 in flight -- measured slower: 163 VGPRs, one workgroup fewer per CU)
  double ra[8], rb[8];
  auto fetch = [&](int64_t c0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t c = c0 + sk0 + 8 * j;                        // even: c0 is a multiple of 32
      const bool ok0 = c < c_end, ok1 = c + 1 < c_end;
      if (al16 && ok1) {
        const double2 va = *(const double2*)(za + c);
        ra[2 * j] = ra_ok ? va.x : 0.0; ra[2 * j + 1] = ra_ok ? va.y : 0.0;
        if (!diag) { const double2 vb = *(const double2*)(zb + c); rb[2 * j] = rb_ok ? vb.x : 0.0; rb[2 * j + 1] = rb_ok ? vb.y : 0.0; }
      } else {
        const int64_t q0 = ok0 ? c : c_begin, q1 = ok1 ? c + 1 : c_begin;
        ra[2 * j] = (ok0 && ra_ok) ? za[q0] : 0.0; ra[2 * j + 1] = (ok1 && ra_ok) ? za[q1] : 0.0;
        if (!diag) { rb[2 * j] = (ok0 && rb_ok) ? zb[q0] : 0.0; rb[2 * j + 1] = (ok1 && rb_ok) ? zb[q1] : 0.0; }
      }
    }
  };
  mv4d acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[u][v] = (mv4d){0.0, 0.0, 0.0, 0.0};
  const int ar = 32 * (wave >> 1), bc = 32 * (wave & 1);     // this wave's quadrant of the block
  if (c_begin < c_end) fetch(c_begin);
  for (int64_t c0 = c_begin; c0 < c_end; c0 += MOM_ROWS) {
    // (the staging stores of a wave go to k-slots 2 p + const, p = lane & 3: 80-double rows put all four on the same banks;
    // the 16-column halves of every second k-slot pair are swapped -- column ^ 16 -- so the four fall on both bank halves
    // twice, which is what a 64-lane 8-byte store costs anyway; the reads undo it by taking the other tile)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int kk = sk0 + 8 * (q >> 1) + (q & 1);
      const int sc = srow ^ (((kk >> 1) & 1) << 4);
      sA[kk * MOM_LDS_LD + sc] = ra[q];
      if (!diag) sB[kk * MOM_LDS_LD + sc] = rb[q];
    }
    __syncthreads();
    if (c0 + MOM_ROWS < c_end) fetch(c0 + MOM_ROWS);        // next chunk in flight while this one is multiplied
    const double* pB = diag ? sA : sB;
#pragma unroll
    for (int ks = 0; ks < MOM_ROWS / 4; ++ks) {
      const int kk = ks * 4 + lk;
      const int sw = ((kk >> 1) & 1) << 4;                   // (= (lk >> 1) << 4: lane groups lk = 2, 3 read the swapped halves)
      double av[2], bv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        av[u] = sA[kk * MOM_LDS_LD + ((ar + 16 * u + li) ^ sw)];
        bv[u] = pB[kk * MOM_LDS_LD + ((bc + 16 * u + li) ^ sw)];
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[v], acc[u][v], 0, 0, 0);
    }
    __syncthreads();
  }
  // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int bi = ar + 16 * u + lk + 4 * r, bj = bc + 16 * v + li;
        if (DIRECT) {
          const int row = I * MOM_BLK + bi, col = J * MOM_BLK + bj;
          if (row < k && col < k && (!diag || row <= col)) {      // diagonal block: the upper triangle decides, as in the reduce kernel
            G[(size_t)row * ldg + col] = acc[u][v][r];
            if (row != col) G[(size_t)col * ldg + row] = acc[u][v][r];
          }
        } else {
          part[((size_t)slice * npairs + pair) * (MOM_BLK * MOM_BLK) + bi * MOM_BLK + bj] = acc[u][v][r];
        }
      }
}

// Slices of the row length: one (the block goes straight to G) when the blocks alone give every CU two workgroups or the
// rows are short; otherwise enough of them for ~3 workgroups per CU, never fewer than 4 chunks of 32 per slice.
static void gram_plan(int k, int d, int* nblk, int* nslices, int64_t* len_per_slice, int* pshift) {
  *nblk = (k + MOM_BLK - 1) / MOM_BLK;
  const int npairs = *nblk * (*nblk + 1) / 2;
  int cus = 256, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  const int64_t chunks = ((int64_t)d + MOM_ROWS - 1) / MOM_ROWS;
  static const int force = [] { const char* e = bcx_dev_env("BCX_GRAM_SLICES"); return e ? atoi(e) : 0; }();    // dev knobs
  static const int force_p = [] { const char* e = bcx_dev_env("BCX_GRAM_PATCH"); return e ? atoi(e) : -1; }();
  int64_t sl = 1;
  if (npairs < 2 * cus) {
    sl = ((int64_t)3 * cus + npairs - 1) / npairs;
    sl = std::max<int64_t>(1, std::min<int64_t>(sl, chunks / 4));
  }
  if (force > 0) sl = std::max<int64_t>(1, std::min<int64_t>(force, chunks));
  *len_per_slice = std::max<int64_t>(MOM_ROWS, (chunks + sl - 1) / sl * MOM_ROWS);
  *nslices = (int)std::max<int64_t>(1, ((int64_t)d + *len_per_slice - 1) / *len_per_slice);
  // patch edge (measured, k x d = 1497 x 1024 / 2048 x 2048 / 4096 x 1024, us per call: P = 1: 90 / 290 / 420, 2: 91 / 283 / 411,
  // 4: 110 / 280 / 409, 8: 120 / 344 / 441 -- the row panels mostly come out of the infinity cache either way, large patches
  // leave XCDs idle at the end)
  *pshift = *nblk >= 32 ? 1 : 0;
  if (force_p >= 0 && force_p <= 3) *pshift = force_p;
}
int64_t bcx_gram_sk_scratch_bytes(int k, int d);      // gram.hip: the chip-balanced kernel (16-byte aligned rows, k >= 192)
int bcx_gram_sk(hipStream_t st, const double* rows, int k, int d, int64_t ld, double* G, int64_t ldg, double* work);
int64_t bcx_gram_rows_scratch_bytes(int k, int d) {
  int nblk, nslices, ps; int64_t lps;
  gram_plan(k, d, &nblk, &nslices, &lps, &ps);
  const int64_t tiled = nslices == 1 ? 8 : (int64_t)nslices * (nblk * (nblk + 1) / 2) * MOM_BLK * MOM_BLK * (int64_t)sizeof(double);
  return std::max<int64_t>(tiled, bcx_gram_sk_scratch_bytes(k, d));     // (which kernel runs depends on the rows' alignment)
}
int bcx_gram_rows(hipStream_t st, const double* rows, int k, int d, int64_t ld, double* G, int64_t ldg, double* work) {
  const int sk = bcx_gram_sk(st, rows, k, d, ld, G, ldg, work);
  if (sk <= 0) return sk;
  int nblk, nslices, ps; int64_t lps;
  gram_plan(k, d, &nblk, &nslices, &lps, &ps);
  const int npairs = nblk * (nblk + 1) / 2;
  const int P = 1 << ps, nsb = (nblk + P - 1) / P, nsu = nsb * (nsb + 1) / 2;
  const unsigned grid = (unsigned)((nsu + 7) / 8 * P * P * 8 * nslices);      // (gram_tile_kernel: patches of blocks, one per XCD at a time)
  if (nslices == 1) {
    hipLaunchKernelGGL(gram_tile_kernel<true>, dim3(grid), dim3(256), 0, st, rows, k, d, ld, nblk, lps, ps, (double*)nullptr, G, ldg);
  } else {
    hipLaunchKernelGGL(gram_tile_kernel<false>, dim3(grid), dim3(256), 0, st, rows, k, d, ld, nblk, lps, ps, work, G, ldg);
    hipLaunchKernelGGL(moments_reduce_kernel, dim3(npairs), dim3(256), 0, st, (const double*)work, nblk, nslices, k, G, ldg);
  }
  return hipGetLastError() == hipSuccess ? BCX_OK : BCX_ERR_HIP;
}
// The same through the C ABI (include/bcx.h): the dense re-weight's Gram as an operator of its own.
extern "C" int64_t bcx_gram_scratch_bytes(int32_t k, int32_t d) {
  if (k < 1 || d < 1 || k > BCX_GRAM_MAX_ROWS) return -1;
  return bcx_gram_rows_scratch_bytes(k, d);
}
extern "C" int bcx_gram(void* stream, const void* rows_dev, int32_t k, int32_t d, int64_t ld, void* G_dev, int64_t ldg,
                        void* work_dev, int64_t work_bytes) {
  if (!rows_dev || !G_dev || !work_dev || k < 1 || d < 1 || k > BCX_GRAM_MAX_ROWS || ld < d || ldg < k) {
    bcx_project_set_error("bcx_gram: bad arguments");
    return BCX_ERR_ARG;
  }
  if (work_bytes < bcx_gram_rows_scratch_bytes(k, d)) { bcx_project_set_error("bcx_gram: scratch too small"); return BCX_ERR_ARG; }
  const int rc = bcx_gram_rows((hipStream_t)stream, (const double*)rows_dev, k, d, ld, (double*)G_dev, ldg, (double*)work_dev);
  if (rc != BCX_OK) bcx_project_set_error("bcx_gram: kernel launch failed");
  return rc;
}

extern "C" int bcx_gram_check(void* stream, const void* work_dev) {
  if (!work_dev) { bcx_project_set_error("bcx_gram_check: bad arguments"); return BCX_ERR_ARG; }
  // (every call so far that used this scratch: the word only ever holds the number of a call that gave up a wait)
  const int t = bcx_gram_sk_timed_out((hipStream_t)stream, (const double*)work_dev, 0ull);
  if (t < 0) { bcx_project_set_error("bcx_gram_check: reading the status word failed"); return t; }
  if (t) { bcx_project_set_error("bcx_gram: a workgroup timed out waiting for a peer's partial tile; G is not valid"); return BCX_ERR_TIMEOUT; }
  return BCX_OK;
}

// Doubles of scratch bcx_project_colsum_moments needs: the (column tile, sample) partials, thetabar, the arrival counter.
extern "C" int64_t bcx_project_colsum_moments_scratch_bytes(int32_t D, int32_t S) {
  if (D < 1 || D >= MOM_MAX_COLS || S < 1) return -1;
  int nct, Spad;
  mq_plan(D, S, &nct, &Spad);
  return ((int64_t)nct * Spad + 1 + (D + 1) / 2 * 2) * (int64_t)sizeof(double);
}

// colsum_dev[s] = sum_n vecs[n][s] of the linear-regression projection of the data whose moments are M_dev (features in
// rows/columns [0, D), response in row/column ycol), for the S parameter rows of theta_dev.  work_dev:
// bcx_project_colsum_moments_scratch_bytes(D, S) bytes, ZERO before the first call (the kernel leaves them zero: a zero word
// is a partial sum not written yet, csrc/moments_quad.h).
// tbar_dev: the point the quadratic is expanded around, D doubles (any point near the draws serves -- the expansion is exact;
// csrc/svi.hip leaves the mean of the draws it makes), or NULL: their mean is formed here.
extern "C" int bcx_project_colsum_moments_at(void* stream, const void* M_dev, int64_t ldm, int32_t D, int32_t ycol,
                                             const void* theta_dev, int32_t S, int32_t ldt, double sigsq, void* colsum_dev,
                                             void* work_dev, const void* tbar_dev) {
  if (!M_dev || !theta_dev || !colsum_dev || !work_dev || D < 1 || D >= MOM_MAX_COLS || ycol < 0 || ycol >= ldm || ldm < D ||
      S < 1 || ldt < D || !(sigsq > 0.0)) {
    bcx_project_set_error("bcx_project_colsum_moments: bad arguments");
    return BCX_ERR_ARG;
  }
  int nct, Spad;
  mq_plan(D, S, &nct, &Spad);
  double* work = (double*)work_dev;
  const double* tbar = (const double*)tbar_dev;
  hipStream_t st = (hipStream_t)stream;
  if (!tbar) {
    double* own = work + (size_t)nct * Spad + 1;
    hipLaunchKernelGGL(moments_mean_kernel, dim3((D + 63) / 64), dim3(256), 0, st, (const double*)theta_dev, (int)S, (int)ldt, (int)D, own);
    tbar = own;
  }
  const int tiles = nct * (Spad / 16);
  const bool al = ((uintptr_t)theta_dev % 16 == 0) && ldt % 2 == 0;
  static const int mqdbg = bcx_dev_env("BCX_MQ_DBG") != nullptr;     // dev: time stamps of the last workgroup (needs tbar_dev)
  MqArgs q;
  q.M = (const double*)M_dev; q.ldm = ldm; q.D = D; q.ycol = ycol; q.theta = (const double*)theta_dev; q.S = S; q.ldt = ldt;
  q.tbar = tbar; q.sigsq = sigsq; q.colsum = (double*)colsum_dev; q.work = work; q.nct = nct; q.Spad = Spad; q.dbg = mqdbg;
  if (al) hipLaunchKernelGGL(moments_quad_kernel<true>, dim3(tiles), dim3(256), 0, st, q);
  else hipLaunchKernelGGL(moments_quad_kernel<false>, dim3(tiles), dim3(256), 0, st, q);
  MOM_HIP(hipGetLastError());
  return BCX_OK;
}

extern "C" int bcx_project_colsum_moments(void* stream, const void* M_dev, int64_t ldm, int32_t D, int32_t ycol,
                                          const void* theta_dev, int32_t S, int32_t ldt, double sigsq, void* colsum_dev,
                                          void* work_dev) {
  return bcx_project_colsum_moments_at(stream, M_dev, ldm, D, ycol, theta_dev, S, ldt, sigsq, colsum_dev, work_dev, nullptr);
}
