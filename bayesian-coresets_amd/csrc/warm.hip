// warm.hip -- warm start of optimize() (snnls.py:82-97: w[active] = nnls(A[:, active], b)).
//
// Lawson-Hanson from the empty passive set (omp_lh.hip optimize_lh_kernel) lets the columns of the support enter one at a
// time: k dependent steps of three grid barriers each (46 ms at k = 1497, d = 1024).  But optimize() is called on a support
// that already carries positive weights, and the NNLS minimiser mostly keeps it: here the passive set STARTS as the largest
// independent subset of the support (weights in descending order, a column that is numerically dependent on the ones before
// it is left out: the incremental kernel's pivot rule with a wider margin), with the explicit inverse H of its Gram block, and
// the incremental kernel only does the pivots that are left (columns whose least-squares weight is not positive leave,
// candidates with a positive dual enter: tens of steps instead of a thousand).  The NNLS minimiser is unique for
// independent columns, so the result is the reference's; for k > d (dependent columns) any vertex with the minimal error
// is a minimiser, as with Lawson-Hanson itself.
//
//   G~ = P G P^T              support permuted into weight order                       warm_order / warm_gather_kernel
//   G~ = L L^T                right-looking blocked Cholesky, 64-column panels; a pivot <= 1e-9 G_jj rejects the column
//                             (its row and column of L become the identity's)           warm_diag / warm_panel / warm_update_kernel
//   Y  = L^-T                 by halving, two products per range of blocks              warm_gemm_kernel (warm_invert)
//   H  = Y Y^T                on the fp64 matrix cores: the Gram kernel of csrc/gram.hip -- k^3 of the ~(5/3) k^3 flops
//   hinv = H[kept, kept]      compacted in passive-set order, low words zero             warm_scan / warm_compact_kernel
// H is a plain-double inverse (error ~ cond(G) eps); the incremental kernel carries it on in double-double, and its
// closing data-space Newton step (x += H V_P (b - V_P^T x)) measures what that cost: if the check fails, optimize() runs
// again from the empty set.
#include <algorithm>
#include "bcx_internal.h"
#include "dev_util.h"
#include "nnls_common.h"

#define WM_NB 64

struct WarmBufs {
  double* A;        // kp x kp: permuted Gram block, then L (lower), then H
  double* Y;        // kp x kp: L^-T (upper), rows zero-padded
  double* T;        // (kp / 64) x 64 x 64: inverses of the diagonal blocks of L
  double* diag0;    // kp: diagonal of the permuted Gram block
  int32_t* perm;    // kp: position in weight order -> slot
  int32_t* rej;     // kp: 1 = not in the passive set (no weight, or numerically dependent)
  int32_t* keptq;   // kp: passive position -> position in weight order
  int32_t* p;       // [0]: size of the passive set, [1]: columns accepted by the factorisation so far
};

// perm: slots ordered by (weight desc, slot asc); slots without weight come last and are rejected from the start.
// (Ordering by weight x row norm -- the size of the column's term in A w -- was tried: a quarter fewer pivots on a saved
// N = 200k support, but 536 instead of 282 entering columns on the N = 1M, k = 1497, d = 1024 one.  Plain weights stay.)
__global__ __launch_bounds__(256) void warm_order_kernel(const double* __restrict__ w, const double* __restrict__ nrm, int k, int kp, WarmBufs wb) {
  extern __shared__ double sw[];               // (every workgroup holds all k weights and ranks its own 256 slots)
  for (int j = threadIdx.x; j < k; j += blockDim.x) sw[j] = w[j];
  __syncthreads();
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < kp; j += gridDim.x * blockDim.x) {
    if (j < k) {
      const double wj = sw[j];
      int rank = 0;
      for (int i = 0; i < k; ++i) rank += (sw[i] > wj) || (sw[i] == wj && i < j);
      wb.perm[rank] = j;
      wb.rej[rank] = !(wj > 0.0);
    } else {
      wb.perm[j] = -1;
      wb.rej[j] = 1;
    }
  }
}

// A[q][r] = G[perm[q]][perm[r]] (identity in the padding), diag0
__global__ __launch_bounds__(256) void warm_gather_kernel(const double* __restrict__ G, int64_t ldg, int k, int kp, WarmBufs wb) {
  const int q = blockIdx.y;
  const int sq = wb.perm[q];
  for (int r = blockIdx.x * 256 + threadIdx.x; r < kp; r += gridDim.x * 256) {
    const int sr = wb.perm[r];
    double v = (q == r) ? 1.0 : 0.0;
    if (sq >= 0 && sr >= 0) v = G[(size_t)sq * ldg + sr];
    wb.A[(size_t)q * kp + r] = v;
    if (q == r) wb.diag0[q] = v;
  }
}

// Diagonal block jj: unblocked Cholesky with the rejection rule, then T = L11^-1 (also written, transposed, as the diagonal
// block of Y = L^-T).  ONE WAVE, everything in registers: lane r owns row r of the block (64 doubles), the loops over columns
// are fully unrolled so that every register index is static; the only exchange is the current column, which goes through
// LDS and is read back as broadcasts (all lanes the same address).  (Versions that kept the block in LDS -- four waves with
// three barriers per column, or one wave walking its row -- spent 110-160 us per panel on LDS latency, most of it in the
// serial dot products of the triangular inverse.)
// Rejection: not a member, pivot <= WM_PIVOT_TOL x the column's own diagonal entry, or the passive set already holds d columns
// (more cannot be independent; in plain doubles the pivot of a dependent column is rounding noise, which may exceed any
// fixed small fraction -- the incremental kernel's own test, in double-double, decides about such a column later).
#define WM_PIVOT_TOL 1e-9
__global__ __launch_bounds__(64) void warm_diag_kernel(int kp, int jj, int d, WarmBufs wb) {
  __shared__ double scol[WM_NB];               // the current column of L
  __shared__ double sL[WM_NB][WM_NB + 1];      // L11 for the inverse
  const int r = threadIdx.x;
  double a[WM_NB];
  // the block comes in row by row (lane = column: 512 contiguous bytes per load) and is handed to its row owners through LDS
  for (int rr = 0; rr < WM_NB; ++rr) sL[rr][r] = wb.A[(size_t)(jj + rr) * kp + jj + r];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < WM_NB; ++c) a[c] = c <= r ? sL[r][c] : 0.0;
  __syncthreads();
  int my_rej = wb.rej[jj + r];
  const double my_d0 = wb.diag0[jj + r];
  int accepted = wb.p[1];                      // columns accepted by the panels before this one
#pragma unroll
  for (int c = 0; c < WM_NB; ++c) {
    const double piv = __shfl(a[c], c, 64);
    const int rej_c = __shfl(my_rej, c, 64);
    const double d0 = __shfl(my_d0, c, 64);
    const bool rej = rej_c || accepted >= d || !(piv > WM_PIVOT_TOL * d0);      // (wave-uniform)
    if (rej) {
      if (r > c) a[c] = 0.0;                   // column c
      if (r == c) {                            // row c
#pragma unroll
        for (int t = 0; t < c; ++t) a[t] = 0.0;
        a[c] = 1.0; my_rej = 1;
      }
    } else {
      ++accepted;
      const double l = sqrt(piv);
      if (r == c) a[c] = l;
      else if (r > c) a[c] = a[c] / l;
      scol[r] = a[c];
      __syncthreads();
      const double lrc = a[c];
#pragma unroll
      for (int r2 = c + 1; r2 < WM_NB; ++r2) a[r2] = (r >= r2) ? a[r2] - lrc * scol[r2] : a[r2];
      __syncthreads();
    }
  }
#pragma unroll
  for (int c = 0; c < WM_NB; ++c) sL[r][c] = a[c];
  __syncthreads();
  for (int rr = 0; rr < WM_NB; ++rr) wb.A[(size_t)(jj + rr) * kp + jj + r] = sL[rr][r];      // (upper part of the block: zeros)
  // T = L11^-1: lane j forms column j by forward substitution -- x[t] = 0 above the diagonal, so the sums can start at 0 and
  // every operand L11[i][t] is a broadcast read
  double x[WM_NB];
  const int j = r;
#pragma unroll
  for (int i = 0; i < WM_NB; ++i) {
    double acc[4] = {0.0, 0.0, 0.0, 0.0};       // (four partial sums: the dependent chain of a row is a quarter as long)
#pragma unroll
    for (int t = 0; t < i; ++t) acc[t & 3] += sL[i][t] * x[t];
    const double lii = sL[i][i];
    x[i] = i < j ? 0.0 : (i == j ? 1.0 / lii : -((acc[0] + acc[1]) + (acc[2] + acc[3])) / lii);
  }
  // T[i][j] = x[i] (lane j): through LDS so that the stores are rows (T) and columns-as-rows (Y_jj = T^T) of 512 bytes
  __syncthreads();
#pragma unroll
  for (int i = 0; i < WM_NB; ++i) sL[i][j] = x[i];
  __syncthreads();
  double* Tj = wb.T + (size_t)(jj / WM_NB) * WM_NB * WM_NB;
  for (int rr = 0; rr < WM_NB; ++rr) {
    Tj[rr * WM_NB + r] = sL[rr][r];                                      // T row rr
    wb.Y[(size_t)(jj + rr) * kp + jj + r] = sL[r][rr];                   // Y row rr = T column rr
  }
  wb.rej[jj + r] = my_rej;
  if (r == 0) wb.p[1] = accepted;
}

// L21 = A21 T^T for the 64 rows of this workgroup (rows below the diagonal block); rejected columns are zero.
__global__ __launch_bounds__(256) void warm_panel_kernel(int kp, int jj, WarmBufs wb) {
  __shared__ double sT[WM_NB][WM_NB + 1];
  __shared__ double sR[WM_NB][WM_NB + 1];
  __shared__ int srej[WM_NB];
  const int tid = threadIdx.x;
  const int r0 = jj + WM_NB + blockIdx.x * WM_NB;
  const double* Tj = wb.T + (size_t)(jj / WM_NB) * WM_NB * WM_NB;
  for (int e = tid; e < WM_NB * WM_NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    sT[r][c] = Tj[e];
    sR[r][c] = wb.A[(size_t)(r0 + r) * kp + jj + c];
  }
  if (tid < WM_NB) srej[tid] = wb.rej[jj + tid];
  __syncthreads();
  // thread -> row tid / 4, columns (tid % 4) + 4 u
  const int r = tid >> 2;
  for (int u = 0; u < 16; ++u) {
    const int c = (tid & 3) + 4 * u;
    double acc = 0.0;
    for (int t = 0; t <= c; ++t) acc += sR[r][t] * sT[c][t];
    wb.A[(size_t)(r0 + r) * kp + jj + c] = srej[c] ? 0.0 : acc;
  }
}

// A22[bi][bj] -= L21[bi] L21[bj]^T over the blocks bi >= bj of the trailing matrix
__global__ __launch_bounds__(256) void warm_update_kernel(int kp, int jj, WarmBufs wb) {
  __shared__ double sI[WM_NB][WM_NB + 1];
  __shared__ double sJ[WM_NB][WM_NB + 1];
  const int tid = threadIdx.x;
  int t = blockIdx.x, bi = 0;
  while (t > bi) { t -= bi + 1; ++bi; }
  const int bj = t;
  const int ri = jj + WM_NB + bi * WM_NB, rj = jj + WM_NB + bj * WM_NB;
  for (int e = tid; e < WM_NB * WM_NB; e += 256) {
    const int r = e >> 6, c = e & 63;
    sI[r][c] = wb.A[(size_t)(ri + r) * kp + jj + c];
    sJ[r][c] = wb.A[(size_t)(rj + r) * kp + jj + c];
  }
  __syncthreads();
  // thread -> rows 4 (tid / 16) .. + 3, columns 4 (tid % 16) .. + 3
  const int rr = 4 * (tid >> 4), cc = 4 * (tid & 15);
  double acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
  for (int x = 0; x < WM_NB; ++x) {
    double ai[4], bj4[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) { ai[a] = sI[rr + a][x]; bj4[a] = sJ[cc + a][x]; }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] += ai[a] * bj4[b];
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int row = ri + rr + a, col = rj + cc + b;
      if (col <= row) wb.A[(size_t)row * kp + col] -= acc[a][b];
    }
}

// Rows and columns of rejected positions become the identity's (row entries left of the diagonal were formed before the
// position was rejected; the columns are already zero).  One workgroup per row.
__global__ __launch_bounds__(256) void warm_clean_kernel(int kp, WarmBufs wb) {
  const int q = blockIdx.x;
  const bool rq = wb.rej[q] != 0;
  for (int c = threadIdx.x; c < kp; c += 256) {
    double* e = wb.A + (size_t)q * kp + c;
    if (c > q) *e = 0.0;                                   // (the strict upper triangle held the symmetric copy of G~)
    else if (rq || wb.rej[c]) *e = (c == q) ? 1.0 : 0.0;
  }
}

// C = alpha A B for the block inversion below: A is M x K with strides (sa_i, sa_k), B is K x N with strides (sb_k, sb_j), C is
// M x N with row stride ldc.  64 x 64 output block per workgroup, 16-deep steps staged through LDS (any strides: the loads
// transpose as needed), each of the four waves a 32 x 32 quadrant as 2 x 2 tiles of v_mfma_f64_16x16x4_f64.
typedef double wm4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void warm_gemm_kernel(const double* __restrict__ A, int64_t sa_i, int64_t sa_k, const double* __restrict__ B,
                                                        int64_t sb_k, int64_t sb_j, double* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                        double alpha) {
  __shared__ double sAt[16][WM_NB + 8];      // [k][i]: rows 72 doubles apart put the four k of an MFMA step on disjoint banks
  __shared__ double sBt[16][WM_NB + 8];      // [k][j]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int i0 = blockIdx.y * WM_NB, j0 = blockIdx.x * WM_NB;
  const int wi = 32 * (wave >> 1), wj = 32 * (wave & 1);
  wm4d acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (wm4d){0.0, 0.0, 0.0, 0.0};
  // this thread's four elements of each operand per 16-deep step (consecutive threads along whichever index is contiguous in
  // memory); the next step's are in flight while this one is multiplied
  double ra[4], rb[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      const int kk = sa_k == 1 ? (e & 15) : (e >> 6), ii = sa_k == 1 ? (e >> 4) : (e & 63);
      const int gi = i0 + ii, gk = k0 + kk;
      ra[q] = (gi < M && gk < K) ? A[(size_t)gi * sa_i + (size_t)gk * sa_k] : 0.0;
      const int kb = sb_j == 1 ? (e >> 6) : (e & 15), jb = sb_j == 1 ? (e & 63) : (e >> 4);
      const int gj = j0 + jb, gkb = k0 + kb;
      rb[q] = (gj < N && gkb < K) ? B[(size_t)gkb * sb_k + (size_t)gj * sb_j] : 0.0;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = tid + 256 * q;
      const int kk = sa_k == 1 ? (e & 15) : (e >> 6), ii = sa_k == 1 ? (e >> 4) : (e & 63);
      sAt[kk][ii] = ra[q];
      const int kb = sb_j == 1 ? (e >> 6) : (e & 15), jb = sb_j == 1 ? (e & 63) : (e >> 4);
      sBt[kb][jb] = rb[q];
    }
    __syncthreads();
    if (k0 + 16 < K) fetch(k0 + 16);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double av[2], bv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) { av[u] = sAt[4 * ks + lk][wi + 16 * u + li]; bv[u] = sBt[4 * ks + lk][wj + 16 * u + li]; }
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[v], acc[u][v], 0, 0, 0);
    }
    __syncthreads();
  }
  // f64 C/D layout: column = lane & 15, row = (lane >> 4) + 4 * register
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gi = i0 + wi + 16 * u + lk + 4 * r, gj = j0 + wj + 16 * v + li;
        if (gi < M && gj < N) C[(size_t)gi * ldc + gj] = alpha * acc[u][v][r];
      }
}

// Y (kp x kp, the strict lower triangle) = 0: the Gram kernel reads whole rows
__global__ __launch_bounds__(256) void warm_zero_lower_kernel(int kp, WarmBufs wb) {
  const int q = blockIdx.x;
  const int cend = q / WM_NB * WM_NB;          // (the diagonal block itself was written whole by warm_diag_kernel)
  for (int c = threadIdx.x; c < cend; c += 256) wb.Y[(size_t)q * kp + c] = 0.0;
}

// keptq / plist / p: the passive set in weight order.  One workgroup.
__global__ __launch_bounds__(1024) void warm_scan_kernel(int k, int kp, WarmBufs wb, int32_t* plist) {
  __shared__ int cnt[1024];
  const int tid = threadIdx.x;
  const int per = (kp + 1023) / 1024;
  int mine = 0;
  for (int u = 0; u < per; ++u) { const int q = tid * per + u; if (q < k && !wb.rej[q]) ++mine; }
  cnt[tid] = mine;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int t = 0; t < 1024; ++t) { const int v = cnt[t]; cnt[t] = run; run += v; }
    wb.p[0] = run;
  }
  __syncthreads();
  int pos = cnt[tid];
  for (int u = 0; u < per; ++u) {
    const int q = tid * per + u;
    if (q < k && !wb.rej[q]) { wb.keptq[pos] = q; plist[pos] = wb.perm[q]; ++pos; }
  }
}

// hinv[a][b] = H[keptq[a]][keptq[b]], low words zero
__global__ __launch_bounds__(256) void warm_compact_kernel(int kp, WarmBufs wb, double* hinv, double* hlo, int64_t ldg) {
  const int p = wb.p[0];
  const int a = blockIdx.y;
  if (a >= p) return;
  const int qa = wb.keptq[a];
  for (int b = blockIdx.x * 256 + threadIdx.x; b < p; b += gridDim.x * 256) {
    hinv[(size_t)a * ldg + b] = wb.A[(size_t)qa * kp + wb.keptq[b]];
    hlo[(size_t)a * ldg + b] = 0.0;
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
static int warm_kp(int k) { return (k + WM_NB - 1) / WM_NB * WM_NB; }

size_t bcx_warm_bytes(int k) {
  const size_t kp = (size_t)warm_kp(k);
  return 2 * kp * kp * 8 + (kp / WM_NB) * WM_NB * WM_NB * 8 + kp * 8 + 3 * kp * 4 + 64 + 256;
}

// Y = L^-T = (L^T)^-1 by halving: for U = [[Ua, C], [0, Ub]] the inverse is [[Ya, -Ya C Yb], [0, Yb]].  The diagonal blocks
// of 64 are in place (warm_diag_kernel wrote T^T); a range [lo, hi) of blocks is the two halves, then the two products
// W = C Yb and Y[lo:mid, mid:hi] = -Ya W.  C = L[mid:hi, lo:mid]^T is read through its strides; W (a x b) is parked in the
// matching block of A's strict UPPER triangle, which nothing reads (warm_clean_kernel zeroed it; the Gram kernel overwrites all
// of A with H afterwards) and which no other range of the recursion touches.
static void warm_invert(bcx_solver* s, hipStream_t st, int kp, const WarmBufs& wb, int lo, int hi) {
  if (hi - lo <= 1) return;
  const int mid = lo + (hi - lo) / 2;
  warm_invert(s, st, kp, wb, lo, mid);
  warm_invert(s, st, kp, wb, mid, hi);
  const int a = (mid - lo) * WM_NB, b = (hi - mid) * WM_NB;
  const size_t o_lo = (size_t)lo * WM_NB, o_mid = (size_t)mid * WM_NB;
  double* W = wb.A + o_lo * kp + o_mid;
  hipLaunchKernelGGL(warm_gemm_kernel, dim3((b + WM_NB - 1) / WM_NB, (a + WM_NB - 1) / WM_NB), dim3(256), 0, st,
                     (const double*)(wb.A + o_mid * kp + o_lo), (int64_t)1, (int64_t)kp,          // C[i][k] = L[mid + k][lo + i]
                     (const double*)(wb.Y + o_mid * kp + o_mid), (int64_t)kp, (int64_t)1,         // Yb
                     W, (int64_t)kp, a, b, b, 1.0);
  hipLaunchKernelGGL(warm_gemm_kernel, dim3((b + WM_NB - 1) / WM_NB, (a + WM_NB - 1) / WM_NB), dim3(256), 0, st,
                     (const double*)(wb.Y + o_lo * kp + o_lo), (int64_t)kp, (int64_t)1,           // Ya
                     (const double*)W, (int64_t)kp, (int64_t)1,
                     wb.Y + o_lo * kp + o_mid, (int64_t)kp, a, b, a, -1.0);
}

// Enqueue the warm start for the k slots of the solver on its stream: on return (asynchronously) s->plist holds the
// passive set in order, *p_dev its size, s->hinv / s->hinv_lo the inverse of its Gram block.  `buf`: bcx_warm_bytes(k)
// bytes; `gram_work`: the Gram kernel's scratch for (kp, kp).  0 ok, 1 not applicable, < 0 error.
int bcx_warm_start(bcx_solver* s, int k, void* buf, double* gram_work, const int32_t** p_dev) {
  if (k < 2 * WM_NB || k > 2048) return 1;
  const int kp = warm_kp(k);
  WarmBufs wb;
  char* base = (char*)buf;
  wb.A = (double*)base; base += (size_t)kp * kp * 8;
  wb.Y = (double*)base; base += (size_t)kp * kp * 8;
  wb.T = (double*)base; base += (size_t)(kp / WM_NB) * WM_NB * WM_NB * 8;
  wb.diag0 = (double*)base; base += (size_t)kp * 8;
  wb.perm = (int32_t*)base; base += (size_t)kp * 4;
  wb.rej = (int32_t*)base; base += (size_t)kp * 4;
  wb.keptq = (int32_t*)base; base += (size_t)kp * 4;
  wb.p = (int32_t*)base;                          // [0] size of the passive set, [1] columns accepted so far (Cholesky)
  if (hipMemsetAsync(wb.p, 0, 2 * sizeof(int32_t), s->stream) != hipSuccess) return BCX_ERR_HIP;
  hipStream_t st = s->stream;
  hipLaunchKernelGGL(warm_order_kernel, dim3((kp + 255) / 256), dim3(256), (size_t)k * 8, st, (const double*)s->act_w, (const double*)s->act_norm, k, kp, wb);
  hipLaunchKernelGGL(warm_gather_kernel, dim3((kp + 255) / 256, kp), dim3(256), 0, st, (const double*)s->gram, (int64_t)s->gram_cap, k, kp, wb);
  const int np = kp / WM_NB;
  for (int j = 0; j < np; ++j) {
    const int jj = j * WM_NB, below = np - 1 - j;
    hipLaunchKernelGGL(warm_diag_kernel, dim3(1), dim3(64), 0, st, kp, jj, (int)s->cfg.d, wb);
    if (below > 0) {
      hipLaunchKernelGGL(warm_panel_kernel, dim3(below), dim3(256), 0, st, kp, jj, wb);
      hipLaunchKernelGGL(warm_update_kernel, dim3(below * (below + 1) / 2), dim3(256), 0, st, kp, jj, wb);
    }
  }
  hipLaunchKernelGGL(warm_clean_kernel, dim3(kp), dim3(256), 0, st, kp, wb);
  hipLaunchKernelGGL(warm_zero_lower_kernel, dim3(kp), dim3(256), 0, st, kp, wb);
  warm_invert(s, st, kp, wb, 0, np);                         // Y = L^-T
  BCX_HIP(hipGetLastError());
  const int rc = bcx_gram_rows(st, wb.Y, kp, kp, (int64_t)kp, wb.A, (int64_t)kp, gram_work);      // H = Y Y^T (fp64 MFMA)
  if (rc != BCX_OK) { s->err = "optimize: Gram kernel (inverse) launch failed"; return rc; }
  hipLaunchKernelGGL(warm_scan_kernel, dim3(1), dim3(1024), 0, st, k, kp, wb, s->plist);
  hipLaunchKernelGGL(warm_compact_kernel, dim3((kp + 255) / 256, kp), dim3(256), 0, st, kp, wb, s->hinv, s->hinv_lo, (int64_t)s->gram_cap);
  BCX_HIP(hipGetLastError());
  *p_dev = wb.p;
  return BCX_OK;
}
