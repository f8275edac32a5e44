// svi.hip -- SparseVI's weight optimisation without the host in the loop (sparsevi.py:69-76, util/opt.py:4-28).
//
// One ADAM step of the reference is: projector.update(w, pts) (draw S parameter vectors from the posterior of the weighted
// coreset, sparsevi.py:25), project the data and the coreset points (sparsevi.py:35-41), gradient -corevecs.resid / S
// (sparsevi.py:72-74), ADAM update + clamp (util/opt.py:19-25) -- 100 times per greedy step, each depending on the last.
// The N-sized part is a projection kernel (csrc/proj.hip) or the closed form from the data's moments (csrc/moments.hip);
// around it the step used to be a k x k eigenproblem on the host, an upload, ~25 vendor / framework launches and a
// read-back (280 us per step: the closed-form path was launch-bound).  Here the weights never leave the device:
//
//   lrs_draw_kernel   the draws of the Gaussian linear-regression model's weighted conjugate posterior
//                     (examples/linear_regression/main.py:124-147: Sigma_w^-1 = Sig0^-1 + X^T diag(w) X / sigsq), as a
//                     rank-k correction of the PRIOR's factor Sig0 = U0 U0^T.  With C = diag(s) X U0, s = sqrt(w / sigsq):
//                         Sigma_w = U0 (I + C^T C)^-1 U0^T,   (I + C^T C)^-1 = F F^T,   F = I - C^T T C,
//                         T = L^-T (I + L)^-1,   L L^T = I + C C^T   (k x k Cholesky; check: T + T^T - T (L L^T - I) T^T = (L L^T)^-1)
//                     so  theta = mu_w + R Uw^T,  Uw^T = U0^T - (X U0)^T [diag(s) T^T diag(s)] (X Sig0),
//                         mu_w = mu0 + (X Sig0)^T (c - s * a'),  c = w y / sigsq,  a' = (L L^T)^-1 [s * (X mu0 + K0 c)],  K0 = (X U0)(X U0)^T
//                     for standard-normal R (S x D).  Everything that depends on the points alone (X U0, X Sig0, K0, X mu0) is
//                     formed once per greedy step by the caller; per ADAM step the kernel needs the k weights.  One workgroup
//                     per 64 draws x 16 columns of theta (one 16 x 16 tile of R Uw^T per wave, v_mfma_f64_16x16x4_f64; every
//                     workgroup repeats the k x k factorisation -- k <= 64, one wave, cheaper than a launch), Uw^T formed in LDS
//                     160 rows at a time, itself on the matrix cores.  The column means of R ride along as one more row: their
//                     "draw" is the mean of the draws (thetabar: what the closed-form column sums are expanded around,
//                     csrc/moments.hip).
//   svi_adam_kernel   resid = scaling colsum - w corevecs, g = -corevecs resid / S, the ADAM moments, the step and the clamp at
//                     zero, on the device-resident weights; the step-size schedule and the bias corrections arrive as data
//                     (the caller evaluates step_sched(i), 1 - b1^(i+1), 1 - b2^(i+1) on the host once per greedy step).
// The host enqueues opt_itrs x (draw, column sums, coreset projection, ADAM) and reads the weights back once.
#include <atomic>
#include <string>
#include "bcx_internal.h"
#include "dev_util.h"

typedef double sv4d __attribute__((ext_vector_type(4)));

#define LRS_KMAX 64          // coreset points
#define LRS_CH 160           // rows of Uw^T per LDS chunk (a multiple of 16)
#define LRS_CHD (LRS_CH / 8) // double-steps per chunk: 8 values of the inner dimension, two MFMAs
#define LRS_LDB 24           // doubles per staged row of Uw^T (16 used): lane groups lk, lk + 1 of a ds_read_b64 on disjoint bank halves
#define LRS_SMAX (1 << 22)   // rows per call

struct LrsArgs {
  const double* w;      // k weights (device-resident, updated by svi_adam_kernel)
  const double* K0;     // k x k
  const double* xmu0;   // k: X mu0
  const double* y;      // k
  const double* XU0;    // k x ld: X U0
  const double* XS0;    // k x ld: X Sig0
  const double* U0T;    // D x ld: U0^T (row i, column n = U0[n][i])
  const double* mu0;    // D
  const double* R;      // S x ld standard normal draws
  const double* Rbar;   // ld: their column means (one more row of R: its "draw" is the mean of the draws)
  double* theta;        // S x ld
  double* tbar;         // D: mean over the draws
  double sigsq;
  int k, D, S, ld;
  int dbg;              // dev (BCX_SVI_DBG=9): time stamps of workgroups (0..7, 2) into the next step's Rbar (tools/svi_step_bench.py)
};

// Workgroup (x, y): columns 16 x .. 16 x + 15 of theta, rows 64 y .. 64 y + 63 of [R; Rbar] (one 16 x 16 tile per wave).
// Everything is latency here (46 MFLOP per call; a dependent global load is ~1 us on this chip), so ALL requests that do not
// depend on the factorisation go out first -- this wave's rows of R for two chunks of the inner dimension (registers), its
// blocks of U0^T and (X U0)^T for the first chunk, the k x k inputs -- then: the k x k system (wave 0, LDS only; the others
// wait) -> per chunk of the inner dimension: Uw^T = U0^T - (X U0)^T B2 on the matrix cores into LDS (the next chunk's blocks
// requested meanwhile), the tile's MFMAs from it -> store.  Every workgroup repeats the k x k factorisation (cheaper than a
// launch and a round trip).
#define LRS_BPW ((LRS_CH / 16 + 3) / 4)            // 16-row blocks of a chunk per wave
__global__ __launch_bounds__(256) void lrs_draw_kernel(LrsArgs a) {
  // one buffer, two lives: I + C C^T and its Cholesky factor (lower) -> the chunks of Uw^T
  __shared__ double sbuf[LRS_KMAX * (LRS_KMAX + 1)];
  double (*sL)[LRS_KMAX + 1] = (double (*)[LRS_KMAX + 1])sbuf;
  double* sUw = sbuf;                               // LRS_CH x LRS_LDB
  __shared__ double sB2[LRS_KMAX][16];              // diag(s) T^T diag(s) X Sig0 [:, cols]
  __shared__ double sXS[LRS_KMAX][16];              // X Sig0 [:, cols]
  __shared__ double ss[LRS_KMAX], sc[LRS_KMAX], sa[LRS_KMAX], smu[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int k = a.k, D = a.D, S = a.S, ld = a.ld;
  const int k4 = (k + 3) & ~3;
  const int n0 = blockIdx.x * 16;
  const int col = n0 + li;
  const int row0 = (blockIdx.y * 4 + wave) * 16;    // rows of [R; Rbar] of this wave's tile

  // this lane's row of the A operand: lane group lk feeds values kq = 8 t + 2 lk (+1) to MFMAs 2 t (2 t + 1): one 16-byte load
  // per two steps
  const int arow = row0 + li;
  const bool arow_ok = arow <= S;
  const double* rp = arow < S ? a.R + (size_t)arow * ld : (arow == S ? a.Rbar : a.R);
  double xa[2][LRS_CHD], xb[2][LRS_CHD];            // two chunks in registers (D <= 320: the whole inner dimension)
  auto load_A = [&](int kb, double (&va)[LRS_CHD], double (&vb)[LRS_CHD]) {
#pragma unroll
    for (int t = 0; t < LRS_CHD; ++t) {
      const int kq = kb + 8 * t + 2 * lk;
      const int kc = kq + 1 < ld ? kq : 0;           // (ld and kq are even: kq < ld implies kq + 1 < ld)
      const double2 v = *(const double2*)(rp + kc);
      va[t] = (arow_ok && kq < D) ? v.x : 0.0;
      vb[t] = (arow_ok && kq + 1 < D) ? v.y : 0.0;
    }
  };
  // this wave's blocks of a chunk: 16 rows of U0^T in the C layout (row lk + 4 r, column li) and, for the first four points,
  // -(X U0)^T in the A layout (row li, point lk)
  double pu[LRS_BPW][4], px[LRS_BPW];
  auto load_U = [&](int kb) {
#pragma unroll
    for (int b = 0; b < LRS_BPW; ++b) {
      const int rb = kb + 16 * (wave + 4 * b);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = rb + lk + 4 * r;
        pu[b][r] = (row < D && col < D) ? a.U0T[(size_t)row * ld + col] : 0.0;
      }
      px[b] = (lk < k && rb + li < D) ? -a.XU0[(size_t)lk * ld + rb + li] : 0.0;
    }
  };
  long long stamp[8];
#define LRS_STAMP(i) do { if (a.dbg == 9) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[i] = wall_clock64(); } } while (0)
  LRS_STAMP(0);
  const long long c0 = clock64();
  load_A(0, xa[0], xb[0]);
  if (LRS_CH < D) load_A(LRS_CH, xa[1], xb[1]);
  load_U(0);
  LRS_STAMP(1);

  // ---- the k x k system: every input straight from memory (no value waits for another) ----
  for (int e = tid; e < k * k; e += 256) {
    const int i = e / k, j = e - i * k;
    const double si = sqrt(fmax(a.w[i], 0.0) / a.sigsq), sj = sqrt(fmax(a.w[j], 0.0) / a.sigsq);
    sL[i][j] = (i == j ? 1.0 : 0.0) + si * sj * a.K0[e];
  }
  if (tid < k) {
    const double wi = fmax(a.w[tid], 0.0), si = sqrt(wi / a.sigsq);
    ss[tid] = si;
    sc[tid] = wi * a.y[tid] / a.sigsq;
    double t = a.xmu0[tid];
    for (int j0 = 0; j0 < k; j0 += 8) {             // (eight points per round of loads)
      double kk[8], ww[8], yy[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = j0 + q < k ? j0 + q : k - 1;
        kk[q] = a.K0[tid * k + j]; ww[q] = a.w[j]; yy[q] = a.y[j];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (j0 + q < k) t += kk[q] * (fmax(ww[q], 0.0) * yy[q] / a.sigsq);
    }
    sa[tid] = si * t;
  }
  for (int e = tid; e < k4 * 16; e += 256) {
    const int j = e >> 4, c = e & 15;
    const bool ok = j < k && n0 + c < D;
    const double x = ok ? a.XS0[(size_t)j * ld + n0 + c] : 0.0;
    sXS[j][c] = x;
    sB2[j][c] = ok ? sqrt(fmax(a.w[j], 0.0) / a.sigsq) * x : 0.0;
  }
  const double mu0c = (tid < 16 && n0 + tid < D) ? a.mu0[n0 + tid] : 0.0;
  __syncthreads();
  LRS_STAMP(2);
  if (wave == 0) {
    // right-looking Cholesky, lane = row (a wave runs in lock step and its LDS accesses complete in order: the exchanges
    // between lanes below need no barrier, only that the compiler keeps the order -- wave_barrier)
    for (int c = 0; c < k; ++c) {
      const double l = sqrt(sL[c][c]);              // (>= 1: the matrix is I + a Gram matrix)
      __builtin_amdgcn_wave_barrier();
      if (lane >= c && lane < k) sL[lane][c] = lane == c ? l : sL[lane][c] / l;
      __builtin_amdgcn_wave_barrier();
      if (lane > c && lane < k) {
        const double lrc = sL[lane][c];
        for (int c2 = c + 1; c2 <= lane; ++c2) sL[lane][c2] -= lrc * sL[c2][c];
      }
      __builtin_amdgcn_wave_barrier();
    }
    // lanes 0..15: B2[:, c] = diag(s) (I + L)^-T L^-1 (s * X Sig0[:, c]);  lane 16: the mean's coefficients c - s * (L L^T)^-1 a
    if (lane < 17) {
      const bool mean = lane == 16;
      double* cl = mean ? sa : &sB2[0][lane];
      const int cs = mean ? 1 : 16;
      for (int i = 0; i < k; ++i) {                 // L u = rhs
        double t = cl[i * cs];
        for (int u = 0; u < i; ++u) t -= sL[i][u] * cl[u * cs];
        cl[i * cs] = t / sL[i][i];
      }
      const double one = mean ? 0.0 : 1.0;          // ((I + L)^T v = u for the factor, L^T v = u for the mean)
      for (int i = k - 1; i >= 0; --i) {
        double t = cl[i * cs];
        for (int u = i + 1; u < k; ++u) t -= sL[u][i] * cl[u * cs];
        cl[i * cs] = t / (one + sL[i][i]);
      }
      for (int i = 0; i < k; ++i) cl[i * cs] = mean ? sc[i] - ss[i] * cl[i * cs] : ss[i] * cl[i * cs];
    }
  }
  __syncthreads();
  LRS_STAMP(3);
  if (tid < 16) {
    double m = mu0c;
    for (int j = 0; j < k; ++j) m += sa[j] * sXS[j][tid];
    smu[tid] = m;
  }

  // ---- [R; Rbar] Uw^T, the inner dimension in chunks of LRS_CH rows of Uw^T ----
  sv4d acc = (sv4d){0.0, 0.0, 0.0, 0.0};
  // Uw^T[rows of the chunk][16 columns] = U0^T - (X U0)^T B2, 16 rows per wave and block: v_mfma_f64_16x16x4_f64 with
  // A = -(X U0)^T (lane: row li of the block, point m + lk), B = B2 (point m + lk, column li), C = U0^T; then the tile's MFMAs
  auto chunk = [&](int kb, double (&va)[LRS_CHD], double (&vb)[LRS_CHD]) {
    __syncthreads();                                // (the previous chunk has been read; first chunk: the factor is no longer needed)
    const int crows = min(LRS_CH, (D - kb + 15) & ~15);      // rows >= D are formed as zeros (LDS may hold anything)
#pragma unroll
    for (int b = 0; b < LRS_BPW; ++b) {
      const int blk = wave + 4 * b;
      if (16 * blk < crows) {                       // (wave-uniform)
        const int rb = kb + 16 * blk;
        sv4d u = (sv4d){pu[b][0], pu[b][1], pu[b][2], pu[b][3]};
        if (k4) u = __builtin_amdgcn_mfma_f64_16x16x4f64(px[b], sB2[lk][li], u, 0, 0, 0);
        const bool rok = rb + li < D;
        for (int m = 4; m < k4; m += 4) {
          const int j = m + lk;
          const double x = (j < k && rok) ? -a.XU0[(size_t)j * ld + rb + li] : 0.0;
          u = __builtin_amdgcn_mfma_f64_16x16x4f64(x, sB2[j][li], u, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sUw[(16 * blk + lk + 4 * r) * LRS_LDB + li] = u[r];
      }
    }
    if (kb + LRS_CH < D) load_U(kb + LRS_CH);       // (in flight during this chunk's MFMAs)
    __syncthreads();
    const int nds = (min(LRS_CH, D - kb) + 7) / 8;  // double-steps with anything in them
#pragma unroll
    for (int t = 0; t < LRS_CHD; ++t) {
      if (t < nds) {
        const int r = 8 * t + 2 * lk;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(va[t], sUw[r * LRS_LDB + li], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vb[t], sUw[(r + 1) * LRS_LDB + li], acc, 0, 0, 0);
      }
    }
    if (kb + 2 * LRS_CH < D) load_A(kb + 2 * LRS_CH, va, vb);      // (D > 320: this register set's next chunk)
  };
  for (int kb = 0; kb < D; kb += 2 * LRS_CH) {
    chunk(kb, xa[0], xb[0]);
    LRS_STAMP(4);
    if (kb + LRS_CH < D) chunk(kb + LRS_CH, xa[1], xb[1]);
  }
  LRS_STAMP(5);
  // ---- theta = mu + R Uw^T (f64 C/D layout: column = lane & 15, row = (lane >> 4) + 4 reg); row S is the mean ----
  const double mu = smu[li];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + lk + 4 * r;
    const double v = col < D ? mu + acc[r] : 0.0;
    if (row < S && col < ld) a.theta[(size_t)row * ld + col] = v;
    else if (row == S && col < D) a.tbar[col] = v;
  }
  if (a.dbg == 9) {
    LRS_STAMP(6);
    const long long c1 = clock64();
    if (tid == 0 && blockIdx.y == 2 && blockIdx.x < 8) {
      double* o = (double*)a.Rbar + ld + 8 * blockIdx.x;      // (the next step's means: a debugging run draws once)
      for (int i = 0; i < 7; ++i) o[i] = (double)(stamp[i] - (i ? stamp[0] : 0)) * 0.01;
      o[7] = (double)(c1 - c0) / ((double)(stamp[6] - stamp[0]) * 0.01);      // shader-clock ticks per microsecond
    }
  }
}

// The same draws for a LOOP of calls at the same points (SparseVI: opt_itrs ADAM steps): G = R U0^T does not depend on the
// weights, so the caller forms it for all steps at once (lrs_draw_kernel with k = 0, one launch per greedy step) and a step is
//     theta = mu_w + G - (G X^T) B2,     B2 = diag(s) T^T diag(s) (X Sig0)   (k x D),
// a rank-k correction of rows that are read once, contiguously -- the per-step kernel above spends 13 us waiting for its
// 16 x 302 tiles of a fresh R (16 rows 2.4 KB apart per load instruction) and 5 us forming Uw^T.  One wave per row of
// [G; Gbar] (Gbar: the column means of G -- its row gives the mean of the draws), four rows per workgroup; every workgroup
// repeats the k x k factorisation and the 2 D triangular solves for B2 (a thread per column), X and B2 live in LDS.
#define LRA_KMAX 32
#define LRA_NI 8             // 16-byte pieces of a row per lane: ld <= 1024
struct LraArgs {
  const double* w; const double* K0; const double* xmu0; const double* y;
  const double* X;      // k x ld: the points' features (padding 0)
  const double* XS0;    // k x ld: X Sig0
  const double* mu0;    // D
  const double* G;      // S x ld: R U0^T
  const double* Gbar;   // ld: column means of G
  const double* gscale; // ld, or null: G and Gbar are R and Rbar, the prior's factor is diag(gscale) (an isotropic / diagonal prior)
  double* theta; double* tbar;
  double sigsq;
  int k, D, S, ld;
};
__global__ __launch_bounds__(256) void lrs_apply_kernel(LraArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lra_dyn[];   // X (k x ld), B2 (k x ld)
  __shared__ double sL[LRA_KMAX][LRA_KMAX + 1];
  __shared__ double ss[LRA_KMAX], sc[LRA_KMAX], sa[LRA_KMAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int k = a.k, D = a.D, S = a.S, ld = a.ld;
  double* sX = lra_dyn;
  double* sB2 = lra_dyn + (size_t)k * ld;
  const int row = blockIdx.x * 4 + wave;            // of [G; Gbar]
  const bool row_ok = row <= S;
  const double* gp = row < S ? a.G + (size_t)row * ld : a.Gbar;
  // ---- every request that depends on nothing: this wave's row, the prior mean, the points, the k x k inputs ----
  double2 g[LRA_NI], m0[LRA_NI];
#pragma unroll
  for (int i = 0; i < LRA_NI; ++i) {
    const int c = 2 * lane + 128 * i;
    const bool ok = c < ld && row_ok;
    g[i] = ok ? *(const double2*)(gp + c) : make_double2(0.0, 0.0);
    if (a.gscale && ok) { const double2 sc = *(const double2*)(a.gscale + c); g[i].x *= sc.x; g[i].y *= sc.y; }
    m0[i].x = (ok && c < D) ? a.mu0[c] : 0.0;
    m0[i].y = (ok && c + 1 < D) ? a.mu0[c + 1] : 0.0;
  }
  for (int e = tid; e < k * ld; e += 256) {
    const int j = e / ld;
    sX[e] = a.X[e];
    sB2[e] = sqrt(fmax(a.w[j], 0.0) / a.sigsq) * a.XS0[e];
  }
  for (int e = tid; e < k * k; e += 256) {
    const int i = e / k, j = e - i * k;
    const double si = sqrt(fmax(a.w[i], 0.0) / a.sigsq), sj = sqrt(fmax(a.w[j], 0.0) / a.sigsq);
    sL[i][j] = (i == j ? 1.0 : 0.0) + si * sj * a.K0[e];
  }
  if (tid < k) {
    const double wi = fmax(a.w[tid], 0.0), si = sqrt(wi / a.sigsq);
    ss[tid] = si;
    sc[tid] = wi * a.y[tid] / a.sigsq;
    double t = a.xmu0[tid];
    for (int j0 = 0; j0 < k; j0 += 8) {
      double kk[8], ww[8], yy[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int j = j0 + q < k ? j0 + q : k - 1;
        kk[q] = a.K0[tid * k + j]; ww[q] = a.w[j]; yy[q] = a.y[j];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q)
        if (j0 + q < k) t += kk[q] * (fmax(ww[q], 0.0) * yy[q] / a.sigsq);
    }
    sa[tid] = si * t;
  }
  __syncthreads();
  if (wave == 0) {                                  // right-looking Cholesky, lane = row (see lrs_draw_kernel)
    for (int c = 0; c < k; ++c) {
      const double l = sqrt(sL[c][c]);
      __builtin_amdgcn_wave_barrier();
      if (lane >= c && lane < k) sL[lane][c] = lane == c ? l : sL[lane][c] / l;
      __builtin_amdgcn_wave_barrier();
      if (lane > c && lane < k) {
        const double lrc = sL[lane][c];
        for (int c2 = c + 1; c2 <= lane; ++c2) sL[lane][c2] -= lrc * sL[c2][c];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __syncthreads();
  // a thread per column n < D: B2[:, n] = diag(s) (I + L)^-T L^-1 (s * X Sig0[:, n]); one more "column" is the mean's coefficient
  // vector c - s * (L L^T)^-1 a
  for (int n = tid; n <= D; n += 256) {             // (n == D: the mean's vector)
    const bool mean = n == D;
    double* cl = mean ? sa : sB2 + n;
    const int cs = mean ? 1 : ld;
    for (int i = 0; i < k; ++i) {                   // L u = rhs
      double t = cl[(size_t)i * cs];
      for (int u = 0; u < i; ++u) t -= sL[i][u] * cl[(size_t)u * cs];
      cl[(size_t)i * cs] = t / sL[i][i];
    }
    const double one = mean ? 0.0 : 1.0;
    for (int i = k - 1; i >= 0; --i) {
      double t = cl[(size_t)i * cs];
      for (int u = i + 1; u < k; ++u) t -= sL[u][i] * cl[(size_t)u * cs];
      cl[(size_t)i * cs] = t / (one + sL[i][i]);
    }
    for (int i = 0; i < k; ++i) cl[(size_t)i * cs] = mean ? sc[i] - ss[i] * cl[i] : ss[i] * cl[(size_t)i * cs];
  }
  __syncthreads();
  if (!row_ok) return;
  // ---- this wave's row: theta = mu0 + (X Sig0)^T coef + g - sum_j (g . X_j) B2_j ----
  double2 acc[LRA_NI];
#pragma unroll
  for (int i = 0; i < LRA_NI; ++i) acc[i] = make_double2(m0[i].x + g[i].x, m0[i].y + g[i].y);
  for (int j0 = 0; j0 < k; j0 += 4) {
    double2 xs[4][LRA_NI];                          // X Sig0, four points per round of loads (cached: every workgroup reads them)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < LRA_NI; ++i) {
        const int c = 2 * lane + 128 * i;
        xs[q][i] = (j0 + q < k && c < ld) ? *(const double2*)(a.XS0 + (size_t)(j0 + q) * ld + c) : make_double2(0.0, 0.0);
      }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (j0 + q < k) {                             // (wave-uniform)
        const int j = j0 + q;
        double pj = 0.0;
#pragma unroll
        for (int i = 0; i < LRA_NI; ++i) {
          const int c = 2 * lane + 128 * i;
          if (c < ld) {
            const double2 x = *(const double2*)(sX + (size_t)j * ld + c);
            pj = fma(g[i].x, x.x, pj);
            pj = fma(g[i].y, x.y, pj);
          }
        }
        pj = wave_allsum(pj);
        const double cj = sa[j];
#pragma unroll
        for (int i = 0; i < LRA_NI; ++i) {
          const int c = 2 * lane + 128 * i;
          if (c < ld) {
            const double2 b = *(const double2*)(sB2 + (size_t)j * ld + c);
            acc[i].x += cj * xs[q][i].x - pj * b.x;
            acc[i].y += cj * xs[q][i].y - pj * b.y;
          }
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < LRA_NI; ++i) {
    const int c = 2 * lane + 128 * i;
    if (c < ld) {
      if (c + 1 >= D) acc[i].y = 0.0;               // (the padding column)
      if (row < S) *(double2*)(a.theta + (size_t)row * ld + c) = acc[i];
      else { a.tbar[c] = acc[i].x; if (c + 1 < D) a.tbar[c + 1] = acc[i].y; }
    }
  }
}

// One projected-ADAM step on the device-resident weights (util/opt.py:19-25 with nn_idcs = None: every weight is clamped).
// sched: rows of 3 doubles per step: step_sched(i), 1 - b1^(i+1), 1 - b2^(i+1).
__global__ __launch_bounds__(256) void svi_adam_kernel(const double* __restrict__ colsum, double scaling, const double* __restrict__ core,
                                                       int64_t ldc, int k, int S, double* __restrict__ w, double* __restrict__ mom1,
                                                       double* __restrict__ mom2, const double* __restrict__ sched, int step,
                                                       double b1, double b2, double eps, double* __restrict__ trace, int raw_core) {
  __shared__ double sw[LRS_KMAX], sg[LRS_KMAX], scm[LRS_KMAX];
  extern __shared__ double resid[];                 // S
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < k) { sw[tid] = w[tid]; scm[tid] = 0.0; }
  __syncthreads();
  if (raw_core) {
    // the projected coreset points arrive as raw log-likelihoods: their row means (projector.py:21) are taken here
    for (int j = wave; j < k; j += 4) {
      double t = 0.0;
      for (int s = lane; s < S; s += 64) t += core[(size_t)j * ldc + s];
      t = wave_allsum(t);
      if (lane == 0) scm[j] = t / (double)S;
    }
    __syncthreads();
  }
  for (int s = tid; s < S; s += 256) {
    double t = 0.0;                                 // w.dot(corevecs) in NumPy's order of the k terms
    for (int j = 0; j < k; ++j) t += sw[j] * (core[(size_t)j * ldc + s] - scm[j]);
    resid[s] = scaling * colsum[s] - t;             // sparsevi.py:72
  }
  __syncthreads();
  for (int j = wave; j < k; j += 4) {
    double t = 0.0;
    for (int s = lane; s < S; s += 64) t += (core[(size_t)j * ldc + s] - scm[j]) * resid[s];
    t = wave_allsum(t);
    if (lane == 0) sg[j] = -t / (double)S;          // sparsevi.py:74
  }
  __syncthreads();
  if (tid < k) {
    const double g = sg[tid];
    const double m1 = b1 * mom1[tid] + (1.0 - b1) * g;
    const double m2 = b2 * mom2[tid] + (1.0 - b2) * g * g;
    mom1[tid] = m1;
    mom2[tid] = m2;
    const double* sc = sched + 3 * (size_t)step;
    const double stp = sc[0] * m1 / sc[1] / (eps + sqrt(m2 / sc[2]));
    const double x = fmax(sw[tid] - stp, 0.0);
    w[tid] = x;
    if (trace) trace[(size_t)step * k + tid] = x;
  }
}


// ---- more than 16 weights: the same step as two launches of one workgroup per slab of SVB_ROWS weights --------------------------
// (the single-workgroup kernel above walks the k x S projected coreset points three times with one wave per row: 50 us at
// k = 300.)  svi_adam_a_kernel: row means of its slab's raw rows and the slab's share of w.dot(corevecs); svi_adam_b_kernel:
// resid from the shares (added in slab order: fixed association), the slab's gradient entries, ADAM moments, step, clamp.
// Slabs of 8 weights; every loop over rows or shares keeps its loads independent and in flight together.
#define SVB_ROWS 8
#define SVB_KMAX 4096
#define SVB_SINGLE_MAX 16    // weights up to which the single-workgroup kernel is the faster form (8 us at 16, 20 at 32, 36 at 64; the two launches: 10-13)
struct SvbArgs {
  const double* colsum; const double* core; const double* sched;
  double* w; double* mom1; double* mom2; double* trace;
  double* cm;          // k: row means of the raw projected points (zeros when they arrive centred)
  double* part;        // nslab x S: sum over the slab's rows of w_j (core[j][s] - cm_j)
  double scaling, b1, b2, eps;
  int64_t ldc;
  int k, S, step, raw_core;
};
__global__ __launch_bounds__(256) void svi_adam_a_kernel(SvbArgs a) {
  __shared__ double scm[SVB_ROWS], sw[SVB_ROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * SVB_ROWS, nj = min(SVB_ROWS, a.k - j0);
  // row means: the four rows of a wave are requested together
  {
    double t[SVB_ROWS / 4];
#pragma unroll
    for (int u = 0; u < SVB_ROWS / 4; ++u) {
      const int q = wave + 4 * u;
      t[u] = 0.0;
      if (a.raw_core && q < nj) {
        const double* row = a.core + (size_t)(j0 + q) * a.ldc;
        for (int s = lane; s < a.S; s += 64) t[u] += row[s];
      }
    }
#pragma unroll
    for (int u = 0; u < SVB_ROWS / 4; ++u) {
      const int q = wave + 4 * u;
      const double m = wave_allsum(t[u]) / (double)a.S;     // projector.py:21
      if (lane == 0 && q < nj) { scm[q] = m; sw[q] = a.w[j0 + q]; a.cm[j0 + q] = m; }
    }
  }
  __syncthreads();
  for (int s = tid; s < a.S; s += 256) {
    double v[SVB_ROWS];
#pragma unroll
    for (int q = 0; q < SVB_ROWS; ++q) v[q] = q < nj ? a.core[(size_t)(j0 + q) * a.ldc + s] : 0.0;
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < SVB_ROWS; ++q) if (q < nj) t += sw[q] * (v[q] - scm[q]);
    a.part[(size_t)blockIdx.x * a.S + s] = t;
  }
}
__global__ __launch_bounds__(256) void svi_adam_b_kernel(SvbArgs a) {
  extern __shared__ double resid[];                 // S
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j0 = blockIdx.x * SVB_ROWS, nj = min(SVB_ROWS, a.k - j0);
  const int nslab = (a.k + SVB_ROWS - 1) / SVB_ROWS;
  for (int s = tid; s < a.S; s += 256) {
    double t = 0.0;
    int b = 0;
    for (; b + 32 <= nslab; b += 32) {              // (thirty-two shares in flight -- 300 weights: ONE round trip; added in slab order)
      double v[32];
#pragma unroll
      for (int q = 0; q < 32; ++q) v[q] = a.part[(size_t)(b + q) * a.S + s];
#pragma unroll
      for (int q = 0; q < 32; ++q) t += v[q];
    }
    for (; b + 8 <= nslab; b += 8) {
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = a.part[(size_t)(b + q) * a.S + s];
#pragma unroll
      for (int q = 0; q < 8; ++q) t += v[q];
    }
    for (; b < nslab; ++b) t += a.part[(size_t)b * a.S + s];
    resid[s] = a.scaling * a.colsum[s] - t;         // sparsevi.py:72
  }
  __syncthreads();
  double g[SVB_ROWS / 4];
#pragma unroll
  for (int u = 0; u < SVB_ROWS / 4; ++u) {
    const int q = wave + 4 * u;
    g[u] = 0.0;
    if (q < nj) {
      const double* row = a.core + (size_t)(j0 + q) * a.ldc;
      const double cm = a.cm[j0 + q];
      for (int s = lane; s < a.S; s += 64) g[u] += (row[s] - cm) * resid[s];
    }
  }
#pragma unroll
  for (int u = 0; u < SVB_ROWS / 4; ++u) {
    const int q = wave + 4 * u, j = j0 + q;
    const double t = wave_allsum(g[u]);
    if (lane == 0 && q < nj) {
      const double gr = -t / (double)a.S;           // sparsevi.py:74
      const double m1 = a.b1 * a.mom1[j] + (1.0 - a.b1) * gr;
      const double m2 = a.b2 * a.mom2[j] + (1.0 - a.b2) * gr * gr;
      a.mom1[j] = m1;
      a.mom2[j] = m2;
      const double* sc = a.sched + 3 * (size_t)a.step;
      const double stp = sc[0] * m1 / sc[1] / (a.eps + sqrt(m2 / sc[2]));
      const double x = fmax(a.w[j] - stp, 0.0);
      a.w[j] = x;
      if (a.trace) a.trace[(size_t)a.step * a.k + j] = x;
    }
  }
}


// ---- the sampler's normal numbers and their column means, without the framework ------------------------------------------------
// svi_normal_kernel: standard normal doubles from the counter-based generator Philox-4x32-10 (Salmon et al., SC'11: the round
// and key constants below are the published ones) + Box-Muller: counter (index / 2 + offset, 0), key = seed; two 53-bit
// uniforms in (0, 1) per counter give two normals.  The same (seed, offset) gives the same numbers whatever the launch shape.
static __device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ __launch_bounds__(256) void svi_normal_kernel(double* __restrict__ out, int64_t count, unsigned long long seed,
                                                         unsigned long long offset) {
  const int64_t pair = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (2 * pair >= count) return;
  const unsigned long long ctr = (unsigned long long)pair + offset;
  unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0u, 0u};
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  // 53 bits of each pair of words, centred in its cell: never 0 or 1
  const double u1 = ((double)((((unsigned long long)c[0] << 32) | c[1]) >> 11) + 0.5) * 0x1.0p-53;
  const double u2 = ((double)((((unsigned long long)c[2] << 32) | c[3]) >> 11) + 0.5) * 0x1.0p-53;
  const double rad = sqrt(-2.0 * log(u1));
  double sn, cs;
  sincospi(2.0 * u2, &sn, &cs);
  out[2 * pair] = rad * cs;
  if (2 * pair + 1 < count) out[2 * pair + 1] = rad * sn;
}
// svi_colmean_kernel: out[b][c] = mean over the n rows of block b of rows (nblocks x n x ld): one workgroup per block and 64
// columns, four row groups added in group order.
__global__ __launch_bounds__(256) void svi_colmean_kernel(const double* __restrict__ rows, int n, int ld, int64_t block_stride,
                                                          double* __restrict__ out, int64_t out_stride) {
  __shared__ double sp[4][64];
  const int c = blockIdx.y * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const double* base = rows + (size_t)blockIdx.x * block_stride;
  double t = 0.0;
  if (c < ld) {
    int r = g;
    for (; r + 12 < n; r += 16) {                   // (four rows in flight)
      const double x0 = base[(size_t)r * ld + c], x1 = base[(size_t)(r + 4) * ld + c];
      const double x2 = base[(size_t)(r + 8) * ld + c], x3 = base[(size_t)(r + 12) * ld + c];
      t += x0; t += x1; t += x2; t += x3;
    }
    for (; r < n; r += 4) t += base[(size_t)r * ld + c];
  }
  sp[g][threadIdx.x & 63] = t;
  __syncthreads();
  if (g == 0 && c < ld) out[(size_t)blockIdx.x * out_stride + c] = (((sp[0][c & 63] + sp[1][c & 63]) + sp[2][c & 63]) + sp[3][c & 63]) / (double)n;
}

void bcx_project_set_error(const std::string& msg);   // proj.hip
#define SVI_HIP(call)                                                             \
  do {                                                                            \
    hipError_t _e = (call);                                                       \
    if (_e != hipSuccess) {                                                       \
      bcx_project_set_error(std::string(#call) + ": " + hipGetErrorString(_e));   \
      return BCX_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

extern "C" int bcx_linreg_posterior_draw(void* stream, int32_t k, int32_t D, int32_t ld, const void* w_dev, const void* K0_dev,
                                         const void* xmu0_dev, const void* y_dev, const void* XU0_dev, const void* XS0_dev,
                                         const void* U0T_dev, const void* mu0_dev, double sigsq, const void* R_dev,
                                         const void* Rbar_dev, int32_t S, void* theta_dev, void* tbar_dev) {
  if (k < 0 || k > LRS_KMAX || D < 1 || ld < D || (ld & 1) || S < 1 || S > LRS_SMAX || !(sigsq > 0.0) || !U0T_dev || !mu0_dev ||
      !R_dev || !Rbar_dev || !theta_dev || !tbar_dev || (k > 0 && (!w_dev || !K0_dev || !xmu0_dev || !y_dev || !XU0_dev || !XS0_dev)) ||
      ((uintptr_t)R_dev & 15) || ((uintptr_t)Rbar_dev & 15)) {
    bcx_project_set_error("bcx_linreg_posterior_draw: bad arguments (k <= 64 points, even leading dimension, "
                          "16-byte aligned normal draws)");
    return BCX_ERR_ARG;
  }
  LrsArgs a;
  a.w = (const double*)w_dev; a.K0 = (const double*)K0_dev; a.xmu0 = (const double*)xmu0_dev; a.y = (const double*)y_dev;
  a.XU0 = (const double*)XU0_dev; a.XS0 = (const double*)XS0_dev; a.U0T = (const double*)U0T_dev; a.mu0 = (const double*)mu0_dev;
  a.R = (const double*)R_dev; a.Rbar = (const double*)Rbar_dev; a.theta = (double*)theta_dev; a.tbar = (double*)tbar_dev;
  a.sigsq = sigsq; a.k = k; a.D = D; a.S = S; a.ld = ld;
  static const char* dbg = bcx_dev_env("BCX_SVI_DBG");
  a.dbg = dbg ? atoi(dbg) : 0;
  hipLaunchKernelGGL(lrs_draw_kernel, dim3((ld + 15) / 16, (S + 1 + 63) / 64), dim3(256), 0, (hipStream_t)stream, a);
  SVI_HIP(hipGetLastError());
  return BCX_OK;
}

// LDS bytes of lrs_apply_kernel's X and B2; the kernel serves k <= 32 points while they fit 128 KiB
extern "C" int bcx_linreg_posterior_apply_ok(int32_t k, int32_t ld) {
  return k >= 1 && k <= LRA_KMAX && ld >= 2 && ld <= 128 * LRA_NI && (int64_t)2 * k * ld * 8 <= 128 * 1024;
}
extern "C" int bcx_linreg_posterior_apply(void* stream, int32_t k, int32_t D, int32_t ld, const void* w_dev, const void* K0_dev,
                                          const void* xmu0_dev, const void* y_dev, const void* X_dev, const void* XS0_dev,
                                          const void* mu0_dev, double sigsq, const void* G_dev, const void* Gbar_dev, int32_t S,
                                          void* theta_dev, void* tbar_dev, const void* gscale_dev) {
  if (!bcx_linreg_posterior_apply_ok(k, ld) || D < 1 || ld < D || (ld & 1) || S < 1 || !(sigsq > 0.0) || !w_dev || !K0_dev || !xmu0_dev ||
      !y_dev || !X_dev || !XS0_dev || !mu0_dev || !G_dev || !Gbar_dev || !theta_dev || !tbar_dev ||
      (((uintptr_t)G_dev | (uintptr_t)Gbar_dev | (uintptr_t)XS0_dev | (uintptr_t)theta_dev | (uintptr_t)gscale_dev) & 15)) {
    bcx_project_set_error("bcx_linreg_posterior_apply: bad arguments (see bcx_linreg_posterior_apply_ok; 16-byte aligned rows)");
    return BCX_ERR_ARG;
  }
  LraArgs a;
  a.w = (const double*)w_dev; a.K0 = (const double*)K0_dev; a.xmu0 = (const double*)xmu0_dev; a.y = (const double*)y_dev;
  a.X = (const double*)X_dev; a.XS0 = (const double*)XS0_dev; a.mu0 = (const double*)mu0_dev; a.G = (const double*)G_dev;
  a.Gbar = (const double*)Gbar_dev; a.gscale = (const double*)gscale_dev; a.theta = (double*)theta_dev; a.tbar = (double*)tbar_dev;
  a.sigsq = sigsq; a.k = k; a.D = D; a.S = S; a.ld = ld;
  const size_t lds = (size_t)2 * k * ld * sizeof(double);
  if (lds > 32 * 1024) {
    // per DEVICE (a process may drive several): the largest request so far on the current one
    static std::atomic<size_t> lds_max[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (lds > lds_max[dev].load(std::memory_order_acquire)) {
      SVI_HIP(hipFuncSetAttribute((const void*)lrs_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      lds_max[dev].store(lds, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(lrs_apply_kernel, dim3((S + 1 + 3) / 4), dim3(256), lds, (hipStream_t)stream, a);
  SVI_HIP(hipGetLastError());
  return BCX_OK;
}

extern "C" int bcx_sparsevi_adam_step(void* stream, int32_t k, int32_t S, const void* colsum_dev, double scaling, const void* core_dev,
                                      int64_t ldc, void* w_dev, void* mom1_dev, void* mom2_dev, const void* sched_dev, int32_t step,
                                      double b1, double b2, double eps, void* trace_dev, int32_t core_is_raw) {
  if (k < 1 || k > LRS_KMAX || S < 1 || S > 8192 || ldc < S || step < 0 || !colsum_dev || !core_dev || !w_dev || !mom1_dev ||
      !mom2_dev || !sched_dev) {
    bcx_project_set_error("bcx_sparsevi_adam_step: bad arguments (1 <= k <= 64 weights, S <= 8192)");
    return BCX_ERR_ARG;
  }
  hipLaunchKernelGGL(svi_adam_kernel, dim3(1), dim3(256), (size_t)S * sizeof(double), (hipStream_t)stream, (const double*)colsum_dev,
                     scaling, (const double*)core_dev, ldc, (int)k, (int)S, (double*)w_dev, (double*)mom1_dev, (double*)mom2_dev,
                     (const double*)sched_dev, (int)step, b1, b2, eps, (double*)trace_dev, (int)core_is_raw);
  SVI_HIP(hipGetLastError());
  return BCX_OK;
}

// The same step for any number of weights up to 4096: up to 16 the single-workgroup kernel, beyond it the two-launch form,
// which needs bcx_sparsevi_adam_scratch_bytes(k, S) bytes of scratch (row means + the slabs' shares of w.dot(corevecs)).
extern "C" int64_t bcx_sparsevi_adam_scratch_bytes(int32_t k, int32_t S) {
  if (k < 1 || k > SVB_KMAX || S < 1 || S > 8192) return -1;
  if (k <= SVB_SINGLE_MAX) return 0;
  return ((int64_t)k + (int64_t)((k + SVB_ROWS - 1) / SVB_ROWS) * S) * (int64_t)sizeof(double);
}
extern "C" int bcx_sparsevi_adam_step_ws(void* stream, int32_t k, int32_t S, const void* colsum_dev, double scaling, const void* core_dev,
                                         int64_t ldc, void* w_dev, void* mom1_dev, void* mom2_dev, const void* sched_dev, int32_t step,
                                         double b1, double b2, double eps, void* trace_dev, int32_t core_is_raw, void* work_dev,
                                         int64_t work_bytes) {
  if (k >= 1 && k <= SVB_SINGLE_MAX)
    return bcx_sparsevi_adam_step(stream, k, S, colsum_dev, scaling, core_dev, ldc, w_dev, mom1_dev, mom2_dev, sched_dev, step, b1, b2, eps,
                                  trace_dev, core_is_raw);
  if (k < 1 || k > SVB_KMAX || S < 1 || S > 8192 || ldc < S || step < 0 || !colsum_dev || !core_dev || !w_dev || !mom1_dev || !mom2_dev ||
      !sched_dev || !work_dev || work_bytes < bcx_sparsevi_adam_scratch_bytes(k, S)) {
    bcx_project_set_error("bcx_sparsevi_adam_step_ws: bad arguments (1 <= k <= 4096 weights, S <= 8192, scratch of "
                          "bcx_sparsevi_adam_scratch_bytes(k, S) bytes)");
    return BCX_ERR_ARG;
  }
  SvbArgs a;
  a.colsum = (const double*)colsum_dev; a.core = (const double*)core_dev; a.sched = (const double*)sched_dev;
  a.w = (double*)w_dev; a.mom1 = (double*)mom1_dev; a.mom2 = (double*)mom2_dev; a.trace = (double*)trace_dev;
  a.cm = (double*)work_dev; a.part = a.cm + k;
  a.scaling = scaling; a.b1 = b1; a.b2 = b2; a.eps = eps; a.ldc = ldc; a.k = k; a.S = S; a.step = step; a.raw_core = core_is_raw;
  const int nslab = (k + SVB_ROWS - 1) / SVB_ROWS;
  hipLaunchKernelGGL(svi_adam_a_kernel, dim3(nslab), dim3(256), 0, (hipStream_t)stream, a);
  hipLaunchKernelGGL(svi_adam_b_kernel, dim3(nslab), dim3(256), (size_t)S * sizeof(double), (hipStream_t)stream, a);
  SVI_HIP(hipGetLastError());
  return BCX_OK;
}

// count standard normal doubles into out_dev (Philox-4x32-10 + Box-Muller; csrc/svi.hip svi_normal_kernel): the numbers of the
// pair counters offset .. offset + ceil(count / 2) - 1 under the key `seed` -- a caller that advances `offset` by
// ceil(count / 2) per call draws one reproducible stream.
extern "C" int bcx_standard_normal(void* stream, uint64_t seed, uint64_t offset, int64_t count, void* out_dev) {
  if (count < 0 || (count > 0 && !out_dev)) { bcx_project_set_error("bcx_standard_normal: bad arguments"); return BCX_ERR_ARG; }
  if (count == 0) return BCX_OK;
  const int64_t pairs = (count + 1) / 2;
  hipLaunchKernelGGL(svi_normal_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (double*)out_dev, count,
                     (unsigned long long)seed, (unsigned long long)offset);
  SVI_HIP(hipGetLastError());
  return BCX_OK;
}
// out_dev[b * out_stride + c] = mean over the n rows of block b of rows_dev (nblocks blocks of n x ld doubles, block_stride doubles
// apart): the column means of every ADAM step's normal numbers in one launch.
extern "C" int bcx_column_means(void* stream, const void* rows_dev, int32_t nblocks, int32_t n, int32_t ld, int64_t block_stride,
                                void* out_dev, int64_t out_stride) {
  if (nblocks < 1 || n < 1 || ld < 1 || block_stride < 0 || out_stride < ld || !rows_dev || !out_dev) {
    bcx_project_set_error("bcx_column_means: bad arguments");
    return BCX_ERR_ARG;
  }
  hipLaunchKernelGGL(svi_colmean_kernel, dim3(nblocks, (ld + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const double*)rows_dev, (int)n,
                     (int)ld, block_stride, (double*)out_dev, out_stride);
  SVI_HIP(hipGetLastError());
  return BCX_OK;
}
