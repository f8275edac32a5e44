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
//                     per 16 columns of theta (16 waves: one 16 x 16 tile of R Uw^T each per 256 draws, v_mfma_f64_16x16x4_f64;
//                     every workgroup repeats the k x k factorisation -- k <= 64, one wave, cheaper than a launch), Uw^T formed
//                     in LDS 128 rows at a time; the workgroup owns its columns for ALL draws and also leaves their mean
//                     (thetabar: what the closed-form column sums are expanded around, csrc/moments.hip).
//   svi_adam_kernel   resid = scaling colsum - w corevecs, g = -corevecs resid / S, the ADAM moments, the step and the clamp at
//                     zero, on the device-resident weights; the step-size schedule and the bias corrections arrive as data
//                     (the caller evaluates step_sched(i), 1 - b1^(i+1), 1 - b2^(i+1) on the host once per greedy step).
// The host enqueues opt_itrs x (draw, column sums, coreset projection, ADAM) and reads the weights back once.
#include <string>
#include "bcx_internal.h"
#include "dev_util.h"

typedef double sv4d __attribute__((ext_vector_type(4)));

#define LRS_KMAX 64          // coreset points
#define LRS_CH 128           // rows of Uw^T per LDS chunk
#define LRS_LDB 24           // doubles per staged row of Uw^T (16 used): lane groups lk, lk + 1 of a ds_read_b64 on disjoint bank halves
#define LRS_MAXT 4           // 16-draw tiles per wave: S <= 16 waves x 4 x 16 = 1024
#define LRS_SMAX (16 * LRS_MAXT * 16)

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
  double* theta;        // S x ld
  double* tbar;         // D: mean over the draws
  double sigsq;
  int k, D, S, ld;
};

__global__ __launch_bounds__(1024) void lrs_draw_kernel(LrsArgs a) {
  // one buffer, three lives: I + C C^T and its Cholesky factor (lower) -> the chunks of Uw^T -> the column-mean partials
  __shared__ double sbuf[LRS_KMAX * (LRS_KMAX + 1)];
  double (*sL)[LRS_KMAX + 1] = (double (*)[LRS_KMAX + 1])sbuf;
  double* sUw = sbuf;                               // LRS_CH x LRS_LDB
  __shared__ double sB2[LRS_KMAX][16];              // s * X Sig0 [:, cols]  ->  diag(s) T^T diag(s) X Sig0 [:, cols]
  __shared__ double ss[LRS_KMAX], sc[LRS_KMAX], sa[LRS_KMAX], smu[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int k = a.k, D = a.D, S = a.S, ld = a.ld;
  const int n0 = blockIdx.x * 16;

  // ---- the k x k system ----
  if (tid < k) {
    const double wj = fmax(a.w[tid], 0.0);
    ss[tid] = sqrt(wj / a.sigsq);
    sc[tid] = wj * a.y[tid] / a.sigsq;
  }
  __syncthreads();
  for (int e = tid; e < k * k; e += 1024) {
    const int i = e / k, j = e - i * k;
    sL[i][j] = (i == j ? 1.0 : 0.0) + ss[i] * ss[j] * a.K0[e];
  }
  if (tid < k) {
    double t = a.xmu0[tid];
    for (int j = 0; j < k; ++j) t += a.K0[tid * k + j] * sc[j];
    sa[tid] = ss[tid] * t;
  }
  for (int e = tid; e < k * 16; e += 1024) {
    const int j = e >> 4, c = e & 15;
    sB2[j][c] = n0 + c < D ? ss[j] * a.XS0[(size_t)j * ld + n0 + c] : 0.0;
  }
  __syncthreads();
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
      double* col = mean ? sa : &sB2[0][lane];
      const int cs = mean ? 1 : 16;
      for (int i = 0; i < k; ++i) {                 // L u = rhs
        double t = col[i * cs];
        for (int u = 0; u < i; ++u) t -= sL[i][u] * col[u * cs];
        col[i * cs] = t / sL[i][i];
      }
      const double one = mean ? 0.0 : 1.0;          // ((I + L)^T v = u for the factor, L^T v = u for the mean)
      for (int i = k - 1; i >= 0; --i) {
        double t = col[i * cs];
        for (int u = i + 1; u < k; ++u) t -= sL[u][i] * col[u * cs];
        col[i * cs] = t / (one + sL[i][i]);
      }
      for (int i = 0; i < k; ++i) col[i * cs] = mean ? sc[i] - ss[i] * col[i * cs] : ss[i] * col[i * cs];
    }
  }
  __syncthreads();
  if (tid < 16) {
    double m = 0.0;
    if (n0 + tid < D) {
      m = a.mu0[n0 + tid];
      for (int j = 0; j < k; ++j) m += sa[j] * a.XS0[(size_t)j * ld + n0 + tid];
    }
    smu[tid] = m;
  }

  // ---- R Uw^T, the inner dimension in chunks of LRS_CH rows of Uw^T ----
  const int ntiles = (S + 15) / 16;
  sv4d acc[LRS_MAXT];
#pragma unroll
  for (int u = 0; u < LRS_MAXT; ++u) acc[u] = (sv4d){0.0, 0.0, 0.0, 0.0};
  for (int kb = 0; kb < D; kb += LRS_CH) {
    __syncthreads();                                // (the previous chunk has been read; first trip: smu / sB2 are complete)
    for (int e = tid; e < LRS_CH * 16; e += 1024) {
      const int r = e >> 4, c = e & 15, kq = kb + r;
      double v = 0.0;
      if (kq < D && n0 + c < D) {
        v = a.U0T[(size_t)kq * ld + n0 + c];
        for (int j = 0; j < k; ++j) v -= a.XU0[(size_t)j * ld + kq] * sB2[j][c];
      }
      sUw[r * LRS_LDB + c] = v;
    }
    __syncthreads();
    const int nds = (min(LRS_CH, D - kb) + 7) / 8;  // double-steps (8 values of the inner dimension) with anything in them
#pragma unroll
    for (int u = 0; u < LRS_MAXT; ++u) {
      const int rt = wave + 16 * u;
      if (rt >= ntiles) break;                      // (wave-uniform)
      const int row = rt * 16 + li;
      const double* rp = a.R + (size_t)(row < S ? row : 0) * ld;
      for (int t0 = 0; t0 < nds; t0 += 4) {
        // lane group lk feeds values kq = 8 t + 2 lk (+1) to steps 2 t (2 t + 1): one 16-byte load per two steps
        double x0[4], x1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kq = kb + 8 * (t0 + q) + 2 * lk;
          const int kc = kq + 1 < ld ? kq : 0;       // (ld is even, kq is even: kq < ld implies kq + 1 < ld)
          const double2 v = *(const double2*)(rp + kc);
          const bool ok = row < S && t0 + q < nds;
          x0[q] = (ok && kq < D) ? v.x : 0.0;
          x1[q] = (ok && kq + 1 < D) ? v.y : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (t0 + q < nds) {
            const int r = 8 * (t0 + q) + 2 * lk;
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(x0[q], sUw[r * LRS_LDB + li], acc[u], 0, 0, 0);
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(x1[q], sUw[(r + 1) * LRS_LDB + li], acc[u], 0, 0, 0);
          }
        }
      }
    }
  }
  // ---- theta = mu + R Uw^T (f64 C/D layout: column = lane & 15, row = (lane >> 4) + 4 reg), column means ----
  const int col = n0 + li;
  const double mu = smu[li];
  double csum = 0.0;
#pragma unroll
  for (int u = 0; u < LRS_MAXT; ++u) {
    const int rt = wave + 16 * u;
    if (rt >= ntiles) break;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = rt * 16 + lk + 4 * r;
      if (row < S && col < ld) {
        const double v = col < D ? mu + acc[u][r] : 0.0;
        a.theta[(size_t)row * ld + col] = v;
        csum += v;
      }
    }
  }
  __syncthreads();                                  // (every wave has read its last chunk)
  double* red = sbuf;                               // 16 waves x 64 partials
  red[wave * 64 + lane] = csum;
  __syncthreads();
  if (tid < 16 && n0 + tid < D) {
    double t = 0.0;
    for (int wv = 0; wv < 16; ++wv)
      for (int g = 0; g < 4; ++g) t += red[wv * 64 + g * 16 + tid];
    a.tbar[n0 + tid] = t / (double)S;
  }
}

// One projected-ADAM step on the device-resident weights (util/opt.py:19-25 with nn_idcs = None: every weight is clamped).
// sched: rows of 3 doubles per step: step_sched(i), 1 - b1^(i+1), 1 - b2^(i+1).
__global__ __launch_bounds__(256) void svi_adam_kernel(const double* __restrict__ colsum, double scaling, const double* __restrict__ core,
                                                       int64_t ldc, int k, int S, double* __restrict__ w, double* __restrict__ mom1,
                                                       double* __restrict__ mom2, const double* __restrict__ sched, int step,
                                                       double b1, double b2, double eps, double* __restrict__ trace) {
  __shared__ double sw[LRS_KMAX], sg[LRS_KMAX];
  extern __shared__ double resid[];                 // S
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < k) sw[tid] = w[tid];
  __syncthreads();
  for (int s = tid; s < S; s += 256) {
    double t = 0.0;                                 // w.dot(corevecs) in NumPy's order of the k terms
    for (int j = 0; j < k; ++j) t += sw[j] * core[(size_t)j * ldc + s];
    resid[s] = scaling * colsum[s] - t;             // sparsevi.py:72
  }
  __syncthreads();
  for (int j = wave; j < k; j += 4) {
    double t = 0.0;
    for (int s = lane; s < S; s += 64) t += core[(size_t)j * ldc + s] * resid[s];
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
                                         const void* U0T_dev, const void* mu0_dev, double sigsq, const void* R_dev, int32_t S,
                                         void* theta_dev, void* tbar_dev) {
  if (k < 0 || k > LRS_KMAX || D < 1 || ld < D || (ld & 1) || S < 1 || S > LRS_SMAX || !(sigsq > 0.0) || !U0T_dev || !mu0_dev ||
      !R_dev || !theta_dev || !tbar_dev || (k > 0 && (!w_dev || !K0_dev || !xmu0_dev || !y_dev || !XU0_dev || !XS0_dev)) ||
      ((uintptr_t)R_dev & 15)) {
    bcx_project_set_error("bcx_linreg_posterior_draw: bad arguments (k <= 64 points, S <= 1024 draws, even leading dimension, "
                          "16-byte aligned normal draws)");
    return BCX_ERR_ARG;
  }
  LrsArgs a;
  a.w = (const double*)w_dev; a.K0 = (const double*)K0_dev; a.xmu0 = (const double*)xmu0_dev; a.y = (const double*)y_dev;
  a.XU0 = (const double*)XU0_dev; a.XS0 = (const double*)XS0_dev; a.U0T = (const double*)U0T_dev; a.mu0 = (const double*)mu0_dev;
  a.R = (const double*)R_dev; a.theta = (double*)theta_dev; a.tbar = (double*)tbar_dev;
  a.sigsq = sigsq; a.k = k; a.D = D; a.S = S; a.ld = ld;
  hipLaunchKernelGGL(lrs_draw_kernel, dim3((ld + 15) / 16), dim3(1024), 0, (hipStream_t)stream, a);
  SVI_HIP(hipGetLastError());
  return BCX_OK;
}

extern "C" int bcx_sparsevi_adam_step(void* stream, int32_t k, int32_t S, const void* colsum_dev, double scaling, const void* core_dev,
                                      int64_t ldc, void* w_dev, void* mom1_dev, void* mom2_dev, const void* sched_dev, int32_t step,
                                      double b1, double b2, double eps, void* trace_dev) {
  if (k < 1 || k > LRS_KMAX || S < 1 || S > 8192 || ldc < S || step < 0 || !colsum_dev || !core_dev || !w_dev || !mom1_dev ||
      !mom2_dev || !sched_dev) {
    bcx_project_set_error("bcx_sparsevi_adam_step: bad arguments (1 <= k <= 64 weights, S <= 8192)");
    return BCX_ERR_ARG;
  }
  hipLaunchKernelGGL(svi_adam_kernel, dim3(1), dim3(256), (size_t)S * sizeof(double), (hipStream_t)stream, (const double*)colsum_dev,
                     scaling, (const double*)core_dev, ldc, (int)k, (int)S, (double*)w_dev, (double*)mom1_dev, (double*)mom2_dev,
                     (const double*)sched_dev, (int)step, b1, b2, eps, (double*)trace_dev);
  SVI_HIP(hipGetLastError());
  return BCX_OK;
}
