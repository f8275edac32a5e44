// dev probe: issue rate of v_mfma_f64_16x16x4_f64 on this chip (register-only loops).
//   variant A: NACC independent accumulators per wave, W waves per SIMD  -> what the matrix pipe sustains
// Prints TFLOP/s per variant; run under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES` for the clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  v4d a[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) a[i] = (v4d){0, 0, 0, 0};
  double x = threadIdx.x * 1e-3, y = threadIdx.x * 2e-3 + 1.0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) a[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a[j], 0, 0, 0);
  }
  v4d s = a[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// the 4x4x4 (4 blocks) form: 512 flops per wave instruction, one double of C per lane
template <int NACC>
__global__ __launch_bounds__(256) void k4(double* out, int iters) {
  double a[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) a[i] = 0.0;
  double x = threadIdx.x * 1e-3, y = threadIdx.x * 2e-3 + 1.0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) a[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a[j], 0, 0, 0);
  }
  double s = a[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run4(double* d, int wg_per_cu) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int wg = 256 * wg_per_cu, iters = 640000 / NACC;
  hipLaunchKernelGGL(k4<NACC>, dim3(wg), dim3(256), 0, 0, d, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k4<NACC>, dim3(wg), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wg * 4 * iters * NACC * 512.0;
  printf("4x4x4 nacc=%2d wg/cu=%d (%d waves/SIMD): %.2f ms  %.1f TFLOP/s fp64 MFMA\n", NACC, wg_per_cu, wg_per_cu, ms, flops / ms / 1e9);
}

template <int NACC> void run(double* d, int wg_per_cu, int threads) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int wg = 256 * wg_per_cu;
  const int iters = 160000 / NACC;
  hipLaunchKernelGGL(k<NACC>, dim3(wg), dim3(threads), 0, 0, d, 100);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(wg), dim3(threads), 0, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wg * (threads / 64) * iters * NACC * 2048.0;
  printf("nacc=%2d threads=%4d wg/cu=%d (%d waves/SIMD): %.2f ms  %.1f TFLOP/s fp64 MFMA\n", NACC, threads, wg_per_cu,
         wg_per_cu * threads / 256, ms, flops / ms / 1e9);
}

int main() {
  double* d; (void)hipMalloc(&d, 256 * 8 * 1024 * 8);
  run<1>(d, 1, 256); run<2>(d, 1, 256); run<4>(d, 1, 256); run<8>(d, 1, 256); run<16>(d, 1, 256);
  run<8>(d, 2, 256); run<8>(d, 4, 256); run<2>(d, 4, 256); run<1>(d, 8, 256);
  run4<8>(d, 1); run4<16>(d, 1); run4<8>(d, 2); run4<8>(d, 4);
  return 0;
}
