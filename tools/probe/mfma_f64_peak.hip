// dev probe: peak rate of v_mfma_f64_16x16x4_f64 on this chip (register-only loop)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  v4d a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
  double x = threadIdx.x * 1e-3, y = threadIdx.x * 2e-3 + 1.0;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a4, 0, 0, 0);
    a5 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a5, 0, 0, 0);
    a6 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a6, 0, 0, 0);
    a7 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a7, 0, 0, 0);
  }
  v4d s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
  double* d; (void)hipMalloc(&d, 256 * 2048 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int wg = 256; wg <= 2048; wg *= 2) {
    const int iters = 20000;
    k<<<wg, 256>>>(d, 100);
    (void)hipEventRecord(e0);
    k<<<wg, 256>>>(d, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)wg * 4 * iters * 8 * 2048.0;
    printf("wg=%d: %.2f ms  %.1f TFLOP/s fp64 MFMA\n", wg, ms, flops / ms / 1e9);
  }
  return 0;
}
