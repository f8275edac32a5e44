// dev probe: operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4) on gfx950.
// A lane l holds value l + 1; B is one-hot at lane m.  D is then non-zero exactly in the lanes of B's block that sit in
// B's column, and the value names the A lane that supplied A[i][k_m]: prints, for every m, the (D lane <- A lane) pairs.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  for (int m = 0; m < 64; ++m) {
    const double a = lane + 1.0, b = lane == m ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    out[m * 64 + lane] = d;
  }
}
int main() {
  double* d; (void)hipMalloc(&d, 64 * 64 * 8);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  double h[64 * 64];
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int m = 0; m < 64; ++m) {
    printf("B one-hot lane %2d:", m);
    for (int l = 0; l < 64; ++l) if (h[m * 64 + l] != 0.0) printf("  D[%2d]<-A[%2d]", l, (int)h[m * 64 + l] - 1);
    printf("\n");
  }
  return 0;
}
