// dev probe: does v_mfma_f64_4x4x4_4b_f64 honour the block-broadcast controls (CBSZ / ABID) on gfx950?
// Same method as mfma4_layout.hip (A lane l holds l + 1, B one-hot at lane m), once per (cbsz, abid) pair.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CBSZ, int ABID> __global__ void k(double* out) {
  const int lane = threadIdx.x;
  for (int m = 0; m < 64; ++m) {
    const double a = lane + 1.0, b = lane == m ? 1.0 : 0.0;
    const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, CBSZ, ABID, 0);
    out[m * 64 + lane] = d;
  }
}
template <int CBSZ, int ABID> void run(double* d) {
  hipLaunchKernelGGL((k<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, d);
  static double h[64 * 64];
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("cbsz %d abid %d\n", CBSZ, ABID);
  for (int m = 0; m < 64; m += 5) {      // a sample of B lanes
    printf("  B one-hot lane %2d:", m);
    for (int l = 0; l < 64; ++l) if (h[m * 64 + l] != 0.0) printf("  D[%2d]<-A[%2d]", l, (int)h[m * 64 + l] - 1);
    printf("\n");
  }
}
int main() {
  double* d; (void)hipMalloc(&d, 64 * 64 * 8);
  run<0, 0>(d); run<2, 0>(d); run<2, 1>(d); run<2, 2>(d); run<2, 3>(d); run<1, 0>(d); run<1, 1>(d);
  return 0;
}
