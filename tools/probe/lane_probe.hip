// dev probe: print the lane mappings of permlane32_swap / permlane16_swap / DPP controls on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ int dppmov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
__global__ void probe(int* out) {
  int l = threadIdx.x;
  v2u r = __builtin_amdgcn_permlane32_swap((unsigned)(l), (unsigned)(100 + l), false, false);
  out[l] = r[0]; out[64 + l] = r[1];
  v2u q = __builtin_amdgcn_permlane16_swap((unsigned)(l), (unsigned)(100 + l), false, false);
  out[128 + l] = q[0]; out[192 + l] = q[1];
  out[256 + l] = dppmov<0xB1>(l);
  out[320 + l] = dppmov<0x4E>(l);
  out[384 + l] = dppmov<0x141>(l);
  out[448 + l] = dppmov<0x140>(l);
}
int main() {
  int* d; hipMalloc(&d, 512 * 4);
  probe<<<1, 64>>>(d);
  int h[512]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[] = {"swap32.r0", "swap32.r1", "swap16.r0", "swap16.r1", "qp1032", "qp2301", "half_mirror", "mirror"};
  for (int k = 0; k < 8; ++k) { printf("%s:", names[k]); for (int l = 0; l < 64; ++l) printf(" %d", h[k * 64 + l]); printf("\n"); }
  return 0;
}
