// dev probe: what v_mfma_f64_4x4x4_4b_f64 sustains when the wave also streams operands from LDS, in the shape of the
// projection kernel's compute loop: groups of 16 MFMAs (8 accumulators, each updated twice), R ds_read_b128 per group
// requested one group ahead (double-buffered), 8 groups per "stage", optionally a workgroup barrier per stage.
// 2 workgroups of 4 waves per CU (2 waves per SIMD), 128 accumulator VGPRs as in the kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double pv2d __attribute__((ext_vector_type(2)));

template <int R, bool BAR>
__global__ __launch_bounds__(256, 2) void kmix(double* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 32768 / 8; i += 256) ((double*)lds)[i] = 1e-6 * i;
  __syncthreads();
  double a[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) a[i] = 0.0;
  const unsigned char* base = lds + lane * 16;      // conflict-free: 64 consecutive 16-byte slots
  constexpr int RB = R > 0 ? R : 1;
  pv2d buf[2][RB];
  const pv2d regop = {tid * 1e-3, tid * 2e-3 + 1.0};
#pragma unroll
  for (int r = 0; r < RB; ++r) buf[0][r] = R > 0 ? *(const pv2d*)(base + r * 1024) : regop;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (R > 0) {
#pragma unroll
        for (int r = 0; r < RB; ++r) buf[(g + 1) & 1][r] = *(const pv2d*)(base + (((g + 1) * RB + r) & 31) * 1024);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const pv2d op = buf[g & 1][m % RB], ob = buf[g & 1][(m + 1) % RB];
        a[(g & 3) * 8 + m] = __builtin_amdgcn_mfma_f64_4x4x4f64(op.x, ob.x, a[(g & 3) * 8 + m], 0, 0, 0);
      }
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const pv2d op = buf[g & 1][m % RB], ob = buf[g & 1][(m + 1) % RB];
        a[(g & 3) * 8 + m] = __builtin_amdgcn_mfma_f64_4x4x4f64(op.y, ob.y, a[(g & 3) * 8 + m], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int R, bool BAR> void run(double* d) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int wg = 512, iters = 2000;
  (void)hipFuncSetAttribute((const void*)kmix<R, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL((kmix<R, BAR>), dim3(wg), dim3(256), 65536, 0, d, 10);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((kmix<R, BAR>), dim3(wg), dim3(256), 65536, 0, d, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)wg * 4 * iters * 128 * 512.0;
  printf("reads/group=%d barrier=%d: %.2f ms  %.1f TFLOP/s fp64 MFMA\n", R, (int)BAR, ms, flops / ms / 1e9);
}
int main() {
  double* d; (void)hipMalloc(&d, 512 * 256 * 8);
  run<0, false>(d); run<1, false>(d); run<2, false>(d); run<4, false>(d); run<5, false>(d); run<8, false>(d);
  run<0, true>(d); run<4, true>(d); run<5, true>(d);
  return 0;
}
