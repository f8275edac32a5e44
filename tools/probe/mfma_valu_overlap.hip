// dev probe: do fp64 VALU instructions (v_fma_f64) and fp64 MFMAs (v_mfma_f64_4x4x4_4b_f64) overlap on one SIMD of this chip?
// A 512-thread workgroup per CU = two waves per SIMD.  Waves 0-3 run an MFMA loop (8 independent accumulators), waves
// 4-7 an FMA loop (8 independent chains); each half is timed alone and both together.  Together ~ max(alone) means the
// epilogue of one resident workgroup can hide under the MFMAs of the other; together ~ sum means the two share the
// fp64 datapath (or its issue port) and an epilogue can only be made shorter, not hidden.
// Third form: ONE wave per SIMD with both instruction kinds interleaved in its stream.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ __launch_bounds__(512) void both(double* out, int im, int iv) {
  const int wave = threadIdx.x >> 6;
  double x = threadIdx.x * 1e-3, y = threadIdx.x * 2e-3 + 1.0;
  double a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = 0.0;
  if (wave < 4) {
    for (int i = 0; i < im; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a[j], 0, 0, 0);
    }
  } else {
    for (int i = 0; i < iv; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = __builtin_fma(a[j], x, y);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// one wave per SIMD: per iteration 8 MFMAs and NV FMAs in one stream
template <int NV>
__global__ __launch_bounds__(256) void mixed(double* out, int it) {
  double x = threadIdx.x * 1e-3, y = threadIdx.x * 2e-3 + 1.0;
  double a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = 0.0; b[i] = i; }
  for (int i = 0; i < it; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = __builtin_amdgcn_mfma_f64_4x4x4f64(x, y, a[j], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) b[(j + v) & 7] = __builtin_fma(b[(j + v) & 7], x, y);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + b[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float timeit(void (*launch)(double*, int, int), double* d, int a, int b) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(d, a > 0 ? 10 : 0, b > 0 ? 10 : 0);
  (void)hipEventRecord(e0);
  launch(d, a, b);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
static void l_both(double* d, int a, int b) { hipLaunchKernelGGL(both, dim3(256), dim3(512), 0, 0, d, a, b); }
template <int NV> static void l_mixed(double* d, int a, int) { hipLaunchKernelGGL(mixed<NV>, dim3(256), dim3(256), 0, 0, d, a); }

int main() {
  double* d; (void)hipMalloc(&d, 256 * 512 * 8);
  const int im = 200000;            // 8 MFMAs x 16 cycles per iteration
  for (int ratio = 1; ratio <= 4; ratio *= 2) {
    // FMA loop sized to `1/ratio` of the MFMA loop's issue time if an fp64 FMA takes 4 cycles per wave
    const int iv = im * 4 / ratio;  // 8 FMAs x 4 cycles per iteration
    const float tm = timeit(l_both, d, im, 0), tv = timeit(l_both, d, 0, iv), tb = timeit(l_both, d, im, iv);
    printf("two waves per SIMD: MFMA alone %.2f ms (%.1f TFLOP/s), FMA alone %.2f ms (%.1f TFLOP/s vector), together %.2f ms  [sum %.2f, max %.2f]\n",
           tm, 256.0 * 4 * im * 8 * 512 / tm / 1e9, tv, 256.0 * 4 * iv * 8 * 128 / tv / 1e9, tb, tm + tv, tm > tv ? tm : tv);
  }
  const float t0 = timeit(l_mixed<0>, d, im, 0), t1 = timeit(l_mixed<1>, d, im, 0), t2 = timeit(l_mixed<2>, d, im, 0), t4 = timeit(l_mixed<4>, d, im, 0);
  printf("one wave per SIMD, per MFMA k interleaved FMAs: k=0 %.2f ms, k=1 %.2f, k=2 %.2f, k=4 %.2f  (an MFMA occupies its pipe 16 cycles, an FMA 4)\n", t0, t1, t2, t4);
  return 0;
}
