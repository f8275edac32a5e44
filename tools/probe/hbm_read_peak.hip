// dev probe: what a bare streaming read reaches on this chip -- the ceiling the correlation scan is measured against in
// practice (the 8 TB/s of the data sheet is the roofline's denominator; this says how much of it any kernel can have).
// Every lane loads 16-byte pieces of a 20.48 GB buffer (the configs[3] matrix size) in a grid-stride loop with UNR loads
// in flight per lane, adds them up and writes one value per workgroup.  Variants: workgroups per CU, loads in flight,
// plain or nontemporal loads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int UNR, bool NT>
__global__ __launch_bounds__(256) void rd(const v4f* __restrict__ p, size_t n, float* out) {
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  v4f acc = {0, 0, 0, 0};
  for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
    v4f x[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) x[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += x[u];
  }
  for (; i < n; i += stride) acc += p[i];
  float s = acc.x + acc.y + acc.z + acc.w;
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, s);
}

// the scan's shape: a wave reads UNR consecutive KiB (a "row"), rows dealt round-robin over all waves of the grid
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void rd_rows(const v4f* __restrict__ p, size_t n, float* out) {
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (size_t)gridDim.x * 4, lane = threadIdx.x & 63;
  const size_t nrows = n / (64 * UNR);
  v4f acc = {0, 0, 0, 0};
  for (size_t r = wave; r < nrows; r += nwaves) {
    const v4f* q = p + r * (64 * UNR) + lane;
    v4f x[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) x[u] = NT ? __builtin_nontemporal_load(q + 64 * u) : q[64 * u];
#pragma unroll
    for (int u = 0; u < UNR; ++u) acc += x[u];
  }
  float s = acc.x + acc.y + acc.z + acc.w;
  for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) atomicAdd(out + blockIdx.x, s);
}
template <int UNR, bool NT> static double run_rows(const v4f* p, size_t n, float* out, int wg_per_cu) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL((rd_rows<UNR, NT>), dim3(grid), dim3(256), 0, 0, p, n, out);
  (void)hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rd_rows<UNR, NT>), dim3(grid), dim3(256), 0, 0, p, n, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)n * 16.0 * reps / (ms * 1e-3) / 1e9;
}

template <int UNR, bool NT> static double run(const v4f* p, size_t n, float* out, int wg_per_cu) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int grid = 256 * wg_per_cu;
  hipLaunchKernelGGL((rd<UNR, NT>), dim3(grid), dim3(256), 0, 0, p, n, out);
  (void)hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((rd<UNR, NT>), dim3(grid), dim3(256), 0, 0, p, n, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)n * 16.0 * reps / (ms * 1e-3) / 1e9;
}

int main() {
  const size_t bytes = 20480000000ull, n = bytes / 16;
  v4f* p; float* out;
  if (hipMalloc(&p, bytes) != hipSuccess) { printf("no memory\n"); return 1; }
  (void)hipMalloc(&out, 65536 * 4);
  (void)hipMemset(p, 0, bytes); (void)hipMemset(out, 0, 65536 * 4);
  double best = 0;
  const int wgs[] = {2, 4, 8, 16};
  for (int w : wgs) {
    const double a = run<2, false>(p, n, out, w), b = run<4, false>(p, n, out, w), c = run<8, false>(p, n, out, w);
    const double d = run<4, true>(p, n, out, w), e = run<8, true>(p, n, out, w);
    printf("%2d workgroups/CU: 2 loads in flight %7.1f GB/s, 4: %7.1f, 8: %7.1f | nontemporal 4: %7.1f, 8: %7.1f\n", w, a, b, c, d, e);
    const double m = fmax(fmax(fmax(a, b), fmax(c, d)), e);
    if (m > best) best = m;
  }
  for (int w : wgs) {
    const double a = run_rows<2, false>(p, n, out, w), b = run_rows<4, false>(p, n, out, w), c = run_rows<8, false>(p, n, out, w);
    const double d = run_rows<2, true>(p, n, out, w), e = run_rows<4, true>(p, n, out, w), f = run_rows<8, true>(p, n, out, w);
    printf("%2d workgroups/CU, a wave reads whole rows: 2 KiB rows %7.1f GB/s, 4 KiB: %7.1f, 8 KiB: %7.1f | nontemporal 2: %7.1f, 4: %7.1f, 8: %7.1f\n", w, a, b, c, d, e, f);
    const double m = fmax(fmax(fmax(a, b), fmax(c, d)), fmax(e, f));
    if (m > best) best = m;
  }
  printf("best bare read of 20.48 GB: %.1f GB/s = %.3f of 8 TB/s\n", best, best / 8000.0);
  return 0;
}
