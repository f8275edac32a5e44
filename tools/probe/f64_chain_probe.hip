// dev probe: issue / dependent latencies of the fp64 primitives the in-wave Cholesky of csrc/lrpost.hip is made of, one wave,
// shader-clock cycles per operation (clock64 around unrolled sequences).  hipcc --offload-arch=gfx950 -O3 -o f64_chain_probe f64_chain_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double v, int l) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)b, l), hi = __builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
#define PIN(v) asm volatile("" : "+v"(v))
__device__ __forceinline__ long long tick() { long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__global__ void probe(double* out, long long* cyc, double seed) {
  __shared__ double s[256];
  const int lane = threadIdx.x;
  s[lane] = seed + lane; s[lane + 64] = seed * 0.5 + lane;
  __syncthreads();
  double a[16];
  for (int i = 0; i < 16; ++i) a[i] = seed + i + lane * 1e-3;
  long long t0, t1;
  // 1: 256 independent FMAs (16 chains)
  __builtin_amdgcn_sched_barrier(0); t0 = tick(); __builtin_amdgcn_sched_barrier(0);
  for (int i = 0; i < 16; ++i) PIN(a[i]);
#pragma unroll
  for (int r = 0; r < 16; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fma(a[i], 1.0000001, 0.5);
  for (int i = 0; i < 16; ++i) PIN(a[i]);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0); t1 = tick(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[0] = t1 - t0;
  // 2: 256 dependent FMAs
  double x = a[0];
  __builtin_amdgcn_sched_barrier(0); t0 = tick(); __builtin_amdgcn_sched_barrier(0);
  PIN(x);
#pragma unroll
  for (int r = 0; r < 256; ++r) x = fma(x, 1.0000001, 0.5);
  PIN(x);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0); t1 = tick(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[1] = t1 - t0;
  // 3: 64 dependent rsq
  double y = fabs(x) + 1.0;
  __builtin_amdgcn_sched_barrier(0); t0 = tick(); __builtin_amdgcn_sched_barrier(0);
  PIN(y);
#pragma unroll
  for (int r = 0; r < 64; ++r) y = __builtin_amdgcn_rsq(y + 1.0);
  PIN(y);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0); t1 = tick(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[2] = t1 - t0;
  // 4: 128 broadcast 16-byte LDS reads, independent, consumed by FMAs
  double acc = 0.0;
  __builtin_amdgcn_sched_barrier(0); t0 = tick(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 64; ++r) { const double2 v = *(const double2*)&s[2 * r]; acc = fma(v.x, 1.5, acc); acc = fma(v.y, 2.5, acc); }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0); t1 = tick(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[3] = t1 - t0;
  // 5: 32 dependent LDS write -> read round trips (lane-crossing)
  double z = acc;
  __builtin_amdgcn_sched_barrier(0); t0 = tick(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 32; ++r) { s[128 + lane] = z; __builtin_amdgcn_wave_barrier(); z = s[128 + ((lane + 1) & 63)] + 1.0; __builtin_amdgcn_wave_barrier(); }
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0); t1 = tick(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[4] = t1 - t0;
  // 6: 64 dependent readlane -> fma hops
  double w = z;
  __builtin_amdgcn_sched_barrier(0); t0 = tick(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 64; ++r) w = fma(w, rl(w, r), 0.25);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0); t1 = tick(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[5] = t1 - t0;
  // 7: 64 independent (readlane pair + fma) on 16 accumulators from one source register
  __builtin_amdgcn_sched_barrier(0); t0 = tick(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 64; ++r) a[r & 15] = fma(-w, rl(w, r), a[r & 15]);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0); t1 = tick(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[6] = t1 - t0;
  // 8: 64 dependent MFMA 16x16x4 f64
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 c = {w, w, w, w};
  __builtin_amdgcn_sched_barrier(0); t0 = tick(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int r = 0; r < 64; ++r) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[r & 15], a[(r + 1) & 15], c, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_sched_barrier(0); t1 = tick(); __builtin_amdgcn_sched_barrier(0); if (lane == 0) cyc[7] = t1 - t0;
  double sum = x + y + acc + z + w + c[0] + c[1] + c[2] + c[3];
  for (int i = 0; i < 16; ++i) sum += a[i];
  out[lane] = sum;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 16 * 8);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc, 1.25);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
  const char* names[] = {"independent fma (16 chains)", "dependent fma", "dependent rsq (+add)", "broadcast ds_read_b128 + 2 fma", "LDS write->read round trip",
                         "dependent readlane->fma hop", "independent readlane pair + fma", "dependent mfma_f64_16x16x4"};
  const int n[] = {256, 256, 64, 64, 32, 64, 64, 64};
  for (int i = 0; i < 8; ++i) printf("%-36s %8lld cycles total  %7.1f per op\n", names[i], h[i], (double)h[i] / n[i]);
  return 0;
}
