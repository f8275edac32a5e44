// dev probe: what a tile hand-off between two workgroups costs when both sit on the SAME XCD (one L2) and when they do not,
// by the instructions used for the payload and the flag.  Two workgroups (block indices A and B of a larger grid, the rest
// exits at once) play ping-pong with a 2 KB payload + a flag, N round trips; the reader keeps the payload's lines warm in its
// L1 (plain loads before every wait), so a mode that may read a stale L1 / L2 line shows as "stale" counts.
//   mode 0  payload sc1 stores / sc1 loads, flag sc1 (agent-scope relaxed atomics): csrc/lrpost.hip's write-through hand-off
//   mode 1  payload plain stores / "buffer_inv sc0" + plain loads, flag sc0 store, "buffer_inv sc0" + plain load
//   mode 2  payload plain stores / sc0 loads, flag sc0 store / sc0 load
//   mode 3  payload plain stores / "buffer_inv sc0" + plain loads, flag sc1
//   mode 4  payload plain stores / sc1 loads, flag sc1
// hipcc --offload-arch=gfx950 -O3 -o xcd_handoff_probe xcd_handoff_probe.hip ; ./xcd_handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
#define CK(x) do { if ((x) != hipSuccess) { printf("hip error at line %d\n", __LINE__); exit(1); } } while (0)

#define LOAD4(SUF) asm volatile("global_load_dwordx2 %0, %4, off" SUF "\n\tglobal_load_dwordx2 %1, %4, off offset:512" SUF \
    "\n\tglobal_load_dwordx2 %2, %4, off offset:1024" SUF "\n\tglobal_load_dwordx2 %3, %4, off offset:1536" SUF "\n\ts_waitcnt vmcnt(0)" \
    : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p) : "memory")
#define STORE4(SUF) asm volatile("global_store_dwordx2 %0, %1, off" SUF "\n\tglobal_store_dwordx2 %0, %1, off offset:512" SUF \
    "\n\tglobal_store_dwordx2 %0, %1, off offset:1024" SUF "\n\tglobal_store_dwordx2 %0, %1, off offset:1536" SUF :: "v"(p), "v"(x) : "memory")
#define LOAD1(SUF) asm volatile("global_load_dword %0, %1, off" SUF "\n\ts_waitcnt vmcnt(0)" : "=&v"(r) : "v"(f) : "memory")
#define STORE1(SUF) asm volatile("global_store_dword %0, %1, off" SUF :: "v"(f), "v"(x) : "memory")

template <int MODE> __device__ __forceinline__ void flag_set(int* f, int x) {
  if (MODE == 1 || MODE == 2) STORE1(" sc0"); else STORE1(" sc1");
}
template <int MODE> __device__ __forceinline__ int flag_get(const int* f) {
  int r;
  if (MODE == 1) { asm volatile("buffer_inv sc0" ::: "memory"); LOAD1(""); }
  else if (MODE == 2) LOAD1(" sc0");
  else LOAD1(" sc1");
  return r;
}
template <int MODE> __device__ __forceinline__ void pay_store(double* p, double x) {
  if (MODE == 0) STORE4(" sc1"); else STORE4("");
}
template <int MODE> __device__ __forceinline__ void pay_load(const double* p, double (&v)[4]) {
  if (MODE == 0 || MODE == 4) LOAD4(" sc1");
  else if (MODE == 2) LOAD4(" sc0");
  else { asm volatile("buffer_inv sc0" ::: "memory"); LOAD4(""); }
}

// payload: 4 doubles per lane of one wave (2 KB)
template <int MODE>
__global__ void pingpong(int A, int B, int* flags, double* pay, int iters, long long* out, int* xcc) {
  const int me = blockIdx.x == A ? 0 : blockIdx.x == B ? 1 : -1;
  if (me < 0) return;
  __shared__ int s_abort;
  const int lane = threadIdx.x;
  if (lane == 0) s_abort = 0;
  __syncthreads();
  if (lane == 0) { int id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); xcc[me] = id & 0xf; }
  int* fmine = flags + me * 64;                     // (own line each)
  int* ftheir = flags + (1 - me) * 64;
  double* pmine = pay + me * 1024 + lane;
  double* ptheir = pay + (1 - me) * 1024 + lane;
  long long stale = 0, spins = 0;
  double warm = 0.0;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) {
      pay_store<MODE>(pmine, (double)i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) flag_set<MODE>(fmine, i);
    }
    {                                               // warm the L1 with the other side's (old) payload
      double v[4]; const double* p = ptheir; LOAD4("");
      warm += v[0] + v[1] + v[2] + v[3];
    }
    if (lane == 0) {
      long long mine = 0;
      while (flag_get<MODE>(ftheir) < i) {
        ++spins; ++mine; __builtin_amdgcn_s_sleep(1);
        if ((mine & 255) == 0 && (mine > 100000 || __hip_atomic_load(flags + 192, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {   // never hang the box
          __hip_atomic_store(flags + 192, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_abort = 1; break;
        }
      }
    }
    __syncthreads();
    if (s_abort) break;
    {
      double v[4]; pay_load<MODE>(ptheir, v);
      for (int t = 0; t < 4; ++t) if (v[t] != (double)i) ++stale;
    }
    if (me == 1) {
      pay_store<MODE>(pmine, (double)i);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) flag_set<MODE>(fmine, i);
    }
  }
  const long long t1 = wall_clock64();
  atomicAdd((u64*)&out[2 + me], (u64)stale);
  if (lane == 0) { out[me] = t1 - t0; out[4 + me] = spins; if (s_abort) out[6] = 1; }
  if (warm == 123.456) out[7] = 1;
}

template <int MODE> void run(int A, int B, int grid, int iters) {
  int* flags; double* pay; long long* out; int* xcc;
  CK(hipMalloc(&flags, 1024)); CK(hipMalloc(&pay, 2 * 1024 * 8)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&xcc, 8));
  CK(hipMemset(flags, 0, 1024)); CK(hipMemset(pay, 0, 2 * 1024 * 8)); CK(hipMemset(out, 0, 64)); CK(hipMemset(xcc, 0, 8));
  hipLaunchKernelGGL(pingpong<MODE>, dim3(grid), dim3(64), 0, 0, A, B, flags, pay, iters, out, xcc);
  CK(hipDeviceSynchronize());
  long long h[8]; int x[2];
  CK(hipMemcpy(h, out, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost));
  // wall_clock64: 100 MHz
  printf("mode %d blocks (%d, %d) xcc (%d, %d): %.3f us per hand-off, stale reads %lld + %lld, polls per wait %.1f%s\n", MODE, A, B,
         x[0], x[1], h[0] * 0.01 / iters / 2, h[2], h[3], (double)(h[4] + h[5]) / (2.0 * iters), h[6] ? "  ABORTED (a wait never saw its flag)" : "");
  CK(hipFree(flags)); CK(hipFree(pay)); CK(hipFree(out)); CK(hipFree(xcc));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int pairs[3][2] = {{0, 8}, {0, 1}, {0, 16}};
  for (int q = 0; q < 3; ++q) {
    const int A = pairs[q][0], B = pairs[q][1];
    run<0>(A, B, 64, iters);
    run<4>(A, B, 64, iters);
    run<3>(A, B, 64, iters);
    run<1>(A, B, 64, iters);
    run<2>(A, B, 64, iters);
  }
  return 0;
}
