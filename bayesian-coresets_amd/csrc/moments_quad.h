// moments_quad.h -- the closing kernel of the closed-form column sums (csrc/moments.hip: what it computes and why) as a device
// function, so that it can run as a launch of its own (moments_quad_kernel) or beside the projection of the coreset points in
// ONE launch (csrc/proj.hip proj_mid_quad_kernel: both only need the draws, and each is a chain of dependent round trips to
// memory that leaves most of the chip idle).
#pragma once
#include "bcx_internal.h"
#include "dev_util.h"

typedef double mv4d __attribute__((ext_vector_type(4)));

// write-through (sc1) store / sc1 load: coherent across the XCDs' L2s whatever the reader's L2 holds (csrc/nnls_common.h)
static __device__ __forceinline__ double mom_ld(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
static __device__ __forceinline__ void mom_st(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// T_s = sum_i (Delta G)[s][i] (delta_si + 2 tbar_i) - 2 delta_si g_i   (= delta_s^T G delta_s + 2 delta_s^T (G tbar - g)),
// Delta G on the fp64 matrix cores: one WORKGROUP per 16 samples x 16 columns tile of Delta G, its four waves take the
// k steps 4 t + wave of the D rows of G (v_mfma_f64_16x16x4_f64; lane group lk feeds k = 8 t + 2 lk (+1) to steps 2 t (2 t + 1):
// one 16-byte load of theta per two steps) and meet in LDS in wave order -- the chain of dependent load batches is a
// quarter as long as with one wave per tile (this kernel is all latency: 46 MFLOP).
// Partials [column tile][sample] go to `work`; workgroup 0 sums them in tile order as they appear, centres and scales.
// work: nct * Spad partials (zero between calls: a zero is "not written yet"), then one word that is no longer used.
#define MQ_U 12
// sc1 load without a wait of its own: the caller issues a batch, waits once (s_waitcnt vmcnt(0)) and ties the values
// (atomic loads are issued and awaited one by one: nct round trips in the closing sum)
static __device__ __forceinline__ double mom_ld_nowait(const double* p) {
  double v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
struct MqArgs {
  const double* M; int64_t ldm; int D, ycol;
  const double* theta; int S, ldt;
  const double* tbar; double sigsq;
  double* colsum; double* work; int nct, Spad, dbg;
};
static inline void mq_plan(int D, int S, int* nct, int* Spad) { *nct = (D + 15) / 16; *Spad = (S + 15) / 16 * 16; }
// workgroup bx of the nblk that take part (a launch of its own: all of its workgroups; inside another launch: a share of them)
template <bool AL>
static __device__ __forceinline__ void moments_quad_body(const MqArgs& q, const int bx, const int nblk) {
  const double* __restrict__ M = q.M; const int64_t ldm = q.ldm; const int D = q.D, ycol = q.ycol;
  const double* __restrict__ theta = q.theta; const int S = q.S, ldt = q.ldt;
  const double* __restrict__ tbar = q.tbar; const double sigsq = q.sigsq;
  double* __restrict__ colsum = q.colsum; double* __restrict__ work = q.work; const int nct = q.nct, Spad = q.Spad, dbg = q.dbg;
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ double red[4][4][64];
  long long stamp[8];
#define MQ_STAMP(i) do { if (dbg) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); stamp[i] = wall_clock64(); } } while (0)
  MQ_STAMP(0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int st = bx / nct, ct = bx - st * nct;
  {
    const int sa = st * 16 + li, cb = ct * 16 + li;
    const bool va = sa < S, vb = cb < D;
    const double* th = theta + (size_t)(va ? sa : 0) * ldt;
    const double* gc = M + (vb ? cb : 0);
    mv4d acc = (mv4d){0.0, 0.0, 0.0, 0.0};
    // what wave 0 needs after the products, requested now (f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg)
    double e_tb = 0.0, e_gy = 0.0, e_th[4] = {0.0, 0.0, 0.0, 0.0};
    if (wave == 0 && vb) {
      e_tb = tbar[cb];
      e_gy = M[(size_t)ycol * ldm + cb];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int sr = st * 16 + lk + 4 * r;
        e_th[r] = theta[(size_t)(sr < S ? sr : 0) * ldt + cb];
      }
    }
    const int ksteps = (D + 7) / 8;
    for (int t0 = wave; t0 < ksteps; t0 += 4 * MQ_U) {
      // MQ_U double-steps per trip (t0, t0 + 4, ...): all their loads are issued before the first MFMA (addresses clamped,
      // values masked); D <= 384 is one trip -- one round of load latency instead of three
      double a0[MQ_U], a1[MQ_U], b0[MQ_U], b1[MQ_U];
#pragma unroll
      for (int u = 0; u < MQ_U; ++u) {
        const int k = 8 * (t0 + 4 * u) + 2 * lk;
        const bool k0 = k < D, k1 = k + 1 < D;
        const int kc = k0 ? k : 0, kd = k1 ? k + 1 : 0;
        double x0, x1;
        if (AL) { const double2 v = *(const double2*)(th + kc); x0 = v.x; x1 = v.y; }   // (kc + 1 <= ldt - 1: ldt is even)
        else { x0 = th[kc]; x1 = th[kd]; }
        const double t0v = tbar[kc], t1v = tbar[kd];
        const double g0 = gc[(size_t)kc * ldm], g1 = gc[(size_t)kd * ldm];
        a0[u] = (va && k0) ? x0 - t0v : 0.0;
        a1[u] = (va && k1) ? x1 - t1v : 0.0;
        b0[u] = (vb && k0) ? g0 : 0.0;
        b1[u] = (vb && k1) ? g1 : 0.0;
      }
#pragma unroll
      for (int u = 0; u < MQ_U; ++u) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc, 0, 0, 0);
      }
    }
    MQ_STAMP(1);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
      // f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
      const double tb = e_tb, gy = e_gy;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double y4 = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
        const int sr = st * 16 + lk + 4 * r;
        double v = 0.0;
        if (vb && sr < S) {
          const double dl = e_th[r] - tb;
          v = y4 * (dl + 2.0 * tb) - 2.0 * dl * gy;
        }
        v += bcx_dpp_f64<0xB1>(v);     // sum over the 16 lanes (columns) of the row group
        v += bcx_dpp_f64<0x4E>(v);
        v += bcx_dpp_f64<0x141>(v);
        v += bcx_dpp_f64<0x140>(v);
        // a partial is its own "ready": the scratch is zero between calls and a partial never is (+0.0 leaves as -0.0,
        // which adds the same)
        if (li == 0) mom_st(&work[(size_t)ct * Spad + sr], __double_as_longlong(v) == 0 ? -0.0 : v);
      }
    }
  }
  MQ_STAMP(2);
  if (bx != 0) return;
  // Workgroup 0 closes: it loads every sample's partials (sc1) until none of them is the zero the scratch held -- no arrival
  // counter, no drained stores, no fences, and the loads that wait are the loads that sum (tile order: the same bits
  // whoever finishes last) -- then centres, scales, and leaves the scratch zero for the next call.
  MQ_STAMP(3);
  __shared__ int late;
  if (tid == 0) late = 0;
  __syncthreads();
  double m[1] = {0.0};
  const long long t0 = wall_clock64();
  // (the padded samples S .. Spad - 1 are waited for too: their producers' stores must have landed before the scratch is
  // zeroed again below)
  for (int u = tid; u < Spad; u += 256) {
    double t = 0.0;
    for (int c0 = 0; c0 < nct; c0 += 16) {
      double v[16];
      for (;;) {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = mom_ld_nowait(work + (size_t)(c0 + q < nct ? c0 + q : c0) * Spad + u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          asm volatile("" : "+v"(v[q]));
          ok &= __double_as_longlong(v[q]) != 0;
        }
        if (ok) break;
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > 200000000LL) { late = 1; break; }      // 2 s: a producer never ran -- NaN out, loudly
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) if (c0 + q < nct) t += v[q];
    }
    if (u < S) {
      work[u] = t;                                  // (this thread's own slot of tile 0: read back below by the same thread)
      m[0] += t;
    }
  }
  __syncthreads();
  MQ_STAMP(4);
  block_allsum<1>(m, scratch);
  const double mean = late ? __longlong_as_double(0x7ff8000000000000LL) : m[0] / (double)S, f = -1.0 / (2.0 * sigsq);
  for (int u = tid; u < S; u += 256) colsum[u] = f * (work[u] - mean);
  __syncthreads();
  for (int e = tid; e < nct * Spad; e += 256) work[e] = 0.0;
  if (dbg) {
    MQ_STAMP(5);
    if (tid == 0) {
      double* o = work + (size_t)nct * Spad + 1;      // (the space of the library's own thetabar: dev runs pass one)
      for (int i = 0; i < 6; ++i) o[i] = (double)(stamp[i] - (i ? stamp[0] : 0)) * 0.01;
    }
  }
}

