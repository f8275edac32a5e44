// scan_core.h -- the pieces of the correlation scan shared by scan.hip (one launch per greedy iteration) and persist.hip
// (several iterations per launch): storage types, streaming loads, the cross-lane sums, the interval arithmetic, the
// per-lane top-2 tracker and scan_body, one pass of a workgroup over its rows.
#pragma once
#include <stdlib.h>
#include <math.h>
#include <hip/hip_fp16.h>
#include "bcx_internal.h"
#include "dev_util.h"

#ifndef BCX_LOADS_IN_FLIGHT
#define BCX_LOADS_IN_FLIGHT 8   // independent 16-byte loads per lane per trip
#endif
#ifndef BCX_UR_MAX
#define BCX_UR_MAX 4
#endif


// Storage types of the scanned matrix.  fp16 storage (half_t) halves the bytes per greedy iteration; it
// accumulates in fp32 against an fp32 query and relies on the same interval + fp64 re-score machinery.
struct half_t {};
struct QH { float4 lo, hi; };   // the 8 fp32 query values matching one 16-byte piece of 8 halves
template <typename ST> struct Stor;
template <> struct Stor<float>  { typedef float4 V;  typedef float4 Q;  typedef float T;  static constexpr int EPL = 4; };
template <> struct Stor<double> { typedef double2 V; typedef double2 Q; typedef double T; static constexpr int EPL = 2; };
template <> struct Stor<half_t> { typedef uint4 V;   typedef QH Q;      typedef float T;  static constexpr int EPL = 8; };

__device__ __forceinline__ float vdot(const float4& a, const float4& b, float acc) {
  acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
  return acc;
}
__device__ __forceinline__ double vdot(const double2& a, const double2& b, double acc) {
  acc = fma(a.x, b.x, acc); acc = fma(a.y, b.y, acc);
  return acc;
}
__device__ __forceinline__ float vdot(const uint4& a, const QH& b, float acc) {
  const float2 f0 = __half22float2(*(const __half2*)&a.x), f1 = __half22float2(*(const __half2*)&a.y);
  const float2 f2 = __half22float2(*(const __half2*)&a.z), f3 = __half22float2(*(const __half2*)&a.w);
  acc = fmaf(f0.x, b.lo.x, acc); acc = fmaf(f0.y, b.lo.y, acc); acc = fmaf(f1.x, b.lo.z, acc); acc = fmaf(f1.y, b.lo.w, acc);
  acc = fmaf(f2.x, b.hi.x, acc); acc = fmaf(f2.y, b.hi.y, acc); acc = fmaf(f3.x, b.hi.z, acc); acc = fmaf(f3.y, b.hi.w, acc);
  return acc;
}
// 16-byte streaming load of matrix data that no other workgroup touches again this launch: the
// non-temporal hint (global_load_dwordx4 ... nt) keeps the one-shot stream from evicting useful lines
typedef unsigned nt_u4 __attribute__((ext_vector_type(4)));
template <typename V> __device__ __forceinline__ V stream_load(const V* p) {
#ifdef BCX_NO_NT
  return *p;
#else
  static_assert(sizeof(V) == 16, "16-byte pieces");
  const nt_u4 r = __builtin_nontemporal_load((const nt_u4*)p);
  V v;
  __builtin_memcpy(&v, &r, 16);
  return v;
#endif
}

// query piece v of query `which` (0/1); qstride = distance between the two queries in pieces
template <typename ST> __device__ __forceinline__ typename Stor<ST>::Q load_q(const void* q, int piece, bool ok) {
  typedef typename Stor<ST>::Q Q;
  if constexpr (sizeof(Q) == 32) {
    QH r;
    r.lo = ok ? ((const float4*)q)[2 * piece] : make_float4(0.f, 0.f, 0.f, 0.f);
    r.hi = ok ? ((const float4*)q)[2 * piece + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
    return r;
  } else {
    // (load from a clamped index, then clear: written as `ok ? q[piece] : zero` hipcc selects between the ADDRESSES -- the
    // zero lands in scratch memory and the query is fetched with four flat dword loads)
    Q r = ((const Q*)q)[ok ? piece : 0];
    if (!ok) memset(&r, 0, sizeof r);
    return r;
  }
}

// ---- cross-lane sums without LDS traffic (gfx950: DPP inside a row of 16 lanes, v_permlane16_swap /
// v_permlane32_swap across rows).  Lane mappings verified with tools/probe/lane_probe.hip.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// lanes 0-31 <- a[l] + a[l+32], lanes 32-63 <- b[l-32] + b[l]
__device__ __forceinline__ float swap32_add(float a, float b) {
  const v2u r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  const unsigned x = r.x, y = r.y;   // (indexing r[0]/r[1] through a bit_cast is miscompiled by ROCm 7.2: both read element 0)
  return __uint_as_float(x) + __uint_as_float(y);
}
// rows of 16 lanes: (a.r0+a.r1, b.r0+b.r1, a.r2+a.r3, b.r2+b.r3)
__device__ __forceinline__ float swap16_add(float a, float b) {
  const v2u r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  const unsigned x = r.x, y = r.y;   // (indexing r[0]/r[1] through a bit_cast is miscompiled by ROCm 7.2: both read element 0)
  return __uint_as_float(x) + __uint_as_float(y);
}
// the same exchanges on doubles: low and high words travel separately
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) { return bcx_dpp_f64<CTRL>(v); }
__device__ __forceinline__ double swap32_add(double a, double b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
  const v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  const unsigned xl = lo.x, yl = lo.y, xh = hi.x, yh = hi.y;
  return __longlong_as_double((long long)(((unsigned long long)xh << 32) | xl)) +
         __longlong_as_double((long long)(((unsigned long long)yh << 32) | yl));
}
__device__ __forceinline__ double swap16_add(double a, double b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
  const v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  const unsigned xl = lo.x, yl = lo.y, xh = hi.x, yh = hi.y;
  return __longlong_as_double((long long)(((unsigned long long)xh << 32) | xl)) +
         __longlong_as_double((long long)(((unsigned long long)yh << 32) | yl));
}
template <int G> __device__ __forceinline__ float group_allsum_f32(float v) {
  if (G >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  if (G >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror
  if (G >= 16) v += dpp_mov<0x140>(v);  // row_mirror
  if (G >= 32) v = swap16_add(v, v);
  if (G >= 64) v = swap32_add(v, v);
  return v;
}
// Four wave-wide partial sums -> one register: the 16-lane row r of the result holds, in every lane,
// the total of input {0, 2, 1, 3}[r].  10 VALU instructions for four 64-lane reductions.
// (generic in T: the float and double overloads of the exchanges above must both be declared before this point --
// a later double overload would silently route doubles through the float versions)
template <typename T> __device__ __forceinline__ T reduce4_rows(T p0, T p1, T p2, T p3) {
  static_assert(sizeof(decltype(swap32_add(p0, p1))) == sizeof(T), "exchange overload narrows");
  const T m01 = swap32_add(p0, p1);
  const T m23 = swap32_add(p2, p3);
  T m = swap16_add(m01, m23);
  m += dpp_mov<0xB1>(m);
  m += dpp_mov<0x4E>(m);
  m += dpp_mov<0x141>(m);
  m += dpp_mov<0x140>(m);
  return m;
}

// xor-4 exchange inside a row of 16 lanes: reverse the quads, then mirror the halves (i -> 7-i -> i^4)
template <typename T> __device__ __forceinline__ T xor4_mov(T v) { return dpp_mov<0x141>(dpp_mov<0x1B>(v)); }

// The same transposed reduction for rows shorter than a wave (G lanes per row, 64/G rows per wave-wide load):
// four row steps p0..p3 are folded into ONE register in which every group of G/4 lanes holds the total of a
// different (step, row) pair -- at each of the first two butterfly levels a lane keeps the half it is going to
// own and hands the other half to its partner, so two registers become one; the remaining levels are plain
// DPP adds.  One interval + one arg-max update per FOUR row steps instead of one per step.
//   lane -> (step u, row-in-load rsub): pack_map<G>
template <int G, typename T> __device__ __forceinline__ T reduce4_pack(T p0, T p1, T p2, T p3) {
  const int lane = __lane_id();
  if constexpr (G == 64) {
    return reduce4_rows(p0, p1, p2, p3);
  } else if constexpr (G == 32) {
    const T m01 = swap16_add(p0, p1), m23 = swap16_add(p2, p3);
    const bool h8 = (lane & 8) != 0;
    T m = (h8 ? m23 : m01) + dpp_mov<0x128>(h8 ? m01 : m23);   // row_ror:8 == xor 8 inside a 16-lane row
    m += dpp_mov<0xB1>(m);
    m += dpp_mov<0x4E>(m);
    m += dpp_mov<0x141>(m);
    return m;
  } else if constexpr (G == 16) {
    const bool h8 = (lane & 8) != 0, h4 = (lane & 4) != 0;
    const T b0 = (h8 ? p1 : p0) + dpp_mov<0x128>(h8 ? p0 : p1);
    const T b1 = (h8 ? p3 : p2) + dpp_mov<0x128>(h8 ? p2 : p3);
    T m = (h4 ? b1 : b0) + xor4_mov(h4 ? b0 : b1);
    m += dpp_mov<0xB1>(m);
    m += dpp_mov<0x4E>(m);
    return m;
  } else if constexpr (G == 8) {
    const bool h4 = (lane & 4) != 0, h2 = (lane & 2) != 0;
    const T b0 = (h4 ? p1 : p0) + xor4_mov(h4 ? p0 : p1);
    const T b1 = (h4 ? p3 : p2) + xor4_mov(h4 ? p2 : p3);
    T m = (h2 ? b1 : b0) + dpp_mov<0x4E>(h2 ? b0 : b1);
    m += dpp_mov<0xB1>(m);
    return m;
  } else {   // G == 4
    const bool h2 = (lane & 2) != 0, h1 = (lane & 1) != 0;
    const T b0 = (h2 ? p1 : p0) + dpp_mov<0x4E>(h2 ? p0 : p1);
    const T b1 = (h2 ? p3 : p2) + dpp_mov<0x4E>(h2 ? p2 : p3);
    return (h1 ? b1 : b0) + dpp_mov<0xB1>(h1 ? b0 : b1);
  }
}
template <int G> __device__ __forceinline__ void pack_map(int lane, int& u, int& rsub) {
  const int r = lane >> 4;
  if constexpr (G == 64) { u = (r & 1) * 2 + (r >> 1); rsub = 0; }
  else if constexpr (G == 32) { u = (r & 1) + 2 * ((lane >> 3) & 1); rsub = r >> 1; }
  else if constexpr (G == 16) { u = ((lane >> 3) & 1) + 2 * ((lane >> 2) & 1); rsub = r; }
  else if constexpr (G == 8) { u = ((lane >> 2) & 1) + 2 * ((lane >> 1) & 1); rsub = lane >> 3; }
  else { u = ((lane >> 1) & 1) + 2 * (lane & 1); rsub = lane >> 2; }
}

template <typename T, int G> __device__ __forceinline__ T group_allsum(T v) {
  if constexpr (sizeof(T) == 4) {
    return group_allsum_f32<G>(v);
  } else {
    if (G >= 2) v += bcx_dpp_f64<0xB1>(v);
    if (G >= 4) v += bcx_dpp_f64<0x4E>(v);
    if (G >= 8) v += bcx_dpp_f64<0x141>(v);
    if (G >= 16) v += bcx_dpp_f64<0x140>(v);
    if (G >= 32) v += bcx_xor16_f64(v);
    if (G >= 64) v += bcx_xor32_f64(v);
    return v;
  }
}

struct ScanArgs {
  const void* An;     // n x ldv vectors of 16 bytes
  const void* q;      // 2 x ldv vectors (query 0, query 1) in storage precision
  const DevState* st;
  PartialView out;
  const double* norms; // non-null: rows are RAW fp64 rows, divide the dot products by the row norm
  int64_t n;
  int ldv;            // row stride in 16-byte vectors
  int qstride;        // distance between query 0 and query 1 in 16-byte vectors
  int nvec;           // 16-byte vectors that hold data in a row (<= ldv)
  int nv_lds;         // scan_long_kernel: query pieces [0, nv_lds) rest in LDS, the others are read from global memory (L2)
  float err_coef;     // |fp32 score - exact score| <= err_coef * qscale   (0 for fp64 storage)
};

// host side: the launch plan of one scan (scan.hip)
struct ScanPlan {
  int G, CH, UR, grid;
  bool f64, f16, dual, long_rows;
};
int bcx_scan_plan(bcx_solver* s, int exact, ScanArgs* a, ScanPlan* pl);

template <typename T> struct Track {
  T U1, U2, U3, L;
  int i1, i2;
};

template <typename T> __device__ __forceinline__ bool better(T ua, int ia, T ub, int ib) {
  return ua > ub || (ua == ub && ia < ib);
}

template <typename T> __device__ __forceinline__ Track<T> merge(const Track<T>& a, const Track<T>& b) {
  const bool bf = better<T>(b.U1, b.i1, a.U1, a.i1);
  const Track<T>& p = bf ? b : a;  // holds the overall best
  const Track<T>& q = bf ? a : b;
  Track<T> r;
  r.U1 = p.U1; r.i1 = p.i1;
  T third;
  if (better<T>(p.U2, p.i2, q.U1, q.i1)) { r.U2 = p.U2; r.i2 = p.i2; third = q.U1; }
  else { r.U2 = q.U1; r.i2 = q.i1; third = p.U2 > q.U2 ? p.U2 : q.U2; }
  T u3 = p.U3 > q.U3 ? p.U3 : q.U3;
  r.U3 = third > u3 ? third : u3;
  r.L = a.L > b.L ? a.L : b.L;
  return r;
}

template <typename T> __device__ __forceinline__ Track<T> shfl_track(const Track<T>& t, int off) {
  Track<T> r;
  r.U1 = __shfl_xor(t.U1, off, BCX_WAVE); r.U2 = __shfl_xor(t.U2, off, BCX_WAVE);
  r.U3 = __shfl_xor(t.U3, off, BCX_WAVE); r.L = __shfl_xor(t.L, off, BCX_WAVE);
  r.i1 = __shfl_xor(t.i1, off, BCX_WAVE); r.i2 = __shfl_xor(t.i2, off, BCX_WAVE);
  return r;
}

// Interval of the GIGA score given s0,s1 known to +-e (fp32 path).  Rows whose |s1| may reach 1
// (parallel to the current iterate: the singular mask of giga.py:33-36) get U = +inf so that the
// fp64 re-score decides.
__device__ __forceinline__ void giga_interval(float s0, float s1, float e, float& U, float& L) {
  const float a = fabsf(s1);
  const float ahi = a + e;
  const bool sure = ahi < 1.0f;          // false also for NaN
  const float ah = sure ? ahi : 0.5f;    // keep the arithmetic finite on the masked branch
  const float alo = fmaxf(a - e, 0.0f);
  // 1 - x^2 as (1-x)(1+x): the subtraction is exact for x >= 0.5
  const float rmin = __builtin_amdgcn_rsqf((1.0f - ah) * (1.0f + ah));   // 1 / smallest possible denominator
  const float rmax = __builtin_amdgcn_rsqf((1.0f - alo) * (1.0f + alo)); // 1 / largest possible denominator
  const float hi = s0 + e, lo = s0 - e;
  float u = hi > 0.0f ? hi * rmin : hi * rmax;
  float l = lo > 0.0f ? lo * rmax : lo * rmin;
  u += fabsf(u) * 4e-6f + 1e-37f;  // slack for the approximate rsq and the rounding of this arithmetic
  l -= fabsf(l) * 4e-6f + 1e-37f;
  U = sure ? u : INFINITY;
  L = sure ? l : -INFINITY;
}
// Exact-mode score, same masking as giga.py:33-38.
__device__ __forceinline__ double giga_score(double s0, double s1) {
  const bool ok = (s1 > -1.0 + 1e-14) && (1.0 - s1 * s1 > 0.0);
  const double den = ok ? sqrt(1.0 - s1 * s1) : INFINITY;
  return s0 / den;
}

// rows arrive in increasing order within a lane, so strict '>' keeps the lowest index on ties
template <typename T> __device__ __forceinline__ void track_update(Track<T>& tr, T U, T L, int ri) {
  const bool g1 = U > tr.U1, g2 = U > tr.U2, g3 = U > tr.U3;
  tr.U3 = g2 ? tr.U2 : (g3 ? U : tr.U3);
  tr.i2 = g1 ? tr.i1 : (g2 ? ri : tr.i2);
  tr.U2 = g1 ? tr.U1 : (g2 ? U : tr.U2);
  tr.i1 = g1 ? ri : tr.i1;
  tr.U1 = g1 ? U : tr.U1;
  tr.L = L > tr.L ? L : tr.L;
}

// One pass of a workgroup over its rows (row block blk of nblk, grid stride) for persist.hip: scan_kernel's arithmetic, row
// mapping and partials (scan.hip keeps its own text: the headline kernel's code does not change with this file), but the loads
// of the first trip are issued before gate() -- which waits for this iteration's query and returns false when the state machine
// has stopped -- and the query is fetched after it (sc1 loads; the partials leave as sc1 stores: no fences on this side).
// Returns false when the gate or the state machine said stop (no partials written).
template <typename ST, bool DUAL, int G, int CH, int UR, typename Gate>
__device__ __forceinline__ bool scan_body(const ScanArgs& a, const unsigned blk, const unsigned nblk, Gate gate) {
  typedef typename Stor<ST>::V V;
  typedef typename Stor<ST>::Q Q;
  typedef typename Stor<ST>::T T;
  constexpr int RPW = 64 / G;                       // rows per wave per step
  constexpr int WAVES = BCX_SCAN_THREADS / 64;
  constexpr int RPB = WAVES * RPW * UR;             // rows per workgroup per trip
  constexpr bool PACK4 = G >= 4 && (UR % 4 == 0);
  constexpr int GSZ = PACK4 ? G / 4 : G;          // lanes that end up tracking the same row
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G, rsub = lane / G;
  int myu = 0, myrs = 0;                            // PACK4: the (row step, row-in-load) pair this lane tracks
  if constexpr (PACK4) pack_map<G>(lane, myu, myrs);

  // query pieces for this lane's columns -> registers
  Q q0[CH], q1[CH];
  int voff[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { const int v = c * G + sub; voff[c] = v < a.nvec ? v : 0; }   // clamp: the load stays inside the row, the zero query kills the product
  T e = 0;

  Track<T> tr;
  tr.U1 = tr.U2 = tr.U3 = tr.L = -INFINITY;
  tr.i1 = tr.i2 = 0x7fffffff;

  const V* base = (const V*)a.An;
  const int64_t n = a.n;
  auto row_of = [&](int64_t r0, int u) { return r0 + (int64_t)(u * WAVES + wave) * RPW + rsub; };
  auto load_trip = [&](V (&x)[UR][CH], int64_t r0) {
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int64_t rw = row_of(r0, u);
      const int64_t rc = rw < n ? rw : n - 1;
      const V* p = base + rc * a.ldv;
#pragma unroll
      for (int c = 0; c < CH; ++c) x[u][c] = stream_load(p + voff[c]);
    }
  };
  auto compute_trip = [&](V (&x)[UR][CH], int64_t r0) {
    int64_t row[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      row[u] = row_of(r0, u);
      if constexpr (sizeof(T) == 8) {
        if (a.norms) {   // raw fp64 rows: An = A / Anorms element by element (giga.py:13)
          const double nr = a.norms[row[u] < n ? row[u] : n - 1];
#pragma unroll
          for (int c = 0; c < CH; ++c) { x[u][c].x /= nr; x[u][c].y /= nr; }
        }
      }
    }
    if constexpr (PACK4) {
      // four row steps at a time: transposed reduction, then every group of G/4 lanes tracks one row
#pragma unroll
      for (int g4 = 0; g4 < UR / 4; ++g4) {
        T a0[4], a1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          T t0 = 0, t1 = 0;
#pragma unroll
          for (int c = 0; c < CH; ++c) {
            t0 = vdot(x[g4 * 4 + u][c], q0[c], t0);
            if (DUAL) t1 = vdot(x[g4 * 4 + u][c], q1[c], t1);
          }
          a0[u] = t0; a1[u] = t1;
        }
        const T s0 = reduce4_pack<G, T>(a0[0], a0[1], a0[2], a0[3]);
        const T s1 = DUAL ? reduce4_pack<G, T>(a1[0], a1[1], a1[2], a1[3]) : (T)0;
        const int64_t myrow = r0 + (int64_t)((g4 * 4 + myu) * WAVES + wave) * RPW + myrs;
        T U, L;
        if constexpr (sizeof(T) == 4) {
          float Uf, Lf;
          if (DUAL) giga_interval((float)s0, (float)s1, (float)e, Uf, Lf);
          else { const float ee = (float)e + fabsf((float)s0) * 2e-7f; Uf = (float)s0 + ee; Lf = (float)s0 - ee; }
          U = Uf; L = Lf;
        } else {
          U = L = DUAL ? (T)giga_score((double)s0, (double)s1) : s0;   // exact mode: the score itself
        }
        if (!(myrow < n)) { U = -INFINITY; L = -INFINITY; }
        track_update<T>(tr, U, L, (int)myrow);
      }
    } else {
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        T s0 = 0, s1 = 0;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
          s0 = vdot(x[u][c], q0[c], s0);
          if (DUAL) s1 = vdot(x[u][c], q1[c], s1);
        }
        s0 = group_allsum<T, G>(s0);
        if (DUAL) s1 = group_allsum<T, G>(s1);
        T U, L;
        if (sizeof(T) == 4) {
          if (DUAL) {
            float Uf, Lf;
            giga_interval((float)s0, (float)s1, (float)e, Uf, Lf);
            U = Uf; L = Lf;
          } else {
            const T ee = e + fabsf((float)s0) * 2e-7f;
            U = s0 + ee; L = s0 - ee;
          }
        } else {
          U = L = DUAL ? (T)giga_score((double)s0, (double)s1) : s0;
        }
        if (!(row[u] < n)) { U = -INFINITY; L = -INFINITY; }
        track_update<T>(tr, U, L, (int)row[u]);
      }
    }
  };
  const int64_t stride = (int64_t)nblk * RPB;
  // (Two trips in flight per wave -- the loads of trip t+1 issued before trip t is reduced -- measured 3-5 points
  //  slower at every row length: this stream wants ~32 KiB in flight per CU, not more.)
  {
    // The rows do not depend on the query: the first trip's loads are issued, THEN the workgroup waits for the query of
    // this iteration (gate: the tail workgroup of the same launch publishes it, persist.hip) and fetches it.
    int64_t r0 = (int64_t)blk * RPB;
    V x[UR][CH];
    if (r0 < n) load_trip(x, r0);
    __builtin_amdgcn_sched_barrier(0);
    if (!gate()) return false;
    // The query, its scale and the state machine's switch were written by another workgroup, possibly on another XCD, while
    // this one was running: sc1 loads (served by the memory side, whatever this XCD's L2 holds from the last iteration) -- an
    // acquire fence here instead costs ~19 ns PER WAVE of the whole launch, one after the other (measured: 2048 waves, +39 us
    // per iteration).  The compiler does not count these loads: one wait for all of them (and for the trip above, which
    // is needed next anyway), then the values are tied to the wait before anything reads them.
    static_assert(sizeof(Q) == 16, "16-byte query pieces (fp32 / fp64 storage)");
    nt_u4 rq0[CH], rq1[CH];
    unsigned ract;
    v2u rqs;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const Q* p0 = (const Q*)a.q + voff[c];
      asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(rq0[c]) : "v"(p0) : "memory");
      if (DUAL) {
        const Q* p1 = p0 + a.qstride;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(rq1[c]) : "v"(p1) : "memory");
      }
    }
    {
      const int* pa = &a.st->active;
      const double* ps = &a.st->qscale;
      asm volatile("global_load_dword %0, %1, off sc1" : "=v"(ract) : "v"(pa) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(rqs) : "v"(ps) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      asm volatile("" : "+v"(rq0[c]));
      if (DUAL) asm volatile("" : "+v"(rq1[c]));
    }
    asm volatile("" : "+v"(ract));
    asm volatile("" : "+v"(rqs));
    if (!ract) return false;          // the state machine stopped (uniform: one word)
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      const bool ok = c * G + sub < a.nvec;
      __builtin_memcpy(&q0[c], &rq0[c], 16);
      if (!ok) memset(&q0[c], 0, sizeof(Q));
      if (DUAL) {
        __builtin_memcpy(&q1[c], &rq1[c], 16);
        if (!ok) memset(&q1[c], 0, sizeof(Q));
      }
    }
    {
      const unsigned long long qb = ((unsigned long long)rqs.y << 32) | rqs.x;
      e = (T)(a.err_coef * (float)__longlong_as_double((long long)qb));
    }
    if (r0 < n) {
      __builtin_amdgcn_sched_barrier(0);
      compute_trip(x, r0);
      for (r0 += stride; r0 < n; r0 += stride) {
        load_trip(x, r0);
        __builtin_amdgcn_sched_barrier(0);
        compute_trip(x, r0);
      }
    }
  }
  // combine the row groups of a wave (lanes with equal `sub` hold distinct row groups)
#pragma unroll
  for (int off = GSZ; off < 64; off <<= 1) tr = merge<T>(tr, shfl_track<T>(tr, off));
  __shared__ Track<T> wtr[WAVES];
  if (lane == 0) wtr[wave] = tr;
  __syncthreads();
  if (threadIdx.x == 0) {
    Track<T> r = wtr[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) r = merge<T>(r, wtr[w]);
    const int b = (int)blk;
    // write-through (sc1) stores: the tail's workgroup reads them from the memory side; the caller waits for them before its stamp
    auto st64 = [](double* p, double v) {
      __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    st64(a.out.U1 + b, (double)r.U1); st64(a.out.U2 + b, (double)r.U2); st64(a.out.U3 + b, (double)r.U3); st64(a.out.L + b, (double)r.L);
    __hip_atomic_store(a.out.i1 + b, r.i1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.out.i2 + b, r.i2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;
}
