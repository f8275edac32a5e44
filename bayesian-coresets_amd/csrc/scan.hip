// scan.hip -- the N-way correlation scan with fused arg-max: the one kernel that streams the
// N x d matrix of normalised rows every greedy iteration (HBM-bound, N*d*sizeof(elem) bytes).
//
//   FW / OMP : score[n] = An[n] . (b - A w)                         frankwolfe.py:16-17, orthopursuit.py:18-19
//   GIGA     : s0 = An[n] . cdir, s1 = An[n] . xw_hat,
//              score[n] = s0 / sqrt(1 - s1^2)  (masked -> s0/inf)   giga.py:31-38
//   arg-max with first-index tie-break (ndarray.argmax)
//
// Mapping (wave64): a group of G lanes owns one row; lane `sub` of the group loads the 16-byte
// pieces sub, sub+G, ... of that row (fully coalesced: one wave-wide load covers 1 KiB of
// contiguous row data) with the non-temporal hint, multiplies them with the matching query pieces
// held in REGISTERS (a lane always touches the same columns), and the partial sums are combined
// without LDS: for G = 64 four rows at a time by a transposed butterfly (v_permlane32_swap,
// v_permlane16_swap, 4 DPP adds), after which each 16-lane row of the wave tracks one data row.
// A workgroup of 4 waves walks the rows with a grid stride; UR row-steps are issued back to back
// (fenced) so every lane keeps CH*UR = 4-8 independent 16-byte loads in flight, and only 1-2
// workgroups are resident per CU (see scan_grid_for): the stream is fastest with few, deep waves.
// Rows may be stored as fp32 (default), fp16 (half the bytes; fp32 accumulation) or fp64.
//
// The fp32 scan does not decide the arg-max alone: every row gets a rigorous interval
// [L, U] around its exact score (forward error bound of the fp32 dot product), each workgroup
// reports its two largest U (with row ids), a bound U3 on all its other rows, and its largest L.
// resolve.hip re-scores in fp64 every row whose U reaches the global max L; if a workgroup's
// U3 reaches it too the iteration is redone with the exact kernel.  With fp64 storage U = L =
// score and the result is the arg-max itself.
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

template <typename ST, bool DUAL, int G, int CH, int UR>
__global__ __launch_bounds__(BCX_SCAN_THREADS) void scan_kernel(ScanArgs a) {
  if (!a.st->active) return;
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
  for (int c = 0; c < CH; ++c) {
    const int v = c * G + sub;
    const bool ok = v < a.nvec;
    voff[c] = ok ? v : 0;  // clamp: the load stays inside the row, the zero query kills the product
    q0[c] = load_q<ST>(a.q, v, ok);
    if (DUAL) q1[c] = load_q<ST>(a.q, a.qstride + v, ok);
  }
  const T e = (T)(a.err_coef * (float)a.st->qscale);

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
  const int64_t stride = (int64_t)gridDim.x * RPB;
  // (Two trips in flight per wave -- the loads of trip t+1 issued before trip t is reduced -- measured 3-5 points
  //  slower at every row length: this stream wants ~32 KiB in flight per CU, not more.)
  for (int64_t r0 = (int64_t)blockIdx.x * RPB; r0 < n; r0 += stride) {
    V x[UR][CH];
    load_trip(x, r0);
    // keep all CH*UR loads of the trip in flight: without this fence hipcc interleaves load/wait/FMA
    // through one register quad to save VGPRs, which serialises the HBM round trips of a wave
    __builtin_amdgcn_sched_barrier(0);
    compute_trip(x, r0);
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
    const int b = blockIdx.x;
    a.out.U1[b] = (double)r.U1; a.out.U2[b] = (double)r.U2; a.out.U3[b] = (double)r.U3; a.out.L[b] = (double)r.L;
    a.out.i1[b] = r.i1; a.out.i2[b] = r.i2;
  }
}

// ---- rows longer than 16 pieces per lane (nvec > 1024: d > 4096 floats / 2048 doubles) ----------------------------------
// The query no longer fits the register file beside the loads, so it rests in LDS (16 bytes per piece and query) and a
// wave walks ONE row in trips of 8 wave-wide loads (8 KiB of contiguous row data in flight per wave), multiplying each
// piece with its query piece read from LDS; one wave-wide reduction, interval and arg-max update per row (at >= 64 KiB a
// row that is noise).  Same scores to the last bit as the register form would give for the same row length?  No: the
// summation order differs (trip by trip instead of chunk-major), which only matters to the interval's error coefficient
// -- computed from the same chunk count -- and to exact-mode ties between different rows, which stay lowest-index-first.
#define BCX_LONG_LCH 8
template <typename ST, bool DUAL>
__global__ __launch_bounds__(BCX_SCAN_THREADS) void scan_long_kernel(ScanArgs a) {
  if (!a.st->active) return;
  typedef typename Stor<ST>::V V;
  typedef typename Stor<ST>::Q Q;
  typedef typename Stor<ST>::T T;
  constexpr int WAVES = BCX_SCAN_THREADS / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char scan_qlds[];
  Q* ql0 = (Q*)scan_qlds;
  Q* ql1 = ql0 + a.nv_lds;
  const int nvl = a.nv_lds;
  for (int v = threadIdx.x; v < nvl; v += blockDim.x) {
    ql0[v] = load_q<ST>(a.q, v, true);
    if (DUAL) ql1[v] = load_q<ST>(a.q, a.qstride + v, true);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const T e = (T)(a.err_coef * (float)a.st->qscale);
  Track<T> tr;
  tr.U1 = tr.U2 = tr.U3 = tr.L = -INFINITY;
  tr.i1 = tr.i2 = 0x7fffffff;
  const V* base = (const V*)a.An;
  const int64_t n = a.n;
  const int nvec = a.nvec;
  if constexpr (!DUAL) {
    // The (row, trip) steps of a wave form one sequence, and the loads of step i + 1 are issued before step i is consumed
    // (two register sets, the loop unrolled by two): the queue never drains between the trips of a row or between rows.
    // Single-query scans (Frank-Wolfe, OMP) only: N = 150k rows, d = 8192 floats 0.80-0.82 -> 0.84 of the HBM peak, d = 20000
    // 0.87 -> 0.89, against the trip-by-trip loop below, which the two-query GIGA scan keeps (pipelined it fell to 0.70).
    const int ntrip = (nvec + 64 * BCX_LONG_LCH - 1) / (64 * BCX_LONG_LCH);
    const int64_t rstride = (int64_t)gridDim.x * WAVES;
    V xa[BCX_LONG_LCH], xb[BCX_LONG_LCH];
    double nra = 1.0, nrb = 1.0, nr = 1.0;          // row norm (raw fp64 rows): fetched with a row's first trip, into its register set's slot
    T s0 = 0, s1 = 0;
    auto issue = [&](int64_t row, int trip, V (&x)[BCX_LONG_LCH], double& nrx) {
      const V* p = base + row * a.ldv;
      const int v0 = trip * 64 * BCX_LONG_LCH;
  #pragma unroll
      for (int c = 0; c < BCX_LONG_LCH; ++c) {
        const int v = v0 + c * 64 + lane;
        x[c] = stream_load(p + (v < nvec ? v : 0));
      }
      if constexpr (sizeof(T) == 8) { if (trip == 0 && a.norms) nrx = a.norms[row]; }
    };
    auto consume = [&](int64_t row, int trip, V (&x)[BCX_LONG_LCH], double nrx) {
      if (trip == 0) { nr = nrx; s0 = 0; s1 = 0; }
      const int v0 = trip * 64 * BCX_LONG_LCH;
  #pragma unroll
      for (int c = 0; c < BCX_LONG_LCH; ++c) {
        const int v = v0 + c * 64 + lane;
        if (v < nvec) {
          if constexpr (sizeof(T) == 8) {
            if (a.norms) { x[c].x /= nr; x[c].y /= nr; }     // raw fp64 rows: An = A / Anorms element by element (giga.py:13)
          }
          // (rows beyond the LDS budget of the query: its tail comes from global memory -- the query is L2 resident)
          s0 = vdot(x[c], v < nvl ? ql0[v] : load_q<ST>(a.q, v, true), s0);
          if (DUAL) s1 = vdot(x[c], v < nvl ? ql1[v] : load_q<ST>(a.q, a.qstride + v, true), s1);
        }
      }
      if (trip != ntrip - 1) return;
      T t0 = group_allsum<T, 64>(s0), t1 = 0;
      if (DUAL) t1 = group_allsum<T, 64>(s1);
      T U, L;
      if (sizeof(T) == 4) {
        if (DUAL) {
          float Uf, Lf;
          giga_interval((float)t0, (float)t1, (float)e, Uf, Lf);
          U = Uf; L = Lf;
        } else {
          const T ee = e + fabsf((float)t0) * 2e-7f;
          U = t0 + ee; L = t0 - ee;
        }
      } else {
        U = L = DUAL ? (T)giga_score((double)t0, (double)t1) : t0;
      }
      track_update<T>(tr, U, L, (int)row);
    };
    auto next = [&](int64_t& row, int& trip) { if (++trip == ntrip) { trip = 0; row += rstride; } };
    int64_t row = (int64_t)blockIdx.x * WAVES + wave;
    int trip = 0;
    if (row < n) issue(row, 0, xa, nra);
    while (row < n) {
      int64_t r1 = row; int t1 = trip;
      next(r1, t1);
      if (r1 < n) issue(r1, t1, xb, nrb);
      __builtin_amdgcn_sched_barrier(0);       // the next step's loads are in flight before this step's first use
      consume(row, trip, xa, nra);
      if (!(r1 < n)) break;
      row = r1; trip = t1;
      next(r1, t1);
      if (r1 < n) issue(r1, t1, xa, nra);
      __builtin_amdgcn_sched_barrier(0);
      consume(row, trip, xb, nrb);
      row = r1; trip = t1;
    }
  } else {
    for (int64_t row = (int64_t)blockIdx.x * WAVES + wave; row < n; row += (int64_t)gridDim.x * WAVES) {
      const V* p = base + row * a.ldv;
      double nr = 1.0;
      if constexpr (sizeof(T) == 8) { if (a.norms) nr = a.norms[row]; }
      T s0 = 0, s1 = 0;
      for (int v0 = 0; v0 < nvec; v0 += 64 * BCX_LONG_LCH) {
        V x[BCX_LONG_LCH];
  #pragma unroll
        for (int c = 0; c < BCX_LONG_LCH; ++c) {
          const int v = v0 + c * 64 + lane;
          x[c] = stream_load(p + (v < nvec ? v : 0));
        }
        __builtin_amdgcn_sched_barrier(0);       // all loads of the trip in flight before the first use
  #pragma unroll
        for (int c = 0; c < BCX_LONG_LCH; ++c) {
          const int v = v0 + c * 64 + lane;
          if (v < nvec) {
            if constexpr (sizeof(T) == 8) {
              if (a.norms) { x[c].x /= nr; x[c].y /= nr; }     // raw fp64 rows: An = A / Anorms element by element (giga.py:13)
            }
            // (rows beyond the LDS budget of the query: its tail comes from global memory -- the query is L2 resident)
            s0 = vdot(x[c], v < nvl ? ql0[v] : load_q<ST>(a.q, v, true), s0);
            if (DUAL) s1 = vdot(x[c], v < nvl ? ql1[v] : load_q<ST>(a.q, a.qstride + v, true), s1);
          }
        }
      }
      s0 = group_allsum<T, 64>(s0);
      if (DUAL) s1 = group_allsum<T, 64>(s1);
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
      track_update<T>(tr, U, L, (int)row);
    }
  }
  __shared__ Track<T> wtr[WAVES];            // (every lane of a wave holds the same track)
  if (lane == 0) wtr[wave] = tr;
  __syncthreads();
  if (threadIdx.x == 0) {
    Track<T> r = wtr[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) r = merge<T>(r, wtr[w]);
    const int b = blockIdx.x;
    a.out.U1[b] = (double)r.U1; a.out.U2[b] = (double)r.U2; a.out.U3[b] = (double)r.U3; a.out.L[b] = (double)r.L;
    a.out.i1[b] = r.i1; a.out.i2[b] = r.i2;
  }
}

template <typename ST, bool DUAL> static int launch_long(bcx_solver* s, const ScanArgs& a_in, int grid) {
  ScanArgs a = a_in;
  const size_t per = sizeof(typename Stor<ST>::Q) * (DUAL ? 2 : 1);
  a.nv_lds = (int)std::min<size_t>((size_t)a.nvec, (144 * 1024) / per);
  const size_t lds = (size_t)a.nv_lds * per;
  if (lds > 48 * 1024)
    BCX_HIP(hipFuncSetAttribute((const void*)scan_long_kernel<ST, DUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((scan_long_kernel<ST, DUAL>), dim3(grid), dim3(BCX_SCAN_THREADS), lds, s->stream, a);
  return BCX_OK;
}

// ---- host side ----------------------------------------------------------------------------
static int pick_group(int nvec) {  // lanes per row
  int g = 1;
  while (g < 64 && g < nvec) g <<= 1;
  return g;
}

// Launch width.  Measured on MI355X (tools/scan_sweep*.sh, interleaved on one box): this stream runs
// fastest with FEW resident waves that each keep many loads in flight -- about 32 KiB of outstanding
// 16-byte loads per CU (4 waves x 8 loads, or 8 waves x 4 loads): 6.8-7.0 TB/s, against 6.0-6.4 TB/s
// with the 8 workgroups per CU one would launch by reflex.  So: grid = 256 CUs x (8 / loads per lane).
int bcx_scan_grid(const bcx_solver* s) { return BCX_MAX_PARTIALS; }   // capacity of the partial arrays

static int scan_grid_for(int64_t n, int rows_per_block, double loads_per_lane, bool heavy) {
  int64_t want = (n + rows_per_block - 1) / rows_per_block;
  if (want < 1) want = 1;
  // 256 CUs x (8 / useful loads per lane), in steps of one workgroup per CU
  if (loads_per_lane < 1.0) loads_per_lane = 1.0;
  if (loads_per_lane > 8.0) loads_per_lane = 8.0;
  int64_t cap = (int64_t)floor(2048.0 / loads_per_lane / 256.0 + 0.5) * 256;
  if (cap < 256) cap = 256;
  if (heavy) cap = BCX_MAX_PARTIALS;   // fp64 GIGA is VALU-heavy (fp64 sqrt/divide per row): it wants the occupancy
  if (const char* e = bcx_dev_env("BCX_SCAN_GRID")) { const long v = atol(e); if (v > 0) cap = v; }
  if (cap > BCX_MAX_PARTIALS) cap = BCX_MAX_PARTIALS;
  if (want > cap) want = cap;
  return (int)want;
}

// (G, CH, UR) menu.  UR row steps are in flight per lane; the default keeps CH * UR = BCX_LOADS_IN_FLIGHT
// 16-byte loads per lane, the "deep" variant doubles UR for row lengths that leave many lanes of the last
// chunk idle (d = 300: 75 pieces in 128 slots), so the USEFUL bytes in flight stay the same.
template <typename T, bool DUAL>
static int launch_t(bcx_solver* s, const ScanArgs& a, int G, int CH, int UR, int grid) {
  dim3 g(grid), b(BCX_SCAN_THREADS);
#define L(GG, CC, UU)                                                                                \
  if (G == GG && CH == CC && UR == UU) {                                                             \
    hipLaunchKernelGGL((scan_kernel<T, DUAL, GG, CC, UU>), g, b, 0, s->stream, a);                   \
    return BCX_OK;                                                                                   \
  }
  L(1, 1, 4) L(2, 1, 4) L(4, 1, 4) L(8, 1, 4) L(16, 1, 4) L(32, 1, 4) L(64, 1, 4) L(64, 2, 4) L(64, 4, 2) L(64, 8, 1) L(64, 16, 1)
  L(1, 1, 8) L(2, 1, 8) L(4, 1, 8) L(8, 1, 8) L(16, 1, 8) L(32, 1, 8) L(64, 1, 8) L(64, 2, 8) L(64, 4, 4) L(64, 8, 2)
#undef L
  s->err = "scan: unsupported row length";
  return BCX_ERR_ARG;
}

int bcx_launch_scan(bcx_solver* s, int exact) {
  // storage fp64            : fp64 kernel over the normalised rows
  // storage fp32 / fp16, exact == 0: fp32-accumulating kernel (interval scan)
  // storage fp32 / fp16, exact == 1: fp64 kernel over the RAW rows (A64) when they are resident, else the
  //                           low-precision kernel again (resolve then takes its arg-max as is)
  const int d = s->cfg.d;
  const int sd = s->cfg.store_dtype;
  const bool raw64 = exact && sd != BCX_F64 && s->A64 != nullptr;
  const bool f64 = (sd == BCX_F64) || raw64;
  const bool f16 = !f64 && sd == BCX_F16;
  ScanArgs a;
  a.st = s->st;
  a.n = s->cfg.n_local;
  a.norms = nullptr;
  const int epl = f64 ? 2 : (f16 ? 8 : 4);
  if (raw64) {
    a.An = s->A64;
    a.norms = s->norms;
    a.q = s->q64;
    a.ldv = s->ld64 / 2;
    a.qstride = s->ld64 / 2;
  } else {
    a.An = s->An;
    a.q = s->qst;
    a.ldv = s->ld / epl;
    a.qstride = s->ld / epl;
  }
  const int nvec = (d + epl - 1) / epl;
  a.nvec = nvec;
  const int G = pick_group(nvec);
  int CH = (nvec + G - 1) / G;
  int chp = 1;
  while (chp < CH) chp <<= 1;
  CH = chp;
  // forward error bound of the dot product, times |q|:
  //   fp32 storage: rounding of row and query (2u) + summation depth (EPL*CH fused multiply-adds, log2 G
  //                 butterfly adds), u = 2^-24; sum|a_i q_i| <= |a||q|.
  //   fp16 storage: |a^_i - a_i| <= 2^-11 |a_i| + 2^-25 (subnormal spacing; |a_i| <= 1), so the storage term
  //                 is 2^-11 |q| + 2^-25 sqrt(d) |q|, plus the fp32 terms above.
  // 30% head room on the fp32 part, 2% on the fp16 storage term.
  const double u = 5.9604644775390625e-08;
  int lg = 0;
  while ((1 << lg) < G) ++lg;
  double coef = 1.3 * u * ((double)epl * CH + lg + 3.0);
  if (f16) coef += 1.02 * (4.8828125e-4 + 2.9802322387695312e-08 * sqrt((double)d));
  a.err_coef = f64 ? 0.0f : (float)coef;
  const bool dual = s->cfg.alg == BCX_ALG_GIGA;
  // row steps in flight: CH * UR = BCX_LOADS_IN_FLIGHT slots per lane, twice that when fewer than ~70 % of the
  // slots carry data (idle lanes in the last chunk / in the row group)
  int ur = (CH >= BCX_LOADS_IN_FLIGHT) ? 1 : (BCX_LOADS_IN_FLIGHT / CH > BCX_UR_MAX ? BCX_UR_MAX : BCX_LOADS_IN_FLIGHT / CH);
  const double util = (double)nvec / ((double)G * CH);
  static const int deep_env = bcx_dev_env("BCX_SCAN_DEEP") ? atoi(bcx_dev_env("BCX_SCAN_DEEP")) : -1;   // dev: force 0 / 1
  // Row lengths that leave lanes of the group idle (util < 0.85; d = 100: 25 of 32 lanes, d = 300: 75 of 128 slots) run
  // best with TWO workgroups per CU and 4.5 - 6 USEFUL loads in flight per lane: the row steps are doubled only when
  // the base depth carries fewer than 4.5 (interleaved A/B on one box, tools/scan_knobs.sh, % of 8 TB/s, FW / GIGA:
  // d = 100 81/80 -> 84/82, d = 200 81/81 -> 84/81, d = 300 82/76 -> 86/83).  Full groups keep one workgroup per CU.
  // Shorter groups (G <= 16) and longer rows (CH >= 4) keep the earlier rule (row steps doubled for G = 64, CH >= 2 below
  // 80 % lane use, launch width from the useful loads per lane): the two-workgroup form measured slower there.
  const bool ragged = util < 0.85 && G >= 32 && CH <= 2;
  bool deep;
  if (deep_env >= 0) deep = CH < 16 && deep_env == 1;
  else if (ragged) deep = (double)CH * ur * util < 4.5;
  else deep = CH < 16 && G == 64 && CH >= 2 && util < 0.80;
  if (deep) ur *= 2;
  const int rpb = (BCX_SCAN_THREADS / 64) * (64 / G) * ur;
  int grid = scan_grid_for(a.n, rpb, CH * ur * util, f64 && dual);   // idle lanes do not count as loads in flight
  if (ragged && !(f64 && dual) && !bcx_dev_env("BCX_SCAN_GRID")) {
    const int64_t want = (a.n + rpb - 1) / rpb;
    grid = (int)(want < 512 ? (want < 1 ? 1 : want) : 512);
  }
  const bool long_rows = nvec > 64 * 16;     // beyond the (G, CH) menu: one wave per row, query in LDS
  if (long_rows) {
    const int64_t want = (a.n + 3) / 4;
    grid = (int)(want < 512 ? (want < 1 ? 1 : want) : 512);
  }
  s->n_partials = grid;                     // resolve reads exactly this launch's partials
  a.out = partial_view(s->partials, grid);
  int rc;
  if (long_rows) {
    if (f64) rc = dual ? launch_long<double, true>(s, a, grid) : launch_long<double, false>(s, a, grid);
    else if (f16) rc = dual ? launch_long<half_t, true>(s, a, grid) : launch_long<half_t, false>(s, a, grid);
    else rc = dual ? launch_long<float, true>(s, a, grid) : launch_long<float, false>(s, a, grid);
  } else if (f64) rc = dual ? launch_t<double, true>(s, a, G, CH, ur, grid) : launch_t<double, false>(s, a, G, CH, ur, grid);
  else if (f16) rc = dual ? launch_t<half_t, true>(s, a, G, CH, ur, grid) : launch_t<half_t, false>(s, a, G, CH, ur, grid);
  else rc = dual ? launch_t<float, true>(s, a, G, CH, ur, grid) : launch_t<float, false>(s, a, G, CH, ur, grid);
  if (rc != BCX_OK) return rc;
  BCX_HIP(hipGetLastError());
  return BCX_OK;
}
