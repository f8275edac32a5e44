// lrpost.hip -- the weighted conjugate posterior of Gaussian linear regression for ANY number of weighted points, from
// weights that live on the device (reference: examples/common/model_linreg.py:26-41 weighted_post, the sampler of
// examples/linear_regression/main.py:141-147):
//
//     P = Sig0^-1 + X^T diag(w) X / sigsq = L L^T,     U = L^-T  (Sigma_w = U U^T),     mu_w = Sigma_w (Sig0^-1 mu0 + X^T (w y) / sigsq),
//     draws  theta = mu_w + R U^T = mu_w + R L^-1      for standard-normal R  -- the reference's own factor and arithmetic.
//
// csrc/svi.hip serves a few points as a rank-k correction of the prior's factor inside ONE workgroup; beyond that the
// k x k system no longer fits a workgroup and, for k near D (the reference's experiment runs the coreset up to 300 points
// at D = 301), the D x D form is the smaller one.  SparseVI needs this factorisation once per ADAM step, 100 times per
// greedy step, each depending on the last (sparsevi.py:69-76), so what counts is the LATENCY of one D x D Cholesky +
// inverse -- 19 MFLOP behind a chain of D dependent pivots -- not its throughput:
//
//   lrp_form_kernel   P (lower triangle, 32 x 32 tiles) on the fp64 matrix cores and the right-hand side; marks every tile the
//                     next kernel's workgroups will hand to each other "not written yet".  One workgroup per tile of P.
//   lrp_chol_kernel   ONE launch of 3 + H co-resident workgroups: blocked Cholesky of the augmented matrix [P; I; rhs^T] --
//                     the row operations that turn P into L turn I into L^-T and rhs^T into (L^-1 rhs)^T, a forward
//                     substitution that rides along (no sweep after the factorisation).  LEFT-looking: tile (R, p) of block
//                     column p is  (init - sum_{q<p} tile(R, q) L_{p,q}^T) W_pp^T,  W_pp = L_pp^-1, formed in one go when
//                     column p's turn comes -- nothing is ever read-modified-written across steps.
//                       * workgroup 0, the CHAIN: wave 0 factors every diagonal tile (a row per lane; the identity rows in
//                         the other 32 lanes give W_pp for free; csrc/chol32.h), all four waves then form the sub-diagonal
//                         tile L_{p+1,p} and the next diagonal tile, so the critical path diag(p) -> diag(p+1) never leaves
//                         the workgroup;
//                       * workgroups 1 and 2, the ASSISTANTS, pre-accumulate for the chain the two sums over columns q <= p - 2
//                         of those tiles (they are ready a step early); while wave 0 factors, the chain's waves 1-3 add the
//                         one term of column p - 1;
//                       * H helpers share the other tiles of column p: they accumulate as the tiles of column p - 1 appear
//                         -- that overlaps diag(p) -- then multiply by W_pp as soon as it appears; finished tiles of L^-T
//                         leave transposed, as L^-1 row-major for the draw kernel.
//                     A tile that crosses workgroups carries its own readiness (a sentinel no arithmetic produces until its
//                     producer has stored it: no flags, counters or fences; see LP_WAIT), and when the launch fits one XCD
//                     (D <= 736) all of its workgroups are placed on XCD 0, where ordinary stores reach the one L2 the
//                     readers' sc1 loads read; otherwise the stores are write-through (sc1).  (First version: release /
//                     acquire fences at every hand-off -- a release fence behind 16 KB of fresh tiles costs ~6.5 us and sat
//                     twice per step on the critical path: 146 us for D = 301; now 62.)
//   the draws         theta = mu_w + [R; Rbar] L^-1: csrc/svi.hip's lrs_draw_kernel with L^-1 in the place of the prior's
//                     factor (k = 0).
//
// Tiles are stored "k-grouped": element (row, col) of a 32 x 32 tile at (col / 4) * 128 + row * 4 + col % 4, so that the
// 16 x 4 operand slices of v_mfma_f64_16x16x4_f64 -- lane l: row l % 16, k = l / 16 -- are 512 contiguous bytes per wave load;
// every product here has the form C (+)= A B^T with both operands read that way.
#include <atomic>
#include <cstdlib>
#include <string>
#include "bcx_internal.h"
#include "dev_util.h"
#include "chol32.h"

typedef double lp4d __attribute__((ext_vector_type(4)));

#define LP_NB 32
#define LP_TILE (LP_NB * LP_NB)
#define LP_MAX_NT 32                 // D <= 1024
#define LP_MAX_H 60
#define LP_FIXED_WGS 3               // the chain and its two assistants
#define LP_SENT 0xFFF7A5A5A5A5A5A5ull  // "not written yet" (see lrp_chol_kernel)

struct LpArgs {
  // inputs of the form kernel
  const double* w; const double* XT; const double* y;     // k weights, D x ldx features BY points (row a: feature a of the k points), k responses
  const double* S0inv; const double* rhs0;                 // D x lds0 prior precision, Sig0^-1 mu0
  double sigsq;
  int k, D, ldx, lds0;
  // work
  double* TT;     // ks * nt * nt tiles: P (lower block triangle, [i * nt + j]) as the SUM of ks slices (each from a share of the
                  // points); written by lrp_form_kernel, read-only afterwards
  double* LT;     // finished tiles of L (strictly lower block triangle)
  double* BL;     // (nt + 1) * nt finished tiles of the bottom part: rows 0 .. nt-1 L^-T (upper block triangle), row nt the row (L^-1 rhs)^T
  double* XW;     // nt tiles: the inverses of the diagonal tiles, W_pp = L_pp^-1 (rows c, k)
  double* AS;     // nt tiles: P_{p+1,p} - sum_{q <= p-2} L_{p+1,q} L_{p,q}^T        (assistant 1 -> chain)
  double* AD;     // nt tiles: P_{p+1,p+1} - sum_{q <= p-2} L_{p+1,q} L_{p+1,q}^T    (assistant 2 -> chain)
  double* rhs;    // 32 nt doubles: Sig0^-1 mu0 + X^T (w y) / sigsq, zero padded
  int* flags;     // [4 nt]: the status word (0 ok, 1 a wait expired, 2 a pivot was not positive); the words before it are not used
  // outputs
  double* U; int64_t ldu;      // D x ldu row-major: U = L^-T, upper triangular (the lower triangle is never written: the caller zeroes it once)
  double* uvec;                // D: u = L^-1 rhs  (mu_w = U u; the draws are U (u + r))
  double* mu;                  // D, or null: the mean is then not formed (lrp_mean_kernel is not launched)
  int nt, H, ks;
  long long timeout_ticks;
  long long* dbg;              // dev (BCX_LRP_DBG=1): wall-clock stamps [workgroup][step][8] (tools/lrp_timeline.py); else null
};

static __device__ __forceinline__ int lp_kg(int row, int col) { return (col >> 2) * 128 + row * 4 + (col & 3); }

// write-through (sc1) store / sc1 load of one double: coherent across XCDs without fences (the reader bypasses its L1, the
// writer's line leaves its L2)
static __device__ __forceinline__ double lp_ld(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
static __device__ __forceinline__ void lp_st(double* p, double v) {
  __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// operand slice of a k-grouped tile: rows 16 half .. 16 half + 15, all 32 values of the contraction index.  COH: the tile is
// (or may have been) written by another workgroup of this launch.
template <bool COH, int GS = 128>
static __device__ __forceinline__ void lp_rows(const double* tile, int half, int lane, double (&v)[8]) {
  const int o = (half * 16 + (lane & 15)) * 4 + (lane >> 4);
#pragma unroll
  for (int t = 0; t < 8; ++t) v[t] = COH ? lp_ld(tile + t * GS + o) : tile[t * GS + o];
}
// W_pp in LDS: groups of four columns 132 doubles apart, not 128 -- the factoring wave writes a COLUMN per lane, eight groups
// at once, which at 128 land in the same banks (8-way conflict on each of its 32 writes: 0.4 us per tile)
#define LP_WGS 132
static __device__ __forceinline__ lp4d lp_mma(const double (&a)[8], const double (&b)[8], lp4d acc) {
#pragma unroll
  for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[t], acc, 0, 0, 0);
  return acc;
}
static __device__ __forceinline__ lp4d lp_mma_neg(const double (&a)[8], const double (&b)[8], lp4d acc) {
#pragma unroll
  for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[t], b[t], acc, 0, 0, 0);
  return acc;
}
// quadrant (rb, cb) of a k-grouped tile in the accumulator layout: register r holds (row 16 rb + lane / 16 + 4 r, col 16 cb + lane % 16)
template <bool COH>
static __device__ __forceinline__ lp4d lp_quad_load(const double* tile, int rb, int cb, int lane) {
  lp4d c;
  const int col = cb * 16 + (lane & 15), row0 = rb * 16 + (lane >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = COH ? lp_ld(tile + lp_kg(row0 + 4 * r, col)) : tile[lp_kg(row0 + 4 * r, col)];
  return c;
}
// tile (i, j) of P: the sum of its slices, in slice order
static __device__ __forceinline__ lp4d lp_quad_load_p(const double* TT, int nt, int ks, int tile, int rb, int cb, int lane) {
  lp4d c = lp_quad_load<false>(TT + (size_t)tile * LP_TILE, rb, cb, lane);
  for (int s = 1; s < ks; ++s) {
    const lp4d d = lp_quad_load<false>(TT + ((size_t)s * nt * nt + tile) * LP_TILE, rb, cb, lane);
    c += d;
  }
  return c;
}
// a store another workgroup of the launch will read.  ONE: every workgroup of the launch sits on the same XCD -- one L2, which
// an ordinary store reaches by itself (the L1 is write-through) and the readers' sc1 loads read; otherwise write-through (sc1)
template <bool ONE>
static __device__ __forceinline__ void lp_gst(double* p, double v) {
  if (ONE) *p = v; else lp_st(p, v);
}
template <bool COH, bool ONE = false>
static __device__ __forceinline__ void lp_quad_store(double* tile, int rb, int cb, int lane, lp4d c) {
  const int col = cb * 16 + (lane & 15), row0 = rb * 16 + (lane >> 4);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (COH) lp_gst<ONE>(tile + lp_kg(row0 + 4 * r, col), c[r]);
    else tile[lp_kg(row0 + 4 * r, col)] = c[r];
  }
}

// ---- P and the right-hand side ------------------------------------------------------------------------------------------
// Workgroups 0 .. ks nt (nt + 1) / 2 - 1: slice s of tile (i, j), i >= j, of P = S0inv + X^T diag(w / sigsq) X (k-grouped, into TT):
// the runs s, s + ks, ... of 32 points (the kernel's time is one workgroup's chain of dependent loads, so the points are
// dealt over ks workgroups per tile and the readers add the slices in slice order); rows / columns >= D are padded with
// the identity.  The features arrive BY points (XT: row a = feature a of all k points), so
// both operands of the product are read along the contraction index: lane (i = lane % 16, g = lane / 16) takes the values
// 8 g .. 8 g + 7 of each run of 32 points as four 16-byte loads, one value per MFMA step (any assignment of the inner index
// to steps serves as long as both operands use the same one); three runs are in flight.  Workgroups after them, one per
// block of 32 columns: that block of the right-hand side; the first of them clears the status word.
typedef double lp2d __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void lp_fill_sent(double* tile, int tid) {
#pragma unroll
  for (int e = 0; e < 4; ++e) ((unsigned long long*)tile)[tid + 256 * e] = LP_SENT;
}
__global__ __launch_bounds__(256) void lrp_form_kernel(LpArgs a) {
  __shared__ double sw[4096 + 32];                  // w_j / sigsq (the second kind of workgroup: w_j y_j / sigsq), zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = a.nt, D = a.D, k = a.k;
  const int ntop = nt * (nt + 1) / 2;
  const int kp = (k + 31) & ~31;
  if ((int)blockIdx.x >= ntop * a.ks) {
    const int j = blockIdx.x - ntop * a.ks;
    if (j == 0) for (int e = tid; e < 4 * nt + 1; e += 256) a.flags[e] = 0;
    lp_fill_sent(a.BL + ((size_t)nt * nt + j) * LP_TILE, tid);
    lp_fill_sent(a.XW + (size_t)j * LP_TILE, tid);
    lp_fill_sent(a.AS + (size_t)j * LP_TILE, tid);
    lp_fill_sent(a.AD + (size_t)j * LP_TILE, tid);
    for (int e = tid; e < kp; e += 256) sw[e] = e < k ? fmax(a.w[e], 0.0) / a.sigsq * a.y[e] : 0.0;
    __syncthreads();
    // rhs0 + X^T (w y) / sigsq: a wave per column, lanes along the points
    for (int q = wave; q < 32; q += 4) {
      const int c = j * 32 + q;
      double s = 0.0;
      if (c < D) {
        const double* row = a.XT + (size_t)c * a.ldx;
        for (int jj = 2 * lane; jj < k; jj += 128) {
          const lp2d x = *(const lp2d*)(row + jj);
          s = fma(sw[jj], x.x, s);
          if (jj + 1 < k) s = fma(sw[jj + 1], x.y, s);
        }
        s = wave_allsum(s) + a.rhs0[c];
      }
      if (lane == 0) a.rhs[c] = s;
    }
    return;
  }
  const int slice = blockIdx.x / ntop;              // this workgroup's share of the points: runs of 32, dealt round robin
  int i = 0, t = blockIdx.x - slice * ntop;
  while (t >= i + 1) { t -= i + 1; ++i; }
  const int j = t;                                  // tile (i, j), j <= i
  if (slice == 0) {                                 // the tiles the next kernel's workgroups hand to each other: "not written yet"
    if (i > j) lp_fill_sent(a.LT + (size_t)(i * nt + j) * LP_TILE, tid);
    lp_fill_sent(a.BL + (size_t)(j * nt + i) * LP_TILE, tid);
  }
  for (int e = tid; e < kp; e += 256) sw[e] = e < k ? fmax(a.w[e], 0.0) / a.sigsq : 0.0;
  const int rb = wave >> 1, cb = wave & 1;
  const int li = lane & 15, lk = lane >> 4;
  const int ra = i * 32 + rb * 16 + li;             // the A operand's row of P (a feature), this lane
  const int rbcol = j * 32 + cb * 16 + li;          // the B operand's
  lp4d acc;
  {
    const int col = j * 32 + cb * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = i * 32 + rb * 16 + lk + 4 * r;
      acc[r] = slice ? 0.0 : ((row < D && col < D) ? a.S0inv[(size_t)row * a.lds0 + col] : (row == col ? 1.0 : 0.0));
    }
  }
  const double* pa = a.XT + (size_t)(ra < D ? ra : 0) * a.ldx + 8 * lk;
  const double* pb = a.XT + (size_t)(rbcol < D ? rbcol : 0) * a.ldx + 8 * lk;
  const bool aok = ra < D, bok = rbcol < D;
  // (ldx >= kp rounded to 32 points and the padding is zero: bcx_linreg_posterior_factor checks the stride, the caller pads)
  auto fetch = [&](int j0, double (&xa)[8], double (&xb)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const lp2d u = *(const lp2d*)(pa + j0 + 2 * q), v = *(const lp2d*)(pb + j0 + 2 * q);
      xa[2 * q] = aok ? u.x : 0.0; xa[2 * q + 1] = aok ? u.y : 0.0;
      xb[2 * q] = bok ? v.x : 0.0; xb[2 * q + 1] = bok ? v.y : 0.0;
    }
  };
  auto scale = [&](int j0, double (&xa)[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) xa[q] *= sw[j0 + 8 * lk + q];
  };
  // runs slice, slice + ks, slice + 2 ks, ... of 32 points; three in flight
  const int step = 32 * a.ks, first = 32 * slice;
  double xa[3][8], xb[3][8];
  if (first < kp) fetch(first, xa[0], xb[0]);
  if (first + step < kp) fetch(first + step, xa[1], xb[1]);
  __syncthreads();                                  // (sw)
  for (int j0 = first; j0 < kp; j0 += 3 * step) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int jc = j0 + step * u;
      if (jc < kp) {
        if (jc + 2 * step < kp) fetch(jc + 2 * step, xa[(u + 2) % 3], xb[(u + 2) % 3]);
        scale(jc, xa[u]);
        acc = lp_mma(xa[u], xb[u], acc);
      }
    }
  }
  lp_quad_store<false>(a.TT + ((size_t)slice * nt * nt + i * nt + j) * LP_TILE, rb, cb, lane, acc);
}

// ---- the factorisation ---------------------------------------------------------------------------------------------------
#define LP_STAMP(slot) do { if (a.dbg && (threadIdx.x & 63) == 0) a.dbg[((size_t)bid * a.nt + p) * 8 + (slot)] = wall_clock64(); } while (0)

// A tile that crosses workgroups carries its own "ready": lrp_form_kernel fills every such tile with LP_SENT, a signalling NaN
// that no arithmetic produces (results are quiet NaNs), the producer overwrites it with 8-byte write-through stores in any
// order, and a consumer simply loads its operands (sc1) until none of them is the sentinel -- one trip through the L2 (or, with
// the workgroups spread over the XCDs, through memory) per hand-off instead of three (drain the stores, raise a flag, see the
// flag, load), and no flag or counter at all.
static __device__ __forceinline__ int lp_set(double v) { return (unsigned long long)__double_as_longlong(v) != LP_SENT ? 1 : 0; }
static __device__ __forceinline__ int lp_set8(const double (&v)[8]) {
  int ok = 1;
#pragma unroll
  for (int t = 0; t < 8; ++t) ok &= lp_set(v[t]);
  return ok;
}
static __device__ __forceinline__ int lp_set4(lp4d c) { return lp_set(c[0]) & lp_set(c[1]) & lp_set(c[2]) & lp_set(c[3]); }
// A wave's waiting state (wave-uniform).  dead: a wait expired here or anywhere else -- no more waiting, the launch runs out
// with whatever it reads and the status word tells the host.
#ifndef LP_SLEEP
#define LP_SLEEP 1
#endif
struct LpSpin { long long t0; int n; bool dead; };
static __device__ __forceinline__ bool lp_again(LpSpin& sp, const LpArgs& a) {     // after a failed check; false: give up
  if (sp.t0 < 0) sp.t0 = wall_clock64();
  __builtin_amdgcn_s_sleep(LP_SLEEP);
  int* status = a.flags + 4 * a.nt;
  if (wall_clock64() - sp.t0 > a.timeout_ticks) {
    if ((threadIdx.x & 63) == 0) atomicCAS(status, 0, 1);
    sp.dead = true;
  } else if (((++sp.n) & 255) == 0 && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) sp.dead = true;
  return !sp.dead;
}
// The statements after PEEK_OK (re)load the wave's operands and clear the lane's `ok` where a value is missing; until every
// lane has everything the wave spins LIGHTLY -- on PEEK_OK, one or two values at the end of the newest tile(s), not on all of
// its operands (up to 60 loads per lane and round).
#define LP_LAST 1023                                // element (31, 31) of a k-grouped tile: the last one every producer here stores
#define LP_PEEK(tile) lp_set(lp_ld((tile) + LP_LAST))
#define LP_WAIT(sp, PEEK_OK, ...)                   \
  for (;;) {                                        \
    int ok = 1;                                     \
    __VA_ARGS__;                                    \
    if ((sp).dead || __all(ok)) break;              \
    do { if (!lp_again(sp, a)) break; } while (!(PEEK_OK)); \
  }

// One wave: Cholesky of the 32 x 32 tile in sdg (row-major, stride 33) and the inverse of its factor (csrc/chol32.h: how, and
// what it cost to get there).  Writes W = L^-1 = X^T k-grouped into sw (LDS, column groups LP_WGS apart) and gw (global: the
// helpers read it from there as soon as it lands).  bad: a pivot was not positive.
template <bool ONE>
static __device__ __forceinline__ void lp_diag(const double* sdg, double* sw, double* gw, double* sc, int lane, int* bad) {
  double a[32];
  const int row = lane & 31;
  const bool top = lane < 32;
#pragma unroll
  for (int c = 0; c < 32; ++c) a[c] = sdg[row * 33 + c];
  // (all 32 reads in flight, ONE wait: left to itself the compiler waits between groups of them -- 0.4 us per tile)
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int c = 0; c < 32; ++c) a[c] = top ? (c <= row ? a[c] : 0.0) : (c == row ? 1.0 : 0.0);
  double dmin;
  chol32_factor_split(a, dmin, sc);
  if (!(dmin > 0.0)) *bad = 1;
  if (!top) {
    // lane 32 + kk holds X[kk][c] = W[c][kk]
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      const int o = lp_kg(c, row);
      sw[(row >> 2) * LP_WGS + c * 4 + (row & 3)] = a[c];
      lp_gst<ONE>(gw + o, a[c]);
    }
  }
}

// acc (quadrant rb, cb) -= sum_{q = q0}^{q1 - 1} tileA(q) tileB(q)^T, tiles stepping by strideA / strideB doubles, each waited
// for as its turn comes; two terms' operands in flight
static __device__ __forceinline__ lp4d lp_accumulate(lp4d acc, const double* ta, size_t strideA, const double* tb, size_t strideB,
                                                     int q0, int q1, int rb, int cb, int lane, LpSpin& sp, const LpArgs& a) {
  int q = q0;
  for (; q + 1 < q1; q += 2) {
    double x0[8], y0[8], x1[8], y1[8];
    LP_WAIT(sp, LP_PEEK(ta + (size_t)(q + 1) * strideA) && LP_PEEK(tb + (size_t)(q + 1) * strideB),
      lp_rows<true>(ta + (size_t)q * strideA, rb, lane, x0);
      lp_rows<true>(tb + (size_t)q * strideB, cb, lane, y0);
      lp_rows<true>(ta + (size_t)(q + 1) * strideA, rb, lane, x1);
      lp_rows<true>(tb + (size_t)(q + 1) * strideB, cb, lane, y1);
      ok = lp_set8(x0) & lp_set8(y0) & lp_set8(x1) & lp_set8(y1));
    acc = lp_mma_neg(x0, y0, acc);
    acc = lp_mma_neg(x1, y1, acc);
  }
  if (q < q1) {
    double x0[8], y0[8];
    LP_WAIT(sp, LP_PEEK(ta + (size_t)q * strideA) && LP_PEEK(tb + (size_t)q * strideB),
      lp_rows<true>(ta + (size_t)q * strideA, rb, lane, x0);
      lp_rows<true>(tb + (size_t)q * strideB, cb, lane, y0);
      ok = lp_set8(x0) & lp_set8(y0));
    acc = lp_mma_neg(x0, y0, acc);
  }
  return acc;
}

template <bool ONE>
__global__ __launch_bounds__(256) void lrp_chol_kernel(LpArgs a) {
  __shared__ double s_dg[32 * 33];                  // the diagonal tile the chain factors next (row-major)
  __shared__ double s_dgp[LP_TILE];                 // ... its value before the last update (k-grouped)
  __shared__ double s_w[8 * LP_WGS];                // W_pp
  __shared__ double s_k2[1088];                     // the factoring wave's scratch (csrc/chol32.h)
  __shared__ double s_a1[LP_TILE];                  // chain: sub-diagonal tile before the multiplication by W_pp^T; helpers: the same role
  __shared__ double s_l1[2][LP_TILE];               // the chain's sub-diagonal tiles L_{p+1,p}, this step's and the last's
  __shared__ int s_ok, s_bad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = a.nt, H = a.H, D = a.D;
  const int rb = wave >> 1, cb = wave & 1;
  // ONE: the launch has 8 workgroups per participant and only those that land on XCD 0 take part (workgroup i of a launch goes
  // to XCD i % 8: checked once per device by lrp_one_xcd)
  if (ONE && (blockIdx.x & 7)) return;
  const int bid = ONE ? blockIdx.x >> 3 : blockIdx.x;
  LpSpin sp = {-1, 0, false};
  if (tid == 0) { s_ok = 1; s_bad = 0; }
  __syncthreads();

  if (bid == 0) {
    // ================= the chain =================
    {
      const lp4d c = lp_quad_load_p(a.TT, nt, a.ks, 0, rb, cb, lane);   // tile (0, 0), written by the kernel before
      const int col = cb * 16 + (lane & 15), row0 = rb * 16 + (lane >> 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) s_dg[(row0 + 4 * r) * 33 + col] = c[r];
    }
    __syncthreads();
    // after the diagonal tile and the side work of step p (barrier): L_{p+1,p} = (tile) W_pp^T, barrier, the next diagonal
    // tile's last update, barrier (hands s_dg to wave 0)
    auto finish_step = [&](int p) {
      double* l1 = s_l1[p & 1];
      {
        double x[8], y[8];
        lp_rows<false>(s_a1, rb, lane, x);
        lp_rows<false, LP_WGS>(s_w, cb, lane, y);
        const lp4d c = lp_mma(x, y, (lp4d){0.0, 0.0, 0.0, 0.0});
        lp_quad_store<false>(l1, rb, cb, lane, c);
        lp_quad_store<true, ONE>(a.LT + (size_t)((p + 1) * nt + p) * LP_TILE, rb, cb, lane, c);
      }
      if (wave == 0) LP_STAMP(2);
      __syncthreads();
      {
        double x[8], y[8];
        lp_rows<false>(l1, rb, lane, x);
        lp_rows<false>(l1, cb, lane, y);
        const lp4d c = lp_mma_neg(x, y, lp_quad_load<false>(s_dgp, rb, cb, lane));
        const int col = cb * 16 + (lane & 15), row0 = rb * 16 + (lane >> 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) s_dg[(row0 + 4 * r) * 33 + col] = c[r];
      }
      __syncthreads();
    };
    // Two loops with the same barriers, one per role, so that the factoring wave's carries nothing of the other role's state
    // (in one loop the register allocator parked ~100 values of the side work in AGPRs around the factorisation and copied
    // them back at the join, on the critical path of every step)
    if (wave == 0) {
      for (int p = 0; p < nt; ++p) {
        LP_STAMP(0);
        lp_diag<ONE>(s_dg, s_w, a.XW + (size_t)p * LP_TILE, s_k2, lane, &s_bad);      // (W_pp is on its way to the helpers: nothing to wait for)
        LP_STAMP(1);
        __syncthreads();
        if (p + 1 >= nt || !s_ok) break;
        finish_step(p);
        LP_STAMP(5);
      }
    } else {
      for (int p = 0; p < nt; ++p) {
        const bool more = p + 1 < nt;
        if (more) {
          // while wave 0 factors: what does not depend on this diagonal tile -- tiles (p+1, p) and (p+1, p+1) up to and
          // including the term of column p - 1.  Eight quadrant jobs over three waves (<= 3 each, their operands requested together).
          const double* t_sub = a.AS + (size_t)p * LP_TILE;
          const double* t_dg = a.AD + (size_t)p * LP_TILE;
          const double* l_far = p >= 1 ? a.LT + (size_t)((p + 1) * nt + p - 1) * LP_TILE : nullptr;      // L_{p+1,p-1}: a helper's
          const double* l_prev = s_l1[(p + 1) & 1];                                                        // L_{p,p-1}: the chain's last
          lp4d c[3];
          double x[3][8], y[3][8];
          if (p == 0) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {
              const int job = wave - 1 + 3 * u;
              if (job < 8) c[u] = lp_quad_load_p(a.TT, nt, a.ks, job < 4 ? nt : nt + 1, (job & 3) >> 1, job & 1, lane);      // (tiles (1, 0) and (1, 1) of P)
            }
          } else {
            LP_WAIT(sp, LP_PEEK(l_far) && LP_PEEK(t_sub) && LP_PEEK(t_dg),
              _Pragma("unroll")
              for (int u = 0; u < 3; ++u) {
                const int job = wave - 1 + 3 * u;
                if (job < 8) {
                  const int q = job & 3, qr = q >> 1, qc = q & 1;
                  c[u] = lp_quad_load<true>(job < 4 ? t_sub : t_dg, qr, qc, lane);
                  lp_rows<true>(l_far, qr, lane, x[u]);
                  ok &= lp_set4(c[u]) & lp_set8(x[u]);
                  if (job < 4) lp_rows<false>(l_prev, qc, lane, y[u]);
                  else { lp_rows<true>(l_far, qc, lane, y[u]); ok &= lp_set8(y[u]); }
                }
              });
          }
          if (wave == 1) LP_STAMP(3);
#pragma unroll
          for (int u = 0; u < 3; ++u) {
            const int job = wave - 1 + 3 * u;
            if (job < 8) {
              const int q = job & 3, qr = q >> 1, qc = q & 1;
              if (p >= 1) c[u] = lp_mma_neg(x[u], y[u], c[u]);
              lp_quad_store<false>(job < 4 ? s_a1 : s_dgp, qr, qc, lane, c[u]);
            }
          }
          if (sp.dead && lane == 0) s_ok = 0;
          if (wave == 1) LP_STAMP(4);
        }
        __syncthreads();
        if (!more || !s_ok) break;
        finish_step(p);
      }
    }
    __syncthreads();
    if (tid == 0 && s_bad) atomicExch(a.flags + 4 * nt, 2);
    return;
  }

  if (bid < LP_FIXED_WGS) {
    // ================= an assistant: the chain's tile of step p, all terms of the columns q <= p - 2 =================
    const bool sub = bid == 1;                      // 1: tile (p+1, p); 2: tile (p+1, p+1)
    for (int p = 1; p + 1 < nt; ++p) {
      if (wave == 0) LP_STAMP(0);
      lp4d c = lp_quad_load_p(a.TT, nt, a.ks, (p + 1) * nt + (sub ? p : p + 1), rb, cb, lane);
      const double* ta = a.LT + (size_t)((p + 1) * nt) * LP_TILE;                     // row p + 1 of L
      const double* tb = a.LT + (size_t)((sub ? p : p + 1) * nt) * LP_TILE;           // row p (or p + 1 again)
      c = lp_accumulate(c, ta, LP_TILE, tb, LP_TILE, 0, p - 1, rb, cb, lane, sp, a);
      lp_quad_store<true, ONE>((sub ? a.AS : a.AD) + (size_t)p * LP_TILE, rb, cb, lane, c);
      if (wave == 0) LP_STAMP(1);
    }
    return;
  }

  // ================= a helper =================
  const int h = bid - LP_FIXED_WGS;
  for (int p = 0; p < nt; ++p) {
    // the jobs of block column p, in a fixed order: top rows p + 2 .. nt - 1, bottom rows 0 .. p, the right-hand side row;
    // job n goes to helper (n + 3 p) mod H
    const int ntopj = nt - p - 2 > 0 ? nt - p - 2 : 0;
    const int njobs = ntopj + p + 2;
    int done = 0;
    for (int n = (h + H - (3 * p) % H) % H; n < njobs; n += H) {
      if (wave == 0 && done == 0) LP_STAMP(0);
      const bool is_top = n < ntopj;
      const int R = is_top ? p + 2 + n : n - ntopj;                     // top row i, or bottom row r (r == p + 1 here means: the rhs row)
      const bool is_rhs = !is_top && R == p + 1;
      const int rrow = is_rhs ? nt : R;
      // init - sum_{q < p} tile(R, q) L_{p,q}^T
      lp4d c;
      int q0 = 0;
      const double* ta;
      if (is_top) {
        c = lp_quad_load_p(a.TT, nt, a.ks, R * nt + p, rb, cb, lane);
        ta = a.LT + (size_t)(R * nt) * LP_TILE;
      } else {
        const int col = cb * 16 + (lane & 15), row0 = rb * 16 + (lane >> 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = row0 + 4 * r;
          c[r] = is_rhs ? (row == 0 ? a.rhs[p * 32 + col] : 0.0) : ((R == p && row == col) ? 1.0 : 0.0);
        }
        ta = a.BL + (size_t)(rrow * nt) * LP_TILE;
        q0 = is_rhs ? 0 : R;                        // (row r of L^-T starts at column r)
      }
      c = lp_accumulate(c, ta, LP_TILE, a.LT + (size_t)(p * nt) * LP_TILE, LP_TILE, q0, p, rb, cb, lane, sp, a);
      __syncthreads();                              // (the previous job's reads of s_a1)
      lp_quad_store<false>(s_a1, rb, cb, lane, c);
      __syncthreads();
      double x[8], wq[8];
      lp_rows<false>(s_a1, rb, lane, x);
      LP_WAIT(sp, LP_PEEK(a.XW + (size_t)p * LP_TILE), lp_rows<true>(a.XW + (size_t)p * LP_TILE, cb, lane, wq); ok = lp_set8(wq));
      if (wave == 0 && done == 0) LP_STAMP(1);
      c = lp_mma(x, wq, (lp4d){0.0, 0.0, 0.0, 0.0});
      if (is_top) lp_quad_store<true, ONE>(a.LT + (size_t)(R * nt + p) * LP_TILE, rb, cb, lane, c);
      else {
        lp_quad_store<true, ONE>(a.BL + (size_t)(rrow * nt + p) * LP_TILE, rb, cb, lane, c);
        if (is_rhs) {
          if (rb == 0 && (lane >> 4) == 0) { const int oc = p * 32 + cb * 16 + (lane & 15); if (oc < D) a.uvec[oc] = c[0]; }      // row 0: u
        } else {
          // finished tile (r, p) of U = L^-T, row-major for the draw kernel
          const int ocol = p * 32 + cb * 16 + (lane & 15);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int orow = R * 32 + rb * 16 + (lane >> 4) + 4 * q;
            if (orow < D && ocol < D) a.U[(size_t)orow * a.ldu + ocol] = c[q];
          }
        }
      }
      ++done;
    }
    if (done && wave == 0) LP_STAMP(2);
  }
}

// mu = U u, u = L^-1 rhs: a launch of its own, and only for a caller that asks for the mean -- the draws do not need it
// (theta = mu + U r = U (u + r): lrp_draw_kernel adds u to the normal numbers).  32 rows of U per workgroup, a wave per row,
// lanes along the row.
__global__ __launch_bounds__(256) void lrp_mean_kernel(LpArgs a) {
  __shared__ double su[LP_NB * LP_MAX_NT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nt = a.nt, D = a.D;
  for (int i = tid; i < nt * 32; i += 256) su[i] = i < D ? a.uvec[i] : 0.0;
  __syncthreads();
  // eight rows per wave; every row's pieces are requested before any is used (D <= 1024: at most 16 per lane and row)
  double s[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = blockIdx.x * 32 + wave + 4 * u;
    const double* row = a.U + (size_t)(c < D ? c : 0) * a.ldu;
    const int base = (c & ~63) + lane;
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int t0 = 0; base + 64 * t0 < D; t0 += 4) {
      double x[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i = base + 64 * (t0 + q); x[q] = (c < D && i < D && i >= c) ? row[i] : 0.0; }
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int i = base + 64 * (t0 + q); v[q] = fma(x[q], i < D ? su[i] : 0.0, v[q]); }
    }
    s[u] = (v[0] + v[1]) + (v[2] + v[3]);
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int c = blockIdx.x * 32 + wave + 4 * u;
    const double m = wave_allsum(s[u]);
    if (lane == 0 && c < D) a.mu[c] = m;
  }
}

// ---- the draws: theta = mu + [R; Rbar] U^T = ([R; Rbar] + 1 u^T) U^T ---------------------------------------------------------
// (examples/linear_regression/main.py:147: muw + randn(n, D).dot(USigw.T), with muw = U u for u = L^-1 rhs: the mean needs no
// product of its own, u is added to every row of the normal numbers as they are read.)  32 x 32 blocks of the product, one
// 16 x 16 tile per wave, both operands read along the contraction index as in lrp_form_kernel; U is upper triangular, so the
// block of columns c0 .. c0 + 31 of theta starts its inner index at c0.  Row S of the left operand is Rbar, the column means
// of R: its "draw" is the mean of the draws (what the closed-form column sums are expanded around, csrc/moments.hip).
struct LpDrawArgs {
  const double* U; const double* u; const double* R; const double* Rbar;
  double* theta; double* tbar;
  int64_t ldu;
  int D, S, ld;
};
__global__ __launch_bounds__(256) void lrp_draw_kernel(LpDrawArgs a) {
  // A 16 x 16 tile of theta per workgroup, its four waves sharing the inner index: wave w takes the runs of 32 numbered
  // w, w + 4, w + 8 (for D <= 384 that is ALL of a wave's loads in flight at once -- one round trip to memory, not four: the
  // kernel is nothing but that latency), the partial tiles meet in LDS in wave order.
  __shared__ double su[LP_NB * LP_MAX_NT + 32];
  __shared__ double red[4][4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int D = a.D, S = a.S;
  const int arow = blockIdx.x * 16 + li;            // row of [R; Rbar]
  const int bcol = blockIdx.y * 16 + li;            // column of theta = row of U
  const double* rp = (arow < S ? a.R + (size_t)arow * a.ld : a.Rbar) + 8 * lk;
  const double* up = a.U + (size_t)(bcol < D ? bcol : 0) * a.ldu + 8 * lk;
  const bool aok = arow <= S, bok = bcol < D;
  const int k0 = (blockIdx.y * 16) & ~31;           // (U[c][i] = 0 for i < c)
  auto fetch = [&](int kb, double (&xa)[8], double (&xb)[8]) {
    const int kk = kb + 8 * lk;
    if (kk + 8 <= D) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const lp2d u = *(const lp2d*)(rp + kb + 2 * q), v = *(const lp2d*)(up + kb + 2 * q);
        xa[2 * q] = u.x; xa[2 * q + 1] = u.y; xb[2 * q] = v.x; xb[2 * q + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = kk + q < D;
        xa[q] = ok ? rp[kb + (ok ? q : 0)] : 0.0;
        xb[q] = ok ? up[kb + (ok ? q : 0)] : 0.0;
      }
    }
    if (!aok) {
#pragma unroll
      for (int q = 0; q < 8; ++q) xa[q] = 0.0;
    }
    if (!bok) {
#pragma unroll
      for (int q = 0; q < 8; ++q) xb[q] = 0.0;
    }
  };
  lp4d acc = (lp4d){0.0, 0.0, 0.0, 0.0};
  double xa[3][8], xb[3][8];
  const int first = k0 + 32 * wave;
  // (the loads do not need u: they go out before the copy of u into LDS is waited for)
#pragma unroll
  for (int u = 0; u < 3; ++u) if (first + 128 * u < D) fetch(first + 128 * u, xa[u], xb[u]);
  for (int i = tid; i < ((D + 31) & ~31); i += 256) su[i] = i < D ? a.u[i] : 0.0;
  __syncthreads();
  for (int kb = first; kb < D; kb += 384) {
    if (kb != first) {
#pragma unroll
      for (int u = 0; u < 3; ++u) if (kb + 128 * u < D) fetch(kb + 128 * u, xa[u], xb[u]);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int kc = kb + 128 * u;
      if (kc < D) {
        if (aok) {                                  // r + u (rows that exist; the pad of su is zero)
#pragma unroll
          for (int q = 0; q < 8; ++q) xa[u][q] += su[kc + 8 * lk + q];
        }
        acc = lp_mma(xa[u], xb[u], acc);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
  __syncthreads();
  if (wave == 0) {
    const int col = blockIdx.y * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = blockIdx.x * 16 + lk + 4 * r;
      const double t = ((red[0][r][lane] + red[1][r][lane]) + red[2][r][lane]) + red[3][r][lane];
      const double v = col < D ? t : 0.0;
      if (row < S && col < a.ld) a.theta[(size_t)row * a.ld + col] = v;
      else if (row == S && col < D) a.tbar[col] = v;
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------------------
void bcx_project_set_error(const std::string& msg);   // proj.hip
#define LRP_HIP(call)                                                             \
  do {                                                                            \
    hipError_t _e = (call);                                                       \
    if (_e != hipSuccess) {                                                       \
      bcx_project_set_error(std::string(#call) + ": " + hipGetErrorString(_e));   \
      return BCX_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

static int lrp_helpers(int nt) {
  static const int forced = [] { const char* e = bcx_dev_env("BCX_LRP_HELPERS"); return e ? atoi(e) : 0; }();      // dev
  if (forced >= 1 && forced <= LP_MAX_H) return forced;
  int h = nt + 2;                                   // a block column has at most nt + 1 jobs for the helpers: one each
  if (h > LP_MAX_H) h = LP_MAX_H;
  return h;
}
// Does workgroup i of a launch run on XCD i % 8 on this device?  (Then lrp_chol_kernel keeps all of its workgroups on one XCD
// and hands tiles over with ordinary stores.)  Checked once per device with a launch that records every workgroup's XCC_ID.
__global__ void lrp_xcc_probe_kernel(int* out) {
  if (threadIdx.x == 0) { int id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id)); out[blockIdx.x] = id & 0xf; }
}
static bool lrp_one_xcd(hipStream_t st) {
  static std::atomic<int> known[64];                // 0 unknown, 1 yes, 2 no
  // BCX_LRP_SPREAD=1 (a runtime option, not a dev switch): keep the workgroups spread over the chip.  One XCD has room for
  // two such launches at a time; more than two PROCESSES factoring on one GPU at once could each hold part of the XCD and wait
  // for the rest until the time-out (status BCX_ERR_TIMEOUT, never a wrong result).  One process per GPU -- the deployment
  // this library is for -- cannot get there: a stream's launches run one after the other.
  static const bool off = [] { const char* e = getenv("BCX_LRP_SPREAD"); return e && e[0] == '1'; }();
  if (off) return false;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  int k = known[dev].load();
  if (k == 0) {
    const int n = 8 * 40;
    int* d = nullptr;
    int h[8 * 40];
    bool ok = hipMalloc(&d, n * sizeof(int)) == hipSuccess;
    if (ok) {
      hipLaunchKernelGGL(lrp_xcc_probe_kernel, dim3(n), dim3(64), 0, st, d);
      ok = hipMemcpyAsync(h, d, n * sizeof(int), hipMemcpyDeviceToHost, st) == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
      for (int i = 0; ok && i < n; ++i) ok = (i % 8 == 0) == (h[i] == h[0]);
      (void)hipFree(d);
    }
    k = ok ? 1 : 2;
    known[dev].store(k);
  }
  return k == 1;
}
#define LP_MAX_KS 4
static int lrp_slices(int k) {                      // workgroups per tile of P: about three runs of 32 points each
  static const int forced = [] { const char* e = bcx_dev_env("BCX_LRP_KS"); return e ? atoi(e) : 0; }();      // dev
  if (forced >= 1 && forced <= LP_MAX_KS) return forced;
  const int runs = (k + 31) / 32;
  return runs >= 12 ? 4 : runs >= 8 ? 3 : runs >= 4 ? 2 : 1;
}
static int64_t lrp_tiles(int nt) { return (int64_t)(1 + LP_MAX_KS) * nt * nt + (int64_t)(nt + 1) * nt + 3 * nt; }
static int64_t lrp_flag_bytes(int nt) { return (int64_t)(4 * nt + 1 + 15) / 16 * 16 * 4; }
#define LRP_DBG_BYTES(nt) ((int64_t)(LP_FIXED_WGS + LP_MAX_H) * (nt) * 8 * 8)

extern "C" int64_t bcx_linreg_posterior_factor_scratch_bytes(int32_t D) {
  if (D < 1 || D > LP_NB * LP_MAX_NT) return -1;
  const int nt = (D + LP_NB - 1) / LP_NB;
  return lrp_tiles(nt) * LP_TILE * 8 + (int64_t)nt * 32 * 8 + lrp_flag_bytes(nt) + LRP_DBG_BYTES(nt);
}

extern "C" int bcx_linreg_posterior_factor(void* stream, int32_t k, int32_t D, int32_t ldx, const void* w_dev, const void* XT_dev,
                                           const void* y_dev, const void* S0inv_dev, int32_t lds0, const void* rhs0_dev, double sigsq,
                                           void* work_dev, int64_t work_bytes, void* U_dev, int64_t ldu, void* u_dev, void* mu_dev) {
  if (k < 0 || k > 4096 || D < 1 || D > LP_NB * LP_MAX_NT || ldx < (k + 31) / 32 * 32 || lds0 < D || ldu < D || (ldu & 1) || !(sigsq > 0.0) ||
      !S0inv_dev || !rhs0_dev || !work_dev || !U_dev || !u_dev || (k > 0 && (!w_dev || !XT_dev || !y_dev)) ||
      work_bytes < bcx_linreg_posterior_factor_scratch_bytes(D) || (((uintptr_t)work_dev | (uintptr_t)XT_dev | (uintptr_t)U_dev) & 15)) {
    bcx_project_set_error("bcx_linreg_posterior_factor: bad arguments (k <= 4096 points, D <= 1024 features, the features by points "
                          "with a row stride of k rounded up to 32, zero padded, 16-byte aligned; scratch of "
                          "bcx_linreg_posterior_factor_scratch_bytes(D) bytes)");
    return BCX_ERR_ARG;
  }
  LpArgs a;
  a.w = (const double*)w_dev; a.XT = (const double*)XT_dev; a.y = (const double*)y_dev;
  a.S0inv = (const double*)S0inv_dev; a.rhs0 = (const double*)rhs0_dev;
  a.sigsq = sigsq; a.k = k; a.D = D; a.ldx = ldx; a.lds0 = lds0;
  const int nt = (D + LP_NB - 1) / LP_NB;
  double* base = (double*)work_dev;
  a.TT = base; base += (size_t)LP_MAX_KS * nt * nt * LP_TILE;
  a.LT = base; base += (size_t)nt * nt * LP_TILE;
  a.BL = base; base += (size_t)(nt + 1) * nt * LP_TILE;
  a.XW = base; base += (size_t)nt * LP_TILE;
  a.AS = base; base += (size_t)nt * LP_TILE;
  a.AD = base; base += (size_t)nt * LP_TILE;
  a.rhs = base; base += (size_t)nt * 32;
  a.flags = (int*)base;
  static const bool dbg = bcx_dev_env("BCX_LRP_DBG") != nullptr;
  a.dbg = dbg ? (long long*)((char*)base + lrp_flag_bytes(nt)) : nullptr;
  a.U = (double*)U_dev; a.ldu = ldu; a.uvec = (double*)u_dev; a.mu = (double*)mu_dev;
  a.nt = nt; a.H = lrp_helpers(nt); a.ks = lrp_slices(k);
  a.timeout_ticks = 200000000LL;                    // 2 s of the 100 MHz wall clock
  hipLaunchKernelGGL(lrp_form_kernel, dim3(a.ks * nt * (nt + 1) / 2 + nt), dim3(256), 0, (hipStream_t)stream, a);
  // (one XCD has 32 CUs and the kernel's registers allow one workgroup per CU: beyond 28 participants the launch spreads out)
  if (LP_FIXED_WGS + a.H <= 28 && lrp_one_xcd((hipStream_t)stream)) hipLaunchKernelGGL(lrp_chol_kernel<true>, dim3(8 * (LP_FIXED_WGS + a.H)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(lrp_chol_kernel<false>, dim3(LP_FIXED_WGS + a.H), dim3(256), 0, (hipStream_t)stream, a);
  if (a.mu) hipLaunchKernelGGL(lrp_mean_kernel, dim3(nt), dim3(256), 0, (hipStream_t)stream, a);
  LRP_HIP(hipGetLastError());
  return BCX_OK;
}

// After a synchronisation of the stream: 0 ok, BCX_ERR_TIMEOUT a wait between the workgroups of lrp_chol_kernel expired (its
// outputs are not valid), BCX_ERR_STATE a pivot was not positive (P not positive definite: NaN / negative weights upstream).
extern "C" int bcx_linreg_posterior_factor_status(void* stream, int32_t D, const void* work_dev) {
  if (D < 1 || D > LP_NB * LP_MAX_NT || !work_dev) { bcx_project_set_error("bcx_linreg_posterior_factor_status: bad arguments"); return BCX_ERR_ARG; }
  const int nt = (D + LP_NB - 1) / LP_NB;
  const int* flags = (const int*)((const double*)work_dev + lrp_tiles(nt) * LP_TILE + (int64_t)nt * 32);
  int v = 0;
  LRP_HIP(hipMemcpyAsync(&v, flags + 4 * nt, sizeof v, hipMemcpyDeviceToHost, (hipStream_t)stream));
  LRP_HIP(hipStreamSynchronize((hipStream_t)stream));
  if (v == 1) { bcx_project_set_error("linreg posterior factorisation: a wait between workgroups timed out (GPU shared or preempted)"); return BCX_ERR_TIMEOUT; }
  if (v == 2) { bcx_project_set_error("linreg posterior factorisation: the precision matrix is not positive definite"); return BCX_ERR_STATE; }
  return BCX_OK;
}

// theta_dev (S x ld) = mu + R U^T, tbar_dev (D) = mu + Rbar U^T for the factor U_dev (D x ldu, upper triangular) and the vector u_dev
// (mu = U u) of bcx_linreg_posterior_factor: the reference's `muw + np.random.randn(n, D).dot(USigw.T)` with R in the place of randn.
extern "C" int bcx_linreg_posterior_draw_factored(void* stream, int32_t D, int32_t ld, const void* U_dev, int64_t ldu, const void* u_dev,
                                                  const void* R_dev, const void* Rbar_dev, int32_t S, void* theta_dev, void* tbar_dev) {
  if (D < 1 || D > LP_NB * LP_MAX_NT || ld < D || (ld & 1) || ldu < D || (ldu & 1) || S < 1 || S > (1 << 22) || !U_dev || !u_dev || !R_dev ||
      !Rbar_dev || !theta_dev || !tbar_dev || (((uintptr_t)U_dev | (uintptr_t)R_dev | (uintptr_t)Rbar_dev) & 15)) {
    bcx_project_set_error("bcx_linreg_posterior_draw_factored: bad arguments (even leading dimensions, 16-byte aligned rows)");
    return BCX_ERR_ARG;
  }
  LpDrawArgs a;
  a.U = (const double*)U_dev; a.u = (const double*)u_dev; a.R = (const double*)R_dev; a.Rbar = (const double*)Rbar_dev;
  a.theta = (double*)theta_dev; a.tbar = (double*)tbar_dev; a.ldu = ldu; a.D = D; a.S = S; a.ld = ld;
  hipLaunchKernelGGL(lrp_draw_kernel, dim3((S + 1 + 15) / 16, (ld + 15) / 16), dim3(256), 0, (hipStream_t)stream, a);
  LRP_HIP(hipGetLastError());
  return BCX_OK;
}
