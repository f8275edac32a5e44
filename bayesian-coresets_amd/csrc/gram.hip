// gram.hip -- G = V V^T over the k active rows of the dense re-weight (optimize(), snnls.py:82-97; OMP's nnls call,
// orthopursuit.py:37-42) on the fp64 matrix cores, balanced over the whole chip ("stream-K").
//
// Why a second Gram kernel (the first one, gram_tile_kernel in moments.hip, stays as the path for rows that are not
// 16-byte aligned and for small supports): that kernel hands out whole 64 x 64 blocks of the upper triangle (at k = 4096:
// 528 x 4 block pairs for 512 resident workgroups -- a ragged last round), stages both operands through registers with two
// workgroup barriers per 32 values, gives a wave 32 x 32 entries (one operand read per MFMA) and stores the mirror image as
// 8-byte entries 32 KiB apart: 0.34 / 0.51 of the fp64 MFMA peak at k = 1497 / 4096 (d = 1024).  Here:
//   * 128 x 128 workgroup tiles (128 x 64 below k = 3072), a wave multiplies 64 rows by 64 (32) columns with
//     v_mfma_f64_16x16x4_f64: 8 (6) ds_read_b128 feed 32 (16) MFMAs per 8 values of the row length.  Operands come by
//     LDS-DMA (global_load_lds_dwordx4, no staging registers) one stage of 16 values ahead into two-slot rings (two ahead,
//     three slots, for the narrow tile), one wait + one bare barrier per stage; 128-byte LDS lines rotated so that every
//     ds_read_b128 is conflict free (SQ_LDS_BANK_CONFLICT = 0, layout of csrc/proj.hip);
//   * the work is the SEQUENCE of (tile, stage) units of the upper triangle, cut into equal contiguous ranges, one per
//     resident workgroup: every workgroup multiplies for the same time whatever k is.  A tile whose stages span several
//     workgroups is finished by the one that holds its last stage: the others leave their partial accumulators in
//     scratch (register layout, 16-byte pieces, coalesced) and raise a flag; the finisher adds them in a fixed order (its own
//     part first, then the contributors from the nearest to the farthest), so the result is deterministic for a given
//     (k, d, workgroup count).  Every workgroup takes the piece its successors finish FIRST and the piece its predecessors
//     began LAST, so nobody waits for a workgroup that has not started yet;
//   * the tile sequence is cut into eight contiguous pieces, one per XCD (workgroups b, b + 8, ... share an XCD under
//     round-robin dispatch: a speed assumption, not a correctness one), so the workgroups of an XCD walk neighbouring tiles
//     of the same block rows and find one of the two row panels in their L2; a workgroup only ever waits for workgroups
//     with a LOWER blockIdx (b - 8, b - 16, ...), which were dispatched before it;
//   * both triangles are written, every entry pair from ONE accumulator (the one on or above the diagonal): G is symmetric
//     bit for bit; the mirror image leaves the wave transposed in registers, as 16-byte pieces that fill 128-byte lines.
// Measured (tools/gram_bench.py, d = 1024): k = 4096 306 us = 0.72 of the peak on the triangle's k (k + 1) d flops (the
// steady state of the stage loop is 0.78; launch, prologue, partial tiles and stores are ~28 us), k = 8192 0.76,
// k = 1497 81 us = 0.36 (19.5 stages per workgroup: the fixed costs dominate).  An earlier version of this kernel on the
// projection kernel's v_mfma_f64_4x4x4_4b_f64 loop (one operand replicated: 4.5 times the LDS reads per flop) reached 0.62.
#include <algorithm>
#include <atomic>
#include <chrono>
#include "bcx_internal.h"
#include "dev_util.h"

typedef double gk4d __attribute__((ext_vector_type(4)));
typedef double gk2d __attribute__((ext_vector_type(2)));

#define GK_KC 16                      // values of the row length per stage
#define GK_ROWS 128                   // rows of G per tile: 2 x 2 waves, 64 rows x (16 NTB) columns each
#define GK_IBYTES (GK_ROWS * GK_KC * 8)
// NTB = 16-column tiles per wave: 4 (a 128 x 128 workgroup tile; rows requested one stage ahead into two-slot rings: 64 KiB,
// two workgroups per CU) or 2 (128 x 64 for smaller supports -- more tiles to balance, half the partial tile; two stages ahead
// into three-slot rings: 72 KiB)
#define GK_COLS(NTB) (32 * (NTB))
#define GK_JBYTES(NTB) (GK_COLS(NTB) * GK_KC * 8)
#define GK_RING(NTB) ((NTB) == 4 ? 2 : 3)
#define GK_JBASE(NTB) (GK_RING(NTB) * GK_IBYTES)
#define GK_LDS_BYTES(NTB) (GK_JBASE(NTB) + GK_RING(NTB) * GK_JBYTES(NTB))
#define GK_MIN_STAGES 6               // a workgroup is not started for fewer stages than this (prologue + fix-up cost)

struct GramSkArgs {
  const double* V;        // k rows of d doubles, row stride ld (even, base 16-byte aligned)
  double* G;              // k x ldg
  double* part;           // gridDim.x x (128 x COLS) partial tiles, register layout
  unsigned long long* flags;   // gridDim.x: epoch when that workgroup's partial tile is complete
  unsigned long long epoch;
  unsigned long long* status;   // == epoch: a wait of THIS call expired (its result is not valid)
  int64_t ld, ldg;
  long long timeout_ticks;
  int k, d, nI, nJ, ntiles, nst;
  int dbg;                // dev (BCX_GRAM_DBG): 1 = no row requests after the first stage, 2 = no partial tiles / fix-up (timing experiments: wrong results)
};

// One LDS-DMA request: 64 lanes x 16 bytes, each lane's own global address -> lds_dst + 16 * lane (csrc/proj.hip pj_glds16:
// inline asm, because with the builtin in flight hipcc turns every counted lgkmcnt wait into lgkmcnt(0)).
static __device__ __forceinline__ void gk_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Partial tiles cross workgroups -- possibly XCDs, whose L2s are not coherent with each other for ordinary accesses: 16-byte
// stores / loads at device scope (sc1: written through to, read from, the memory side).  The compiler does not count these
// (inline asm): the callers wait with s_waitcnt vmcnt(0) themselves.
static __device__ __forceinline__ void gk_st2(double* p, gk2d v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
static __device__ __forceinline__ gk2d gk_ld2(const double* p) {
  gk2d v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}

// Exchanges between the four 16-lane rows of a wave (lane >> 4 = 0 .. 3), on doubles:
//   gk_swap16(a, b): rows (r0 r1 r2 r3) of a and b -> a = (a.r0, b.r0, a.r2, b.r2), b = (a.r1, b.r1, a.r3, b.r3)
//   gk_swap32(a, b):                                 a = (a.r0, a.r1, b.r0, b.r1), b = (a.r2, a.r3, b.r2, b.r3)
// (v_permlane16_swap / v_permlane32_swap on the two halves of the value; element reads through named scalars, see scan.hip)
typedef unsigned gk_v2u __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ void gk_swap16(double& a, double& b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const gk_v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
  const gk_v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  const unsigned al = lo.x, bl = lo.y, ah = hi.x, bh = hi.y;
  a = __longlong_as_double((long long)(((unsigned long long)ah << 32) | al));
  b = __longlong_as_double((long long)(((unsigned long long)bh << 32) | bl));
}
static __device__ __forceinline__ void gk_swap32(double& a, double& b) {
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const gk_v2u lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
  const gk_v2u hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  const unsigned al = lo.x, bl = lo.y, ah = hi.x, bh = hi.y;
  a = __longlong_as_double((long long)(((unsigned long long)ah << 32) | al));
  b = __longlong_as_double((long long)(((unsigned long long)bh << 32) | bl));
}

struct GkPos { int tile, s, I, J; };

// first column block of block row I that holds an entry on or above the diagonal
template <int NTB> static __host__ __device__ __forceinline__ int gk_jmin(int I) { return (I * GK_ROWS) / GK_COLS(NTB); }

template <int NTB>
__global__ __launch_bounds__(256, 2) void gram_sk_kernel(GramSkArgs p) {
  constexpr int COLS = GK_COLS(NTB), JBYTES = GK_JBYTES(NTB), JBASE = GK_JBASE(NTB), RING = GK_RING(NTB), AHEAD = RING - 1;
  constexpr int TCH = NTB, NREQ = 4 + TCH;     // LDS-DMA requests per wave and stage: 4 chunks of the row block, TCH of the column block
  extern __shared__ __attribute__((aligned(16))) unsigned char gk_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int nst = p.nst;
  // ---- this workgroup's range of (tile, stage) units: XCD piece of the tile sequence, then an equal share of its units ----
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const int t_lo = (int)((int64_t)xcd * p.ntiles / 8), t_hi = (int)((int64_t)(xcd + 1) * p.ntiles / 8);
  const int64_t units = (int64_t)(t_hi - t_lo) * nst;
  const int64_t u0 = units * slot / per, u1 = units * (slot + 1) / per;       // relative to the piece's first unit
  if (u0 >= u1) return;
  auto locate = [&](int64_t v) {             // unit -> (tile, stage, I, J): block rows in order, columns jmin(I) .. nJ - 1
    GkPos a;
    a.tile = t_lo + (int)(v / nst); a.s = (int)(v % nst);
    int t = a.tile, I = 0;
    while (t >= p.nJ - gk_jmin<NTB>(I)) { t -= p.nJ - gk_jmin<NTB>(I); ++I; }
    a.I = I; a.J = gk_jmin<NTB>(I) + t;
    return a;
  };
  auto advance = [&](GkPos a) {
    if (++a.s == nst) {
      a.s = 0; a.tile += 1;
      if (++a.J == p.nJ) { a.I += 1; a.J = gk_jmin<NTB>(a.I); }
    }
    return a;
  };
  // Order of work.  A tile whose stages are split over workgroups is finished by the holder of its LAST stage, which needs
  // the others' partial sums; a workgroup's range is [tail of a tile begun by its predecessors | whole tiles | head of a tile
  // its successors finish].  It works through them in the order: head of the next tile FIRST (segment 0: published at once,
  // nobody waits long for it), then the whole tiles (segment 1), and the piece that needs its predecessors' partial sums
  // LAST (segment 2) -- by then they were published long ago.  (In range order every workgroup would wait for its
  // predecessor to finish: a chain through the XCD.)  Empty segments have lo == hi.
  int64_t lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;
  {
    const int64_t first_end = (u0 / nst + 1) * nst;
    if (first_end >= u1) {                    // the whole range lies inside one tile
      if (u1 == first_end) { lo2 = u0; hi2 = u1; } else { lo0 = u0; hi0 = u1; }
    } else {
      const int64_t tail_start = (u1 / nst) * nst, head_end = (u0 % nst) ? first_end : u0;
      lo0 = tail_start; hi0 = u1;
      lo1 = head_end; hi1 = tail_start;
      lo2 = u0; hi2 = head_end;
    }
  }
  struct It { int64_t u; int seg; GkPos g; };      // seg 3: past the end
  auto seg_start = [&](int sg) {              // first unit of the first non-empty segment >= sg
    It it;
    it.seg = sg;
    if (it.seg == 0 && lo0 == hi0) it.seg = 1;
    if (it.seg == 1 && lo1 == hi1) it.seg = 2;
    if (it.seg == 2 && lo2 == hi2) it.seg = 3;
    it.u = it.seg == 0 ? lo0 : (it.seg == 1 ? lo1 : lo2);
    if (it.seg < 3) it.g = locate(it.u); else it.g = GkPos{0, 0, 0, 0};
    return it;
  };
  auto succ = [&](const It& a) {
    const int64_t hi = a.seg == 0 ? hi0 : (a.seg == 1 ? hi1 : hi2);
    if (a.u + 1 < hi) { It b; b.u = a.u + 1; b.seg = a.seg; b.g = advance(a.g); return b; }
    return seg_start(a.seg + 1);
  };
  // ---- the prefetch stream: this lane's share of a stage -- slot `lane` of I chunks 4 wave + j (rows 32 wave + 8 j + fr) and
  // of J chunks TCH wave + j (rows 8 (TCH wave + j) + fr of the column block), piece fq of the row's 128-byte line ----
  const int fr = lane >> 3, fq = ((lane & 7) - 2 * ((fr >> 1) & 3)) & 7;
  const int kmax = (p.d - 1) & ~1;
  int ip[4], jp[TCH];                        // element offsets into V of the rows being requested (k ld < 2^31: bcx_gram_sk) --
                                             // one VGPR each instead of a pointer pair
  int ip_I = -1, jp_J = -1;
  auto set_i = [&](int I) {
    ip_I = I;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = I * GK_ROWS + 32 * wave + 8 * j + fr;
      ip[j] = (row < p.k ? row : p.k - 1) * (int)p.ld;
    }
  };
  auto set_j = [&](int J) {
    jp_J = J;
#pragma unroll
    for (int j = 0; j < TCH; ++j) {
      const int row = J * COLS + 8 * (TCH * wave + j) + fr;
      jp[j] = (row < p.k ? row : p.k - 1) * (int)p.ld;
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)gk_lds;
  auto issue_all = [&](const GkPos& a, int rs) {
    if (a.I != ip_I) set_i(a.I);
    if (a.J != jp_J) set_j(a.J);
    const int kc = min(a.s * GK_KC + 2 * fq, kmax);
#pragma unroll
    for (int j = 0; j < 4; ++j) gk_glds16(p.V + (ip[j] + kc), lds0 + (unsigned)(rs * GK_IBYTES + (4 * wave + j) * 1024));
#pragma unroll
    for (int j = 0; j < TCH; ++j) gk_glds16(p.V + (jp[j] + kc), lds0 + (unsigned)(JBASE + rs * JBYTES + (TCH * wave + j) * 1024));
  };
  // pieces beyond the row length (last stage, d not a multiple of 16): zeroed by the lane that requested them, after its
  // requests have landed and before the barrier.  Rows beyond k are copies of row k - 1: their products are never stored.
  auto zero_tail = [&](const GkPos& a, int rs) {
    const int k0 = a.s * GK_KC + 2 * fq;
    if (k0 + 1 < p.d) return;
#pragma unroll
    for (int j = 0; j < NREQ; ++j) {
      unsigned char* dst = j < 4 ? gk_lds + rs * GK_IBYTES + (4 * wave + j) * 1024 + lane * 16
                                 : gk_lds + JBASE + rs * JBYTES + (TCH * wave + j - 4) * 1024 + lane * 16;
      if (k0 >= p.d) *(gk2d*)dst = (gk2d){0.0, 0.0};
      else *(double*)(dst + 8) = 0.0;
    }
  };
  const bool ragged = (p.d & (GK_KC - 1)) != 0;
  // Read side.  v_mfma_f64_16x16x4_f64 takes one double per lane from each operand: lane (li = lane & 15, lk = lane >> 4)
  // supplies A[row li][k lk] and B[k lk][column li], and D[row (lane >> 4) + 4 reg][column lane & 15] comes back.  Both operands are
  // read in the "natural" form of csrc/proj.hip: one ds_read_b128 gives the lane the two consecutive values {2 lk, 2 lk + 1}
  // of an 8-value step of its row li -- the .x halves of the eight lanes groups form one 4-value MFMA step (values 0, 2, 4, 6),
  // the .y halves the next (1, 3, 5, 7); any assignment of values to steps is fine as long as both operands use the same one.
  // Byte offset inside a 16-row tile (2 chunks) for an even 8-value step; ^ 64 for an odd one.  A wave multiplies 64 rows by
  // 16 NTB columns: 4 + NTB reads per 8-value step feed 8 NTB MFMAs (the 4x4x4 form of the projection kernel needs 4.5 times
  // the LDS traffic per flop: it replicates one operand).
  const unsigned nat0 = (unsigned)((li >> 3) * 1024 + (li & 7) * 128 + ((lk + 2 * ((li >> 1) & 3)) & 7) * 16);
  const int wi = wave >> 1, wj = wave & 1;     // this wave's 64 rows / 16 NTB columns of the workgroup tile
  // acc[ta][tb][r] at lane (li, lk) = C[row 128 I + 64 wi + 16 ta + lk + 4 r][column COLS J + 16 NTB wj + 16 tb + li]
  gk4d acc[4][NTB];
#pragma unroll
  for (int ta = 0; ta < 4; ++ta)
#pragma unroll
    for (int tb = 0; tb < NTB; ++tb) acc[ta][tb] = (gk4d){0.0, 0.0, 0.0, 0.0};
  double* const mypart = p.part + (size_t)blockIdx.x * (GK_ROWS * COLS);

  // ---- the stage pipeline: rows are requested AHEAD stages ahead into rings of RING slots ----
  It cur = seg_start(0), n1 = succ(cur), n2 = succ(n1);
  int rs = 0;
  issue_all(cur.g, 0);
  if (AHEAD == 2 && n1.seg < 3) { issue_all(n1.g, 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NREQ) : "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ragged && cur.g.s == nst - 1) zero_tail(cur.g, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int piece_s0 = cur.g.s;                     // stage of its tile at which the current piece began
  for (;;) {
    const int rs1 = rs == RING - 1 ? 0 : rs + 1, rsq = AHEAD == 2 ? (rs1 == RING - 1 ? 0 : rs1 + 1) : rs1;
    const It& rq = AHEAD == 2 ? n2 : n1;       // the stage whose rows are requested while this one is multiplied
    const bool ahead = rq.seg < 3 && !(p.dbg & 1);
    int kcq = 0;
    if (ahead) {
      if (rq.g.I != ip_I) set_i(rq.g.I);
      if (rq.g.J != jp_J) set_j(rq.g.J);
      kcq = min(rq.g.s * GK_KC + 2 * fq, kmax);
    }
    {
      const unsigned char* ibase = gk_lds + rs * GK_IBYTES + wi * 8192;                   // this wave's four row tiles
      const unsigned char* jbase = gk_lds + JBASE + rs * JBYTES + wj * (NTB * 2048);      // and its NTB column tiles
      const unsigned char* ib[2] = {ibase + nat0, ibase + (nat0 ^ 64u)};                  // even / odd 8-value step
      const unsigned char* jb[2] = {jbase + nat0, jbase + (nat0 ^ 64u)};
      gk2d av[2][4], bv[2][NTB];
#pragma unroll
      for (int t = 0; t < 4; ++t) av[0][t] = *(const gk2d*)(ib[0] + t * 2048);
#pragma unroll
      for (int t = 0; t < NTB; ++t) bv[0][t] = *(const gk2d*)(jb[0] + t * 2048);
#pragma unroll
      for (int u = 0; u < GK_KC / 8; ++u) {
        if (u + 1 < GK_KC / 8) {              // the other 8-value step's operands are on their way while this one is multiplied
#pragma unroll
          for (int t = 0; t < 4; ++t) av[(u + 1) & 1][t] = *(const gk2d*)(ib[(u + 1) & 1] + t * 2048);
#pragma unroll
          for (int t = 0; t < NTB; ++t) bv[(u + 1) & 1][t] = *(const gk2d*)(jb[(u + 1) & 1] + t * 2048);
        }
        // this step's share of the LDS-DMA requests for the stage ahead
        if (ahead) {
#pragma unroll
          for (int q = u * (NREQ / 2); q < (u + 1) * (NREQ / 2) + (u == GK_KC / 8 - 1 ? NREQ % 2 : 0); ++q) {
            if (q < 4) gk_glds16(p.V + (ip[q < 4 ? q : 0] + kcq), lds0 + (unsigned)(rsq * GK_IBYTES + (4 * wave + q) * 1024));
            else gk_glds16(p.V + (jp[q >= 4 && q < NREQ ? q - 4 : 0] + kcq), lds0 + (unsigned)(JBASE + rsq * JBYTES + (TCH * wave + q - 4) * 1024));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
          for (int tb = 0; tb < NTB; ++tb)
            acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u & 1][ta].x, bv[u & 1][tb].x, acc[ta][tb], 0, 0, 0);
#pragma unroll
        for (int ta = 0; ta < 4; ++ta)
#pragma unroll
          for (int tb = 0; tb < NTB; ++tb)
            acc[ta][tb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u & 1][ta].y, bv[u & 1][tb].y, acc[ta][tb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const bool tile_done = cur.g.s == nst - 1;
    const bool seg_ends = n1.seg != cur.seg;
    bool stored = false;
    if (tile_done) {
      // ---- this workgroup holds the tile's last stage: add what the holders of its earlier stages left, then store ----
      if (piece_s0 > 0 && !(p.dbg & 2)) {
        // the contributors are the slots j_last .. slot - 1 of this XCD (every workgroup has a non-empty range: gram_sk_plan)
        const int64_t tile_u0 = (int64_t)(cur.g.tile - t_lo) * nst;
        int j_last = slot - 1;
        while (j_last > 0 && units * j_last / per > tile_u0) --j_last;
        const int nc = slot - j_last;
        // Few contributors (the usual case): they are taken one at a time, nearest first -- wait for its flag, add its share --
        // so that a late one does not hold up the others.  Many (a small support with long rows: tens per tile): lane c of
        // wave 0 polls contributor c, all at once.  (A/B on one box, us per call, all flags at once + shares requested one
        // contributor ahead: 20.6 / 50 / 71 against 24.2 / 53.7 / 87 at k, d = 400, 512 / 1024, 1024 / 512, 4096, but 89.5 against
        // 79.7 at 1497, 1024 with its 3-4 contributors; the request-ahead form also cost the narrow tile 80 VGPRs.)
        const bool many = nc > 4;
        auto wait_for = [&](int c) {            // thread 0: contributor c's flag
          const int b = 8 * (slot - 1 - c) + xcd;
          const long long t0 = wall_clock64();
          // relaxed polls, ONE acquire fence when the flag is up (the hand-off recipe of csrc/nnls_common.h; acquire polls
          // measured the same to 1-2 %)
          while (__hip_atomic_load(&p.flags[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > p.timeout_ticks) { atomicExch(p.status, (unsigned long long)p.epoch); break; }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        };
        if (many) {
          if (wave == 0) {
            for (int c0 = 0; c0 < nc; c0 += 64)
              if (c0 + lane < nc) wait_for(c0 + lane);
          }
          __syncthreads();
        }
        // partial tiles in register order ([wave][ta][tb][register pair][lane] pairs of doubles), added nearest contributor
        // first, half a share (8 NTB 16-byte loads per lane) in flight at a time
        auto part_of = [&](int c) {
          return p.part + (size_t)(8 * (slot - 1 - c) + xcd) * (GK_ROWS * COLS) + (size_t)wave * (4 * NTB * 4 * 64) + 2 * lane;
        };
        {
          for (int c = 0; c < nc; ++c) {
            if (!many) {
              if (tid == 0) wait_for(c);
              __syncthreads();
            }
            const double* src = part_of(c);
            asm volatile("" : "+v"(src));
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              gk2d v[2][NTB][2];
#pragma unroll
              for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
                  for (int h = 0; h < 2; ++h) v[ta][tb][h] = gk_ld2(src + (((2 * half + ta) * NTB + tb) * 2 + h) * 128);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
              for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
                  for (int h = 0; h < 2; ++h) {
                    asm volatile("" : "+v"(v[ta][tb][h]));      // (volatile asms keep their order: the values are read after the wait)
                    acc[2 * half + ta][tb][2 * h] += v[ta][tb][h].x; acc[2 * half + ta][tb][2 * h + 1] += v[ta][tb][h].y;
                  }
            }
          }
        }
      }
      // Only the entries on or above the diagonal are taken from the accumulators, each together with its mirror image: an
      // entry below the diagonal of a block on the diagonal may also be formed (with its operands swapped) in ANOTHER tile, whose
      // stages may be split differently -- same products, other rounding.  One source per entry pair keeps G symmetric bit for bit.
      // The entries themselves go out as the accumulators hold them: 16 consecutive columns of a row per 16-lane row (128-byte
      // lines).  For the mirror image the 16 x 16 tile is transposed inside the wave first -- register r of lane row lk holds
      // row lk + 4 r; after exchanging registers with lane rows (a 4 x 4 transpose: two permlane swaps per pair) lane row lk
      // holds rows 4 lk .. 4 lk + 3 of its column, i.e. four CONSECUTIVE entries of a row of the mirror image: two 16-byte
      // stores, the four lane rows of a column filling one 128-byte line (as 8-byte stores 32 KiB apart -- one entry per line
      // and instruction -- the mirror images of all workgroups' tiles took ~45 us of the call at k = 4096).
      const int rowb = cur.g.I * GK_ROWS + 64 * wi, col0 = cur.g.J * COLS + 16 * NTB * wj + li;
      const bool vec_ok = (p.ldg & 1) == 0 && ((uintptr_t)p.G & 15) == 0;
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) {
          const int col = col0 + 16 * tb;
          double x0 = acc[ta][tb][0], x1 = acc[ta][tb][1], x2 = acc[ta][tb][2], x3 = acc[ta][tb][3];
          {
            const int row = rowb + 16 * ta + lk;
            if (col < p.k) {
              double* g = p.G + (size_t)row * p.ldg + col;
              if (row <= col) g[0] = x0;
              if (row + 4 <= col) g[4 * p.ldg] = x1;
              if (row + 8 <= col) g[8 * p.ldg] = x2;
              if (row + 12 <= col) g[12 * p.ldg] = x3;
            }
          }
          gk_swap16(x0, x1); gk_swap16(x2, x3); gk_swap32(x0, x2); gk_swap32(x1, x3);
          {
            const int row = rowb + 16 * ta + 4 * lk;      // x0 .. x3: rows row .. row + 3 of column col
            if (col < p.k && row <= col) {
              double* g = p.G + (size_t)col * p.ldg + row;
              if (vec_ok && row + 3 <= col) { *(gk2d*)g = (gk2d){x0, x1}; *(gk2d*)(g + 2) = (gk2d){x2, x3}; }
              else {
                g[0] = x0;
                if (row + 1 <= col) g[1] = x1;
                if (row + 2 <= col) g[2] = x2;
                if (row + 3 <= col) g[3] = x3;
              }
            }
          }
        }
      stored = true;
    } else if (seg_ends && !(p.dbg & 2)) {
      // ---- the piece ends inside a tile: leave the partial accumulators for the workgroup that finishes it, raise the flag ----
      double* dst = mypart + (size_t)wave * (4 * NTB * 4 * 64) + 2 * lane;
      asm volatile("" : "+v"(dst));           // (formed here: not hoisted out of the stage loop)
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb)
#pragma unroll
          for (int h = 0; h < 2; ++h) gk_st2(dst + ((ta * NTB + tb) * 2 + h) * 128, (gk2d){acc[ta][tb][2 * h], acc[ta][tb][2 * h + 1]});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&p.flags[blockIdx.x], p.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      stored = true;
    }
    if (n1.seg == 3) break;
    if (tile_done || seg_ends) {
#pragma unroll
      for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < NTB; ++tb) acc[ta][tb] = (gk4d){0.0, 0.0, 0.0, 0.0};
      piece_s0 = n1.g.s;
    }
    // the NEXT stage's rows have to have landed; with two stages ahead the requests of the stage after it stay in flight
    // (they complete in issue order).  After an epilogue the queue also holds stores, which do not: drain it all.
    if (AHEAD == 2 && ahead && !stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NREQ) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ragged && n1.g.s == nst - 1) zero_tail(n1.g, rs1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur = n1; n1 = n2; n2 = succ(n2);
    rs = rs1;
  }
}

// ---- host side -----------------------------------------------------------------------------------------------------
struct GramSkPlan { int ntb, nI, nJ, ntiles, nst, wgs; };

// Resident workgroups of the kernel on the CURRENT device (the function attribute and the CU count are per device: a process
// that drives several GPUs gets each one's own figure).
template <int NTB> static int gk_resident_wgs() {
  static std::atomic<int> cache[64];                   // 0: not yet asked on that device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int n = cache[dev].load(std::memory_order_acquire);
  if (n > 0) return n;
  int cus = 256, per_cu = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  (void)hipFuncSetAttribute((const void*)gram_sk_kernel<NTB>, hipFuncAttributeMaxDynamicSharedMemorySize, GK_LDS_BYTES(NTB));
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gram_sk_kernel<NTB>, 256, GK_LDS_BYTES(NTB)) != hipSuccess || per_cu < 1)
    per_cu = 1;
  if (per_cu > 2) per_cu = 2;
  n = std::max(8, cus * per_cu / 8 * 8);
  cache[dev].store(n, std::memory_order_release);
  return n;
}

// Applies to rows that the LDS-DMA can fetch (16-byte pieces) and supports large enough to be worth a chip-wide launch.
static bool gram_sk_plan(int k, int d, GramSkPlan* pl) {
  if (k < 192 || d < 4 * GK_KC) return false;
  static const int forced = [] { const char* e = bcx_dev_env("BCX_GRAM_NTB"); return e ? atoi(e) : 0; }();      // dev: 2 / 4 / -1 (gram_tile_kernel)
  if (forced < 0) return false;
  pl->ntb = forced == 2 || forced == 4 ? forced : (k >= 3072 ? 4 : 2);
  const int cols = GK_COLS(pl->ntb);
  pl->nI = (k + GK_ROWS - 1) / GK_ROWS;
  pl->nJ = (k + cols - 1) / cols;
  pl->ntiles = 0;
  for (int I = 0; I < pl->nI; ++I) pl->ntiles += pl->nJ - (I * GK_ROWS) / cols;
  pl->nst = (d + GK_KC - 1) / GK_KC;
  const int64_t units = (int64_t)pl->ntiles * pl->nst;
  const int resident = pl->ntb == 4 ? gk_resident_wgs<4>() : gk_resident_wgs<2>();
  const int64_t want = std::max<int64_t>(1, units / GK_MIN_STAGES);
  pl->wgs = (int)std::min<int64_t>(resident, (want + 7) / 8 * 8);
  return true;
}

// scratch: status (64 bytes, at a place that does not depend on the plan: bcx_gram_sk_timed_out) | partial tiles | flags
int64_t bcx_gram_sk_scratch_bytes(int k, int d) {
  GramSkPlan pl;
  if (!gram_sk_plan(k, d, &pl)) return 0;
  return (int64_t)pl.wgs * GK_ROWS * GK_COLS(pl.ntb) * 8 + (int64_t)pl.wgs * 8 + 64;
}

// Flag words carry the number of the call that raised them: it starts from the clock, so the stale contents of a scratch
// buffer (an earlier call, an earlier process) never match a later call's number.
static const unsigned long long gram_epoch0 = ((unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() << 16) | 1ull;
static std::atomic<unsigned long long>& gram_epoch_counter() {
  static std::atomic<unsigned long long> e{gram_epoch0};
  return e;
}
static unsigned long long gram_next_epoch() { return gram_epoch_counter().fetch_add(1) + 1; }
// Every call launched after this value was read carries a later number (since == 0: every call of this process).
unsigned long long bcx_gram_sk_epoch_now() { return gram_epoch_counter().load(); }
// Did a workgroup of a call numbered after `since` that used the scratch `work` give up waiting for a peer's partial tile
// (the GPU shared with another process, preemption, fewer resident workgroups than planned)?  Its G is then not valid.
// Synchronises the stream.  1: yes, 0: no, < 0: HIP error.  (Differences, not magnitudes: the counter starts from the clock
// and may wrap; stale contents of the scratch fall outside the few numbers issued since.)
int bcx_gram_sk_timed_out(hipStream_t st, const double* work, unsigned long long since) {
  unsigned long long v = 0;
  if (hipMemcpyAsync(&v, work, sizeof v, hipMemcpyDeviceToHost, st) != hipSuccess) return BCX_ERR_HIP;
  if (hipStreamSynchronize(st) != hipSuccess) return BCX_ERR_HIP;
  if (since == 0) since = gram_epoch0;
  const unsigned long long age = v - since, span = gram_epoch_counter().load() - since;
  return age != 0 && age <= span ? 1 : 0;
}

// 1: not applicable (alignment, size: the caller uses gram_tile_kernel), 0: launched, < 0: error
int bcx_gram_sk(hipStream_t st, const double* rows, int k, int d, int64_t ld, double* G, int64_t ldg, double* work) {
  GramSkPlan pl;
  if (((uintptr_t)rows & 15) != 0 || (ld & 1) != 0 || (int64_t)k * ld >= (1ll << 31) || !gram_sk_plan(k, d, &pl)) return 1;
  GramSkArgs a;
  a.V = rows; a.G = G; a.ld = ld; a.ldg = ldg; a.k = k; a.d = d;
  a.nI = pl.nI; a.nJ = pl.nJ; a.ntiles = pl.ntiles; a.nst = pl.nst;
  a.status = (unsigned long long*)work;
  a.part = work + 8;
  a.flags = (unsigned long long*)(a.part + (size_t)pl.wgs * GK_ROWS * GK_COLS(pl.ntb));
  a.epoch = gram_next_epoch();                      // (never 0)
  static const int dbg = [] { const char* e = bcx_dev_env("BCX_GRAM_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  a.timeout_ticks = 500000000LL;                    // 5 s of the 100 MHz wall clock
  if (pl.ntb == 4) hipLaunchKernelGGL(gram_sk_kernel<4>, dim3(pl.wgs), dim3(256), GK_LDS_BYTES(4), st, a);
  else hipLaunchKernelGGL(gram_sk_kernel<2>, dim3(pl.wgs), dim3(256), GK_LDS_BYTES(2), st, a);
  return hipGetLastError() == hipSuccess ? BCX_OK : BCX_ERR_HIP;
}
