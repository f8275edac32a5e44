// gram.hip -- G = V V^T over the k active rows of the dense re-weight (optimize(), snnls.py:82-97; OMP's nnls call,
// orthopursuit.py:37-42) on the fp64 matrix cores, balanced over the whole chip ("stream-K").
//
// Why a second Gram kernel (the first one, gram_tile_kernel in moments.hip, stays as the path for rows that are not
// 16-byte aligned and for small supports): that kernel hands out whole 64 x 64 blocks of the upper triangle, stages both
// operands through registers with two workgroup barriers per 32 values, and uses v_mfma_f64_16x16x4_f64, which tops out
// at 47.6 TFLOP/s on this chip (tools/probe/mfma_f64_peak.hip).  At k = 4096 that is 528 x 4 block pairs for 512
// resident workgroups -- a ragged last round -- and 0.51 of the fp64 MFMA peak.  Here:
//   * the inner loop is the projection kernel's (csrc/proj.hip): 128 x 64 workgroup tiles, operands fetched by LDS-DMA
//     (global_load_lds_dwordx4, no staging registers) TWO 16-value stages ahead into three-slot rings, one counted wait +
//     one bare barrier per stage, v_mfma_f64_4x4x4_4b_f64 (72.8 TFLOP/s register-only) with one operand natural and one
//     replicated, rotated 128-byte LDS lines so that every ds_read_b128 is conflict free;
//   * the work is the SEQUENCE of (tile, stage) units of the upper triangle, cut into equal contiguous ranges, one per
//     resident workgroup: every workgroup multiplies for the same time whatever k is.  A tile whose stages span several
//     workgroups is finished by the one that holds its last stage: the others leave their partial accumulators in
//     scratch (register layout, 512-byte coalesced) and raise a flag; the finisher adds them in a fixed order (its own
//     part first, then the contributors from the nearest to the farthest), so the result is deterministic for a given
//     (k, d, workgroup count);
//   * the tile sequence is cut into eight contiguous pieces, one per XCD (workgroups b, b + 8, ... share an XCD under
//     round-robin dispatch: a speed assumption, not a correctness one), so the workgroups of an XCD walk neighbouring tiles
//     of the same block rows and find one of the two row panels in their L2; a workgroup only ever waits for workgroups
//     with a LOWER blockIdx (b - 8, b - 16, ...), which were dispatched before it.
// Both triangles are written, every entry pair from ONE accumulator (the one on or above the diagonal, stored together with
// its mirror image): G is symmetric bit for bit.
#include <algorithm>
#include <atomic>
#include <chrono>
#include "bcx_internal.h"
#include "dev_util.h"

typedef double gk4d __attribute__((ext_vector_type(4)));
typedef double gk2d __attribute__((ext_vector_type(2)));

#define GK_KC 16                      // values of the row length per stage
#define GK_ROWS 128                   // rows of G per tile (4 waves x 32)
#define GK_IBYTES (GK_ROWS * GK_KC * 8)
#define GK_COLS(NCT) (16 * (NCT))
#define GK_JBYTES(NCT) (GK_COLS(NCT) * GK_KC * 8)
#define GK_RING 3                     // stages of rows resident in LDS (requests run two stages ahead)
#define GK_JBASE (GK_RING * GK_IBYTES)
#define GK_LDS_BYTES(NCT) (GK_JBASE + GK_RING * GK_JBYTES(NCT))
#define GK_NCT 4                      // 16-column tiles per workgroup tile: 128 x 64 (128 x 128 needs 96 KiB for the three-slot
                                      // rings -- one workgroup per CU -- and measured slower with two slots: 0.52 against 0.59 at k = 4096)
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

struct GkPos { int tile, s, I, J; };

// first column block of block row I that holds an entry on or above the diagonal
template <int NCT> static __host__ __device__ __forceinline__ int gk_jmin(int I) { return (I * GK_ROWS) / GK_COLS(NCT); }

template <int NCT>
__global__ __launch_bounds__(256, 2) void gram_sk_kernel(GramSkArgs p) {
  constexpr int COLS = GK_COLS(NCT), JBYTES = GK_JBYTES(NCT), TCH = NCT / 2, NREQ = 4 + TCH;
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
    while (t >= p.nJ - gk_jmin<NCT>(I)) { t -= p.nJ - gk_jmin<NCT>(I); ++I; }
    a.I = I; a.J = gk_jmin<NCT>(I) + t;
    return a;
  };
  auto advance = [&](GkPos a) {
    if (++a.s == nst) {
      a.s = 0; a.tile += 1;
      if (++a.J == p.nJ) { a.I += 1; a.J = gk_jmin<NCT>(a.I); }
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
    for (int j = 0; j < TCH; ++j) gk_glds16(p.V + (jp[j] + kc), lds0 + (unsigned)(GK_JBASE + rs * JBYTES + (TCH * wave + j) * 1024));
  };
  // pieces beyond the row length (last stage, d not a multiple of 16): zeroed by the lane that requested them, after its
  // requests have landed and before the barrier.  Rows beyond k are copies of row k - 1: their products are never stored.
  auto zero_tail = [&](const GkPos& a, int rs) {
    const int k0 = a.s * GK_KC + 2 * fq;
    if (k0 + 1 < p.d) return;
#pragma unroll
    for (int j = 0; j < NREQ; ++j) {
      unsigned char* dst = j < 4 ? gk_lds + rs * GK_IBYTES + (4 * wave + j) * 1024 + lane * 16
                                 : gk_lds + GK_JBASE + rs * JBYTES + (TCH * wave + j - 4) * 1024 + lane * 16;
      if (k0 >= p.d) *(gk2d*)dst = (gk2d){0.0, 0.0};
      else *(double*)(dst + 8) = 0.0;
    }
  };
  const bool ragged = (p.d & (GK_KC - 1)) != 0;
  // read-side bases of this lane inside a 16-row operand tile (csrc/proj.hip): natural piece (row li, k-slot lk) and replicated
  // piece (row lane & 3 of the strip, k-slot lk) of an even 8-value step; ^ 64 for an odd step
  const unsigned nat0 = (unsigned)((li >> 3) * 1024 + (li & 7) * 128 + ((lk + 2 * ((li >> 1) & 3)) & 7) * 16);
  const unsigned rep0 = (unsigned)((lane & 3) * 128 + ((lk + 2 * ((lane >> 1) & 1)) & 7) * 16);
  // acc[i][t][r] at lane (li, lk) = C[row 128 I + 32 wave + 16 i + li][column COLS J + 16 t + 4 r + lk]
  gk4d acc[2][NCT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < NCT; ++t) acc[i][t] = (gk4d){0.0, 0.0, 0.0, 0.0};
  double* const mypart = p.part + (size_t)blockIdx.x * (GK_ROWS * COLS);

  // ---- the stage pipeline: rows are requested TWO stages ahead into a ring of three slots (the panels come from L2 / the
  // infinity cache, a microsecond or two away under load; one stage of this tile is ~1.5 us) ----
  It cur = seg_start(0), n1 = succ(cur), n2 = succ(n1);
  int rs = 0;
  issue_all(cur.g, 0);
  if (n1.seg < 3) { issue_all(n1.g, 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NREQ) : "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ragged && cur.g.s == nst - 1) zero_tail(cur.g, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int piece_s0 = cur.g.s;                     // stage of its tile at which the current piece began
  for (;;) {
    const int rs1 = rs == GK_RING - 1 ? 0 : rs + 1, rs2 = rs1 == GK_RING - 1 ? 0 : rs1 + 1;
    const bool ahead = n2.seg < 3;
    int kc2 = 0;
    if (ahead) {
      if (n2.g.I != ip_I) set_i(n2.g.I);
      if (n2.g.J != jp_J) set_j(n2.g.J);
      kc2 = min(n2.g.s * GK_KC + 2 * fq, kmax);
    }
    {
      const unsigned char* ib = gk_lds + rs * GK_IBYTES + wave * 4096;          // this wave's two row tiles (natural operand)
      const unsigned char* jb = gk_lds + GK_JBASE + rs * JBYTES;                // the column block's NCT tiles (replicated operand)
      const unsigned char* natb[2] = {ib + nat0, ib + (nat0 ^ 64u)};
      const unsigned char* repb[2] = {jb + rep0, jb + (rep0 ^ 64u)};
      constexpr int NU = GK_KC / 8, NG = NCT * NU;
      auto nat_ptr = [&](int uu, int t) { return (const gk2d*)(natb[uu & 1] + t * 2048); };
      auto rep_ptr = [&](int g, int r) {
        const int uu = g / NCT, t = g % NCT;
        return (const gk2d*)(repb[(uu + r) & 1] + t * 2048 + (r >> 1) * 1024 + (r & 1) * 512);
      };
      gk2d rp[2][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rp[0][r] = *rep_ptr(0, r);
      gk2d nt[2];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int uu = g / NCT, t = g % NCT;
        if (t == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) nt[i] = *nat_ptr(uu, i);
        }
        if (g + 1 < NG) {
#pragma unroll
          for (int r = 0; r < 4; ++r) rp[(g + 1) & 1][r] = *rep_ptr(g + 1, r);
        }
        // this group's share of the LDS-DMA requests of the stage after next
        if (ahead && g < NREQ) {
          if (g < 4) gk_glds16(p.V + (ip[g < 4 ? g : 0] + kc2), lds0 + (unsigned)(rs2 * GK_IBYTES + (4 * wave + g) * 1024));
          else gk_glds16(p.V + (jp[g >= 4 && g < NREQ ? g - 4 : 0] + kc2), lds0 + (unsigned)(GK_JBASE + rs2 * JBYTES + (TCH * wave + g - 4) * 1024));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][t][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(rp[g & 1][r].x, nt[i].x, acc[i][t][r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][t][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(rp[g & 1][r].y, nt[i].y, acc[i][t][r], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    const bool tile_done = cur.g.s == nst - 1;
    const bool seg_ends = n1.seg != cur.seg;
    bool stored = false;
    if (tile_done) {
      // ---- this workgroup holds the tile's last stage: add what the holders of its earlier stages left, then store ----
      if (piece_s0 > 0) {
        const int64_t tile_u0 = (int64_t)(cur.g.tile - t_lo) * nst;
        for (int j = slot - 1; j >= 0; --j) {
          const int64_t ju0 = units * j / per, ju1 = units * (j + 1) / per;
          if (ju1 <= tile_u0) break;
          if (ju0 < ju1) {
            const int b = 8 * j + xcd;
            if (tid == 0) {
              const long long t0 = wall_clock64();
              while (__hip_atomic_load(&p.flags[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != p.epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (wall_clock64() - t0 > p.timeout_ticks) { atomicExch(p.status, (unsigned long long)p.epoch); break; }
              }
            }
            __syncthreads();
            // (partial tile in register order: [wave][i][t][register pair][lane] pairs of doubles; all 4 NCT loads of the
            // workgroup's share in flight together: one trip to the memory side per contributor)
            const double* src = p.part + (size_t)b * (GK_ROWS * COLS) + (size_t)wave * (2 * NCT * 4 * 64) + 2 * lane;
            asm volatile("" : "+v"(src));
            gk2d v[2][NCT][2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) v[i][t][h] = gk_ld2(src + ((i * NCT + t) * 2 + h) * 128);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int t = 0; t < NCT; ++t)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                  asm volatile("" : "+v"(v[i][t][h]));      // (volatile asms keep their order: the values are read after the wait)
                  acc[i][t][2 * h] += v[i][t][h].x; acc[i][t][2 * h + 1] += v[i][t][h].y;
                }
          }
          if (ju0 <= tile_u0) break;
        }
      }
      // Only the entries on or above the diagonal are taken from the accumulators, each together with its mirror image: an
      // entry below the diagonal of a block on the diagonal is also formed (with its operands swapped) in ANOTHER tile, whose
      // stages may be split differently -- same products, other rounding.  One source per entry pair keeps G symmetric bit for bit.
      const int row0 = cur.g.I * GK_ROWS + 32 * wave + li, col0 = cur.g.J * COLS + lk;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = row0 + 16 * i;
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = col0 + 16 * t + 4 * r;
            if (row <= col && col < p.k) {
              const double v = acc[i][t][r];
              p.G[(size_t)row * p.ldg + col] = v;
              p.G[(size_t)col * p.ldg + row] = v;
            }
          }
      }
      stored = true;
    } else if (seg_ends) {
      // ---- the piece ends inside a tile: leave the partial accumulators for the workgroup that finishes it, raise the flag ----
      double* dst = mypart + (size_t)wave * (2 * NCT * 4 * 64) + 2 * lane;
      asm volatile("" : "+v"(dst));           // (formed here: not hoisted out of the stage loop)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NCT; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) gk_st2(dst + ((i * NCT + t) * 2 + h) * 128, (gk2d){acc[i][t][2 * h], acc[i][t][2 * h + 1]});
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(&p.flags[blockIdx.x], p.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      stored = true;
    }
    if (n1.seg == 3) break;
    if (tile_done || seg_ends) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int t = 0; t < NCT; ++t) acc[i][t] = (gk4d){0.0, 0.0, 0.0, 0.0};
      piece_s0 = n1.g.s;
    }
    // the NEXT stage's rows have to have landed; the requests of the stage after it stay in flight (they complete in issue
    // order).  After an epilogue the queue also holds stores, which do not: drain it all.
    if (ahead && !stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NREQ) : "memory");
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
struct GramSkPlan { int nct, nI, nJ, ntiles, nst, wgs; };

template <int NCT> static int gk_resident_wgs() {
  static const int n = [] {
    int dev = 0, cus = 256, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    (void)hipFuncSetAttribute((const void*)gram_sk_kernel<NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, GK_LDS_BYTES(NCT));
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)gram_sk_kernel<NCT>, 256, GK_LDS_BYTES(NCT)) != hipSuccess || per_cu < 1)
      per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    return std::max(8, cus * per_cu / 8 * 8);
  }();
  return n;
}

// Applies to rows that the LDS-DMA can fetch (16-byte pieces) and supports large enough to be worth a chip-wide launch.
static bool gram_sk_plan(int k, int d, GramSkPlan* pl) {
  if (k < 192 || d < 4 * GK_KC) return false;
  // (the kernel keeps row offsets as 32-bit element offsets; the caller checks k * ld)
  static const bool tiled = bcx_dev_env("BCX_GRAM_TILED") != nullptr;      // dev: gram_tile_kernel for everything
  if (tiled) return false;
  pl->nct = GK_NCT;
  const int cols = 16 * pl->nct;
  pl->nI = (k + GK_ROWS - 1) / GK_ROWS;
  pl->nJ = (k + cols - 1) / cols;
  pl->ntiles = 0;
  for (int I = 0; I < pl->nI; ++I) pl->ntiles += pl->nJ - (I * GK_ROWS) / cols;
  pl->nst = (d + GK_KC - 1) / GK_KC;
  const int64_t units = (int64_t)pl->ntiles * pl->nst;
  const int resident = gk_resident_wgs<GK_NCT>();
  const int64_t want = std::max<int64_t>(1, units / GK_MIN_STAGES);
  pl->wgs = (int)std::min<int64_t>(resident, (want + 7) / 8 * 8);
  return true;
}

// scratch: partial tiles | flags | status
int64_t bcx_gram_sk_scratch_bytes(int k, int d) {
  GramSkPlan pl;
  if (!gram_sk_plan(k, d, &pl)) return 0;
  return (int64_t)pl.wgs * GK_ROWS * 16 * pl.nct * 8 + (int64_t)pl.wgs * 8 + 64;
}

// Flag words carry the number of the call that raised them: it starts from the clock, so the stale contents of a scratch
// buffer (an earlier call, an earlier process) never match a later call's number.
static unsigned long long gram_next_epoch() {
  static std::atomic<unsigned long long> e{((unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() << 16) | 1ull};
  return e.fetch_add(1) + 1;
}

// 1: not applicable (alignment, size: the caller uses gram_tile_kernel), 0: launched, < 0: error
int bcx_gram_sk(hipStream_t st, const double* rows, int k, int d, int64_t ld, double* G, int64_t ldg, double* work) {
  GramSkPlan pl;
  if (((uintptr_t)rows & 15) != 0 || (ld & 1) != 0 || (int64_t)k * ld >= (1ll << 31) || !gram_sk_plan(k, d, &pl)) return 1;
  GramSkArgs a;
  a.V = rows; a.G = G; a.ld = ld; a.ldg = ldg; a.k = k; a.d = d;
  a.nI = pl.nI; a.nJ = pl.nJ; a.ntiles = pl.ntiles; a.nst = pl.nst;
  a.part = work;
  a.flags = (unsigned long long*)(work + (size_t)pl.wgs * GK_ROWS * 16 * pl.nct);
  a.status = a.flags + pl.wgs;
  a.epoch = gram_next_epoch();                      // (never 0)
  a.timeout_ticks = 500000000LL;                    // 5 s of the 100 MHz wall clock
  hipLaunchKernelGGL(gram_sk_kernel<GK_NCT>, dim3(pl.wgs), dim3(256), GK_LDS_BYTES(GK_NCT), st, a);
  return hipGetLastError() == hipSuccess ? BCX_OK : BCX_ERR_HIP;
}
