// proj.hip -- device-native log-likelihood projection:
//   vecs[n][s] = loglik(z_n, theta_s) - mean_s' loglik(z_n, theta_s')        projector.py:19-21
// for the three example likelihoods of the reference
//   logistic regression   examples/common/model_lr.py:25-32
//   Poisson (softplus)    examples/common/model_poiss.py:25-38
//   Gaussian linear regr. examples/common/model_linreg.py:4-10
// as one fused kernel: Z . Theta^T on the fp64 matrix cores (v_mfma_f64_16x16x4_f64; operands staged through LDS by
// LDS-DMA, 128 x 64 or 128 x 128 workgroup tiles) with the likelihood as the epilogue, and three
// consumers that never need the N x S matrix twice:
//   WRITE   : store the uncentred values (a second pass forms the row means and subtracts them)
//   COLSUM  : only the column sums  sum_n vecs[n][s]        (SparseVI gradient, sparsevi.py:70-74)
//   SELECT  : per row  corr_n = vecs[n].resid / ||vecs[n]|| / S  and its arg-max   (sparsevi.py:44-55)
// Arithmetic is fp64 throughout (the selection compares correlations to ~1e-7).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <string>
#include <utility>
#include <vector>
#include "bcx_internal.h"
#include "dev_util.h"
#include "proj_math.h"
#include "moments_quad.h"

enum { FAM_LOGISTIC = 0, FAM_POISSON = 1, FAM_LINREG = 2 };
enum { PMODE_WRITE = 0, PMODE_COLSUM = 1, PMODE_SELECT = 2 };

typedef double pv4d __attribute__((ext_vector_type(4)));

struct ProjArgs {
  const double* Z;      // N x ldz: features in columns [0, D), response (if any) in column ycol
  const double* theta;  // S x ldt
  int64_t N;
  int64_t ldz;
  int ldt, D, S, ycol;
  double param;         // linreg: sigma^2
  double* out;          // WRITE: N x ldo
  int64_t ldo;
  double* rowsum;       // (unused)
  double* colpart;      // COLSUM: gridDim.x x S partial column sums
  const double* resid;  // SELECT: S
  double resid_sum;     // SELECT: sum_s resid[s]
  int team;             // column groups of a row block spread over `team` workgroups of one XCD (0: one workgroup walks them)
  double* part;         // SELECT on teams: ngc x N records {shift, s1, s2, sd} of partial row moments (select_combine_kernel)
  double* best_val;     // SELECT: gridDim.x
  int64_t* best_idx;    // SELECT: gridDim.x
  const double* tab;    // the likelihood's tables (proj_math.h), PJT_DOUBLES doubles in device memory
};

// gammaln(y + 1) of the Poisson likelihood (model_poiss.py:37).  Only WRITE forms it: along a row it is a constant, which
// cancels in (value - shift) of the column sums and of the row moments.  Responses are counts: log(y!) comes from a table
// for the integers 0 .. 255 (rounded from long double); anything else goes to the library lgamma as a real call --
// inlined, its polynomial tables and temporaries land in the epilogue's register budget (WRITE spilled 166 VGPRs).  The
// library routine is some hundreds of fp64 instructions on the unit the MFMAs run on, and WRITE needs it for 8 rows
// per lane and tile of 32 values.
__device__ __attribute__((noinline)) double pj_lgamma1p_call(double y) { return lgamma(y + 1.0); }

// The transcendental pieces (exp on (-inf, 0], log1p on [0, 1], log of a positive number) are table driven: proj_math.h.
// Their tables (PJT_DOUBLES doubles, built on the host from long double) sit behind the kernel's other LDS regions.
typedef const double __attribute__((address_space(3)))* pj_tab_t;

// log(1 + exp(t)) = max(t, 0) + log1p(exp(-|t|))
__device__ __forceinline__ double pj_softplus(double t, pj_tab_t tab) { return pjm_relu(t) + pjm_log1p01(pjm_exp_nonpos(-fabs(t), tab), tab); }

// (Round 2/3 evaluated the logistic and Poisson likelihoods as real calls in SELECT and WRITE: inlined, the temporaries of
// their long series spilled registers inside the k loop.  The table forms are short enough to inline everywhere.)

template <int FAM> __device__ __forceinline__ double loglik(double m, double y, double param, double c0, pj_tab_t tab) {
  if (FAM == FAM_LOGISTIC) {
    // model_lr.py:29-31 switches from -log1p(exp(t)) to -t at t = 100: there exp(-t) < 4e-44 and max(t, 0) + log1p(.) IS t to
    // every bit, so the softplus form needs no branch.
    // -log1p(exp(-m)) = min(m, 0) - log1p(exp(-|m|)); min(m, 0) by the sign bit (pjm_relu)        model_lr.py:28-31
    return (pjm_hi(m) < 0 ? m : 0.0) - pjm_log1p01(pjm_exp_nonpos(-fabs(m), tab), tab);
  } else if (FAM == FAM_POISSON) {
    // model_poiss.py:25-38: s' = log(lam) with the rate lam = log(1 + e^s) = max(s, 0) + log1p(exp(-|s|)) where s > -100,
    // s' = s below (there lam = e^s to every bit: log1p(e) = e for e < 4e-44); log-likelihood y s' - gammaln(y + 1) - exp(s').
    // exp(s') IS lam: the rate is used as computed instead of exponentiating its logarithm again (one 42-instruction exp
    // less per element, and a few ulp closer to the exact value than the reference's exp(log(.))).
    const double lam = pj_softplus(m, tab);
    const double sl = m > -100.0 ? pjm_log_pos(lam, tab) : m;
    return y * sl - c0 - lam;                              // (c0 = gammaln(y+1))
  } else {
    // model_linreg.py:10: -0.5 log(2 pi sigsq) - (y^2 - 2 m y + m^2) / (2 sigsq).  The row's own part is formed once per
    // row by the caller -- it hands in 2 y for y and c0 = -0.5 log(2 pi sigsq) - y^2 / (2 sigsq) -- and param = 1 / (2 sigsq)
    // once per kernel, so an element costs three fp64 instructions (subtract, multiply, multiply-add) on the unit the
    // MFMAs run on instead of seven.
    return fma(-((m - y) * m), param, c0);
  }
}

// loglik - shift.  For the column sums of the linear-regression family the shift is the likelihood at a zero linear
// predictor (see the epilogue), and the difference has a closed form without the cancellation:
//   [c0 - (y^2 - 2 m y + m^2) / (2 sigsq)] - [c0 - y^2 / (2 sigsq)] = (2 y - m) m / (2 sigsq)
// -- three operations instead of seven per element, and more accurate than forming both terms.
// (The epilogue hands in 2 y for y and applies the common factor 1 / (2 sigsq) once per column sum: two fp64 instructions
// per element -- subtract, multiply-add into the sum -- on the unit the MFMAs run on.)
// SELECT uses the same form about the value in the group's first column, (2 y - m) m - (2 y - m0) m0, and leaves the
// factor out altogether: a correlation does not change when its vector is scaled by 2 sigsq > 0 (the records of p.part
// are then moments of vecs * 2 sigsq, consistently; select_combine_kernel's quotient is the same).
template <int FAM, int MODE> __device__ __forceinline__ double loglik_shifted(double m, double y, double param, double c0, double shift, pj_tab_t tab) {
  if (FAM == FAM_LINREG && MODE == PMODE_COLSUM) return (y - m) * m;
  if (FAM == FAM_LINREG && MODE == PMODE_SELECT) return fma(y - m, m, -shift);
  return loglik<FAM>(m, y, param, c0, tab) - shift;
}

// sum over the 16 lanes of a DPP row (lanes that share l >> 4)
__device__ __forceinline__ double row16_sum(double v) {
  v += bcx_dpp_f64<0xB1>(v);
  v += bcx_dpp_f64<0x4E>(v);
  v += bcx_dpp_f64<0x141>(v);
  v += bcx_dpp_f64<0x140>(v);
  return v;
}

// ---- geometry of the LDS-staged kernel ---------------------------------------------------------------------
// A workgroup (4 waves) owns 128 rows x 64 columns of the product at a time; wave w computes rows 32w..32w+31 of
// it as 2 x 4 MFMA tiles (64 accumulator VGPRs).  The k (feature) dimension advances in stages of 16 values: per stage
// the workgroup needs 128 x 16 values of Z (16 KiB) and 64 x 16 values of Theta (8 KiB) in LDS.  The stage sequence runs
// across tile boundaries (next column group / next row block).
// Global -> LDS goes through the LDS-DMA path (global_load_lds_dwordx4: no staging registers, no ds_write pass): one
// instruction moves 64 16-byte pieces to 1 KiB of consecutive LDS ("chunk" = 8 rows x the 128-byte line of the stage),
// each wave issues four Z chunks and two Theta chunks per stage.  Z -- streamed once per column group from HBM / the
// infinity cache, a microsecond away under load -- is requested TWO stages ahead into a ring of three 16 KiB buffers,
// Theta (L2 resident) one stage ahead into two 8 KiB buffers; a stage ends with a counted s_waitcnt vmcnt (the four Z
// instructions of the stage after next stay in flight) and a bare s_barrier.  (With register staging -- loads into 24
// VGPRs during one stage's MFMAs, six ds_write_b128 before the barrier -- the prefetch distance was one stage, about the
// memory latency itself: fetch + park cost a fifth of the kernel's time.)
// The DMA writes lane l's piece at chunk + 16 l, so the LDS image is shaped by which global piece a lane asks for: lane l
// loads piece q = ((l & 7) - 2 ((r >> 1) & 3)) & 7 of chunk row r = l >> 3, i.e. piece q of row r sits in 16-byte slot
//   r * 8 + ((q + 2 ((r >> 1) & 3)) & 7)
// of its chunk: rows stay 128-byte lines (the eight lanes of a row read one full line of global memory) and the rotation
// by two slots per row pair makes every ds_read_b128 of the compute loop conflict-free.  ds_read_b128 is serviced in
// four fixed 16-lane groups that mix two k-slots -- {0-3, 12-15 of lk 0 with 4-11 of lk 1}, ... (MI355X_MICROARCH.md, LDS):
//   natural operand (lane (li, lk) reads piece 4 u + lk of row li): per group rows {0-3, 12-15} at one piece and {4-11}
//     at the next -> slot offsets {0, 2, 5, 7, 4, 6, 1, 3} in the lower and the upper half of the 256-byte bank row;
//   replicated operand (lane reads piece 4 u + lk of row 4 r' + (lane & 3)): eight distinct pieces per group (the other
//     lanes broadcast), two per row, on eight different slots.
// Per lane every address is one of two bases (step u even / odd: the rotated slot index flips bit 2, base ^ 64) plus an
// immediate offset.  Which matrix is natural and which replicated depends on the orientation (TRP).
// Pieces beyond D, and rows / columns beyond the matrix, are loaded from a clamped address and overwritten with zeros
// by the lane that requested them, after the wait and before the barrier -- only in the last stage of a k range or in a
// tile that hangs over the edge.  Rows that are not 16-byte aligned (ALIGNED = false: odd leading dimension or base)
// cannot use the 16-byte DMA: that instantiation keeps the register staging (8-byte loads one stage ahead, masked and
// written to the same LDS image).
#define PJ_KC 16
#define PJ_ROWS 128
#define PJ_ZBYTES (PJ_ROWS * PJ_KC * 8)
// NCT = 16-column tiles per wave: 4 (a 128 x 64 workgroup tile) or 8 (128 x 128: two thirds of the global -> LDS traffic
// and half the barriers per flop, 64 more accumulator VGPRs; the Z ring is then two deep so that two workgroups and the
// column-sum accumulators still fit the CU's 160 KiB)
#define PJ_COLS(NCT) (16 * (NCT))
#define PJ_TBYTES(NCT) (PJ_COLS(NCT) * PJ_KC * 8)
#define PJ_ZRING(NCT) ((NCT) == 4 ? 3 : 2)
#define PJ_TRING 2
#define PJ_TBASE(NCT) (PJ_ZRING(NCT) * PJ_ZBYTES)
#define PJ_STAGING_BYTES(NCT) (PJ_TBASE(NCT) + PJ_TRING * PJ_TBYTES(NCT))

typedef double pv2d __attribute__((ext_vector_type(2)));

// One LDS-DMA request: 64 lanes x 16 bytes from each lane's own global address to lds_dst + 16 * lane (lds_dst: wave-
// uniform LDS byte address, goes through M0).  Inline asm, not __builtin_amdgcn_global_load_lds: with the builtin in
// flight hipcc (ROCm 7.2) turns every counted lgkmcnt wait of the compute loop into lgkmcnt(0), which serialises the
// operand double-buffering.  The request is invisible to the compiler's s_waitcnt bookkeeping: the kernel counts vmcnt
// itself (its own waits for ordinary loads can only come out stricter than needed: requests complete in issue order).
__device__ __forceinline__ void pj_glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// NumPy's arg-max order on correlations: NaN (0/0 of a zero vector) beats every number, the first one wins;
// otherwise larger value, then lower row (sparsevi.py:55 `corrs.argmax()`).
__device__ __forceinline__ bool corr_better(double a, long long ia, double b, long long ib) {
  const bool an = a != a, bn = b != b;
  if (an || bn) return an && (!bn || ia < ib);
  return a > b || (a == b && ia < ib);
}

// Zero-fill of a 16-byte piece {X[k], X[k+1]} beyond D and for rows that do not exist (register-staged path).  k is even.
__device__ __forceinline__ pv2d mask_piece(pv2d v, bool valid, int k, int D) {
  v.x = (valid && k < D) ? v.x : 0.0;
  v.y = (valid && k + 1 < D) ? v.y : 0.0;
  return v;
}

// one level of the transposed butterfly: keep b (a) if this lane's bit is set (clear), add the partner's other half
template <int CTRL> __device__ __forceinline__ double pj_fold(double a, double b, bool hi) {
  const double keep = hi ? b : a, send = hi ? a : b;
  const unsigned long long w = (unsigned long long)__double_as_longlong(send);
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)w, CTRL, 0xf, 0xf, true);
  const unsigned hi32 = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(w >> 32), CTRL, 0xf, 0xf, true);
  return keep + __longlong_as_double((long long)(((unsigned long long)hi32 << 32) | lo));
}

template <int CTRL> __device__ __forceinline__ double pj_foldm(double a, double b, bool hi) {   // the same exchange, multiplying
  const double keep = hi ? b : a, send = hi ? a : b;
  const unsigned long long w = (unsigned long long)__double_as_longlong(send);
  const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)w, CTRL, 0xf, 0xf, true);
  const unsigned hi32 = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(w >> 32), CTRL, 0xf, 0xf, true);
  return keep * __longlong_as_double((long long)(((unsigned long long)hi32 << 32) | lo));
}

// lane rows (16 lanes each) r0 r1 r2 r3 of a and b  ->  a = (a.r0, b.r0, a.r2, b.r2), b = (a.r1, b.r1, a.r3, b.r3)
__device__ __forceinline__ void pj_swap16(double& a, double& b) {
  typedef unsigned pj_v2u __attribute__((ext_vector_type(2)));
  const unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
  const pj_v2u lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
  const pj_v2u hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  const unsigned al = lo.x, bl = lo.y, ah = hi.x, bh = hi.y;   // (element reads through named scalars: see scan.hip)
  a = __longlong_as_double((long long)(((unsigned long long)ah << 32) | al));
  b = __longlong_as_double((long long)(((unsigned long long)bh << 32) | bl));
}

struct PjPos {       // one stage of the workgroup's sequence: k stage s of column group cg of row block br
  int s, cg;
  int64_t br;
};

template <int FAM, int MODE, bool ALIGNED, int NCT>
__global__ __launch_bounds__(256, 2) void proj_kernel(ProjArgs p) {
  constexpr int COLS = PJ_COLS(NCT), TBYTES = PJ_TBYTES(NCT), TBASE = PJ_TBASE(NCT), ZRING = PJ_ZRING(NCT);
  constexpr int TCH = NCT / 2;               // Theta chunks (8 columns each) per wave and stage
  // ONE LDS object: staging rings | COLSUM: 4 x S column sums | SELECT: the four waves' candidates (a second __shared__
  // array makes hipcc drain the DMA queue before every LDS read)
  extern __shared__ __attribute__((aligned(16))) unsigned char pj_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const int S = p.S, D = p.D;
  const int ngc = (S + COLS - 1) / COLS;
  const int nst = (D + PJ_KC - 1) / PJ_KC;
  const int64_t nblk = (p.N + PJ_ROWS - 1) / PJ_ROWS;
  const double clin = (FAM == FAM_LINREG) ? -0.5 * log(2.0 * 3.14159265358979323846 * p.param) : 0.0;
  const double parg = (FAM == FAM_LINREG) ? 1.0 / (2.0 * p.param) : p.param;

  // Orientation of the wave's 32 rows x 64 columns (see the compute loop): COLSUM / SELECT compute the transposed product.
  // (WRITE on the 128-column tile runs transposed as well: two data rows per lane instead of eight -- see its epilogue)
  constexpr bool TRP = MODE != PMODE_WRITE || NCT == 8;

  // Block -> tile sequence.  Default: workgroup b takes row blocks b, b + gridDim, ... and walks their column groups
  // itself -- Z is then streamed from HBM once per column group (the 64 workgroups of an XCD push ~20 MB through its 4 MB
  // L2 between two passes of one of them).  With p.team = number of column groups (COLSUM and WRITE: independent tiles;
  // SELECT: the members leave partial row moments in p.part for select_combine_kernel)
  // the column groups of a row block go to workgroups that sit on the SAME XCD and run in step: blocks b and b + 8 share
  // an XCD (dispatch is round-robin over the 8 XCDs: a speed assumption, not a correctness one), team = (b / 8) / ngc,
  // member = column group = (b / 8) % ngc; the second reader of a Z line finds it in L2.
  const bool teamed = p.team > 1;
  int cg0 = 0;
  int64_t br0 = blockIdx.x, brstep = gridDim.x;
  if (teamed) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = (int)(gridDim.x >> 3) / ngc;   // teams per XCD
    cg0 = j % ngc;
    br0 = j < per * ngc ? (int64_t)xcd * per + j / ngc : nblk;      // the (gridDim / 8) % ngc left-over workgroups of an XCD idle
    brstep = 8 * per;
  }
  // COLSUM: per-wave column-sum accumulators in LDS -- S of them, or only the COLS of the workgroup's own column group
  // when it is a member of a team (S = 1024 then costs 4 KiB instead of 32: two workgroups per CU stay resident)
  const int cacc_n = teamed ? COLS : S, cacc_0 = teamed ? cg0 * COLS : 0;
  double* colacc = (double*)(pj_lds + PJ_STAGING_BYTES(NCT)) + (size_t)wave * cacc_n;
  if (MODE == PMODE_COLSUM) {
    for (int c = lane; c < cacc_n; c += 64) colacc[c] = 0.0;
  }
  // the likelihood's tables (proj_math.h), behind the accumulators; copied before the first LDS-DMA request is issued, so
  // the compiler's wait for these ordinary loads is exact; the barrier that ends the first stage publishes them
  constexpr int NTAB = FAM == FAM_POISSON ? PJT_DOUBLES : FAM == FAM_LOGISTIC ? PJT_DOUBLES_LOGISTIC : 0;
  unsigned char* const tab_lds = pj_lds + PJ_STAGING_BYTES(NCT) + (MODE == PMODE_COLSUM ? (size_t)4 * cacc_n * sizeof(double) : 0);
  const pj_tab_t tab = (pj_tab_t)tab_lds;
  if (NTAB) {
    for (int k = tid; k < NTAB; k += 256) ((double*)tab_lds)[k] = p.tab[k];
  }
  if (br0 >= nblk) {
    if (MODE == PMODE_COLSUM) {
      __syncthreads();
      double* outp = p.colpart + (size_t)blockIdx.x * S;
      for (int c = tid; c < S; c += blockDim.x) outp[c] = 0.0;
    }
    return;
  }
  // ---- the prefetch stream -------------------------------------------------------------------------------------
  // this lane's share of a stage: slot `lane` of Z chunks 4 wave + j (rows 32 wave + 8 j + fr) and of Theta chunks
  // 2 wave + j (columns 16 wave + 8 j + fr), piece fq of the row's 128-byte line
  const int fr = lane >> 3, fq = ((lane & 7) - 2 * ((fr >> 1) & 3)) & 7;
  // Request state.  What a lane asks for is, per operand, row 8 j + fr of a tile = 8 j ldz (8 j ldt) elements behind row fr,
  // clamped to the matrix's last row, from the row block's own base pointer (a scalar): 32-bit element offsets, one VGPR per
  // request (rounds 3-4 carried four row POINTERS: 8 VGPRs, and the three 128-column instantiations at the 256-VGPR limit
  // parked up to 24 registers in scratch around the k loop).  LEAN (off: with the 16x16x4 loop's smaller operand buffers
  // every instantiation fits without it; measured ~1-2 % slower per call where it was on) carries only the offset of row
  // fr and forms the others -- stride, clamp, the lane's piece offset -- where a request is issued.
  constexpr bool LEAN = false;
  int zpv[LEAN ? 1 : 4], tpv[LEAN ? 1 : TCH];
  int zmax = 0;
  const double* zbase = p.Z;
  const int zstep = 8 * (int)p.ldz, tstep = 8 * p.ldt, tmax = (S - 1) * p.ldt;
  auto zp = [&](int j) { return LEAN ? min(zpv[0] + j * zstep, zmax) : zpv[LEAN ? 0 : j]; };
  auto tp = [&](int j) { return LEAN ? min(tpv[0] + j * tstep, tmax) : tpv[LEAN ? 0 : j]; };
  int64_t zp_br = -1;
  int tp_cg = -1;
  const int kmax = ALIGNED ? ((D - 1) & ~1) : (D - 1);
  auto set_z = [&](int64_t fbr) {
    zp_br = fbr;
    zbase = p.Z + fbr * PJ_ROWS * p.ldz;
    zmax = (int)((p.N - 1 - fbr * PJ_ROWS) * p.ldz);      // (>= 0: the block's first row exists)
    const int z0 = (32 * wave + fr) * (int)p.ldz;
    if (LEAN) zpv[0] = z0;
    else {
#pragma unroll
      for (int j = 0; j < (LEAN ? 1 : 4); ++j) zpv[j] = min(z0 + j * zstep, zmax);
    }
  };
  auto set_t = [&](int fcg) {
    tp_cg = fcg;
    const int t0 = (fcg * COLS + 8 * TCH * wave + fr) * p.ldt;
    if (LEAN) tpv[0] = t0;
    else {
#pragma unroll
      for (int j = 0; j < (LEAN ? 1 : TCH); ++j) tpv[j] = min(t0 + j * tstep, tmax);
    }
  };
  auto advance = [&](PjPos a) {
    if (++a.s == nst) {
      a.s = 0;
      if (teamed) a.br += brstep;
      else if (++a.cg == ngc) { a.cg = 0; a.br += brstep; }
    }
    return a;
  };
  // every piece of the stage exists (no zero-fill needed): scalar
  auto plain = [&](const PjPos& a) {
    return a.s < nst - 1 && (a.br + 1) * PJ_ROWS <= p.N && (a.cg + 1) * COLS <= S;
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)pj_lds;   // LDS byte address of the staging area
  // LDS-DMA requests (ALIGNED).  zslot / tslot: ring slot.  The clamped piece index keeps every address inside its row.
  auto issue_z = [&](const PjPos& a, int zslot) {
    if (a.br != zp_br) set_z(a.br);
    const int kc = min(a.s * PJ_KC + 2 * fq, kmax);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pj_glds16(zbase + (zp(j) + kc), lds0 + (unsigned)(zslot * PJ_ZBYTES + (4 * wave + j) * 1024));
  };
  auto issue_t = [&](const PjPos& a, int tslot) {
    if (a.cg != tp_cg) set_t(a.cg);
    const int kc = min(a.s * PJ_KC + 2 * fq, kmax);
#pragma unroll
    for (int j = 0; j < TCH; ++j)
      pj_glds16(p.theta + (tp(j) + kc), lds0 + (unsigned)(TBASE + tslot * TBYTES + (TCH * wave + j) * 1024));
  };
  // zero-fill of this lane's own slots of a stage that is not plain (after its DMA has landed)
  auto zero_fill = [&](const PjPos& a, int zslot, int tslot) {
    const int k = a.s * PJ_KC + 2 * fq;
    const int zrows = (int)min((int64_t)PJ_ROWS, p.N - a.br * PJ_ROWS);      // rows of the block that exist (a scalar: the row test stays on 32 bits)
#pragma unroll
    for (int j = 0; j < 4 + TCH; ++j) {
      const bool valid = j < 4 ? 32 * wave + 8 * j + fr < zrows : a.cg * COLS + 8 * (TCH * wave + j - 4) + fr < S;
      unsigned char* dst = j < 4 ? pj_lds + zslot * PJ_ZBYTES + (4 * wave + j) * 1024 + lane * 16
                                 : pj_lds + TBASE + tslot * TBYTES + (TCH * wave + j - 4) * 1024 + lane * 16;
      if (!valid || k >= D) *(pv2d*)dst = (pv2d){0.0, 0.0};
      else if (k + 1 >= D) *(double*)(dst + 8) = 0.0;
    }
  };
  // register staging (!ALIGNED): 8-byte loads of the next stage during this stage's MFMAs, masked and written before the barrier
  pv2d sreg[ALIGNED ? 1 : 4 + TCH];
  auto fetch_regs = [&](const PjPos& a) {
    if (a.br != zp_br) set_z(a.br);
    if (a.cg != tp_cg) set_t(a.cg);
    const int k = a.s * PJ_KC + 2 * fq;
    const int k0 = min(k, D - 1), k1 = min(k + 1, D - 1);
#pragma unroll
    for (int j = 0; j < 4 + TCH; ++j) {
      const double* src = j < 4 ? zbase + zp(j) : p.theta + tp(j < 4 ? 0 : j - 4);
      sreg[ALIGNED ? 0 : j].x = src[k0];
      sreg[ALIGNED ? 0 : j].y = src[k1];
    }
  };
  auto park_regs = [&](const PjPos& a, int zslot, int tslot) {
    const int k = a.s * PJ_KC + 2 * fq;
    const int zrows = (int)min((int64_t)PJ_ROWS, p.N - a.br * PJ_ROWS);
#pragma unroll
    for (int j = 0; j < 4 + TCH; ++j) {
      const bool valid = j < 4 ? 32 * wave + 8 * j + fr < zrows : a.cg * COLS + 8 * (TCH * wave + j - 4) + fr < S;
      unsigned char* dst = j < 4 ? pj_lds + zslot * PJ_ZBYTES + (4 * wave + j) * 1024 + lane * 16
                                 : pj_lds + TBASE + tslot * TBYTES + (TCH * wave + j - 4) * 1024 + lane * 16;
      *(pv2d*)dst = mask_piece(sreg[ALIGNED ? 0 : j], valid, k, D);
    }
  };

  PjPos cur = {0, cg0, br0};
  PjPos n1 = advance(cur), n2 = advance(n1);
  int zs = 0, ts = 0;                        // ring slots of the current stage (Z: stage mod ZRING, Theta: stage mod 2)
  if (ALIGNED) {
    issue_z(cur, 0);
    issue_t(cur, 0);
    if (ZRING == 3 && n1.br < nblk) { issue_z(n1, 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!plain(cur)) zero_fill(cur, 0, 0);
  } else {
    fetch_regs(cur);
    park_regs(cur, 0, 0);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  // read-side bases of this lane inside a 16-row operand tile (2 chunks): natural piece (row li, k-slot lk) and
  // replicated piece (row lane & 3 of the strip, k-slot lk) of an even step; ^ 64 for an odd step
  const unsigned nat0 = (unsigned)((li >> 3) * 1024 + (li & 7) * 128 + ((lk + 2 * ((li >> 1) & 3)) & 7) * 16);
#ifdef PJ_USE_4X4X4
  const unsigned rep0 = (unsigned)((lane & 3) * 128 + ((lk + 2 * ((lane >> 1) & 1)) & 7) * 16);
#endif
  // Orientation of the wave's 32 rows x 64 columns on the MFMA tiles.  WRITE: A = Z, B = Theta -- a lane holds 8 rows
  // x 4 columns, the 16 lanes of a DPP row hold 16 consecutive columns of one data row (128-byte stores).
  // COLSUM / SELECT: A = Theta, B = Z (the transposed product) -- a lane holds only 2 data rows (li, li + 16) x 16
  // columns, so the per-row state of the epilogue (response, shift, three moments) is 2 values per lane instead
  // of 8: what makes the kernel fit 256 registers at two waves per SIMD.
  pv4d acc[2][NCT];                  // [row tile][column tile]
  double yv[TRP ? 2 : 8], cp[TRP ? 2 : 8];
  double piv[2];
  while (true) {
    const int s = cur.s, cg = cur.cg;
    const int64_t br = cur.br;
    const bool more = n1.br < nblk, more2 = n2.br < nblk;
    const bool last = s == nst - 1;
    const int zs1 = zs == ZRING - 1 ? 0 : zs + 1, zs2 = zs1 == ZRING - 1 ? 0 : zs1 + 1, ts1 = ts ^ 1;
    // requests for the stages ahead fly while this stage's MFMAs run
    int kts = 0, kzs = 0;                      // stage bases (scalars); LEAN: the lane's piece offset is added where a request is formed,
    int kct = 0, kcz = 0;                      // otherwise it is carried through the stage
    if (ALIGNED) {
      // (the requests themselves are spread over the first groups of the compute loop)
      if (more) { if (n1.cg != tp_cg) set_t(n1.cg); kts = n1.s * PJ_KC; if (!LEAN) kct = min(kts + 2 * fq, kmax); }
      if (ZRING == 3 ? more2 : more) {
        const PjPos& zn = ZRING == 3 ? n2 : n1;
        if (zn.br != zp_br) set_z(zn.br);
        kzs = zn.s * PJ_KC;
        if (!LEAN) kcz = min(kzs + 2 * fq, kmax);
      }
    } else if (more) {
      fetch_regs(n1);
    }
    if (s == 0) {
#pragma unroll
      for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < NCT; ++tc) acc[tr][tc] = (pv4d){0.0, 0.0, 0.0, 0.0};
    }
    {
      // (-DPJ_USE_4X4X4, rounds 2-4; kept for A/B runs, tools/proj_ab.sh)  v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4
      // products per instruction (lanes 16k + 4b + i hold A_b[i][k], lanes 16k + 4b + j hold B_b[k][j], D_b[i][j] comes back
      // in lane 16i + 4b + j).  A register-only probe put it at 72 TFLOP/s and the 16x16x4 form at 47.6
      // (tools/probe/mfma_f64_peak.hip), so the 16x16 tile was built from it:
      // the "B" operand is a natural 16-wide tile (block b = its columns 4b .. 4b+3), the "A" operand four rows replicated
      // into all four blocks -- D is then a 4 x 16 strip, and register r of the old 16x16 accumulator IS the strip of rows
      // 4r .. 4r+3, so accumulator layout and epilogue are those of the 16x16x4 version.
      const unsigned char* zb = pj_lds + zs * PJ_ZBYTES + wave * 4096;       // this wave's two Z tiles
      const unsigned char* tb = pj_lds + TBASE + ts * TBYTES;
      const unsigned char* natb[2] = {(TRP ? zb : tb) + nat0, (TRP ? zb : tb) + (nat0 ^ 64u)};
#ifdef PJ_USE_4X4X4
      const unsigned char* repb[2] = {(TRP ? tb : zb) + rep0, (TRP ? tb : zb) + (rep0 ^ 64u)};
#endif
      // Operand registers are double-buffered by hand: the four replicated pieces of group g + 1 are requested before the
      // 16 MFMAs of group g are issued (an LDS read takes ~130 cycles, four MFMAs 64), and within a group the .x MFMAs of
      // all accumulators precede their dependent .y MFMAs.  A group = one column tile (TRP) / one row tile (WRITE) of one
      // 8-k step; NG groups per stage.
      constexpr int NU = PJ_KC / 8, NA = TRP ? NCT : 2, NB = TRP ? 2 : NCT, NG = NA * NU;
      auto nat_ptr = [&](int u, int t) {      // natural operand tile t of 8-k step u
        return (const pv2d*)(natb[u & 1] + t * 2048);
      };
#ifdef PJ_USE_4X4X4
      auto rep_ptr = [&](int g, int r) {      // replicated operand, group g = (u, tile), rows 4 r .. 4 r + 3
        const int u = g / NA, t = g % NA;
        return (const pv2d*)(repb[(u + r) & 1] + t * 2048 + (r >> 1) * 1024 + (r & 1) * 512);
      };
#endif
#ifndef PJ_USE_4X4X4
      // Round 5: v_mfma_f64_16x16x4_f64 with BOTH operands natural.  The 4x4x4 form below was chosen in round 2 on a probe that
      // put the 16x16x4 form at 47.6 TFLOP/s; the Gram kernel of csrc/gram.hip runs it at 0.78-0.87 of the 78.6 TFLOP/s peak
      // in its stage loop and a vendor GEMM in the round-4 traces at 0.93 MFMA-busy, so that probe was wrong.  The
      // accumulator layout is the same (register r of lane row lk = row lk + 4 r of the A tile, column li of the B tile), so
      // the epilogues do not change; the "A" tile is simply read like the "B" tile (one ds_read_b128 per tile and 8-value
      // step instead of four replicated ones: 6 instead of 36 LDS reads per 32 x 64 wave tile and stage), and its double
      // buffer shrinks from 32 to 8 VGPRs.  Measured A/B at the configs[4] shard shape, all nine family x consumer pairs:
      // the same to 0.5 % (the kernel is not bound by its operand reads) -- kept for the registers and the simpler loop.
      const unsigned char* nata[2] = {(TRP ? tb : zb) + nat0, (TRP ? tb : zb) + (nat0 ^ 64u)};
      auto a_ptr = [&](int g) { return (const pv2d*)(nata[(g / NA) & 1] + (g % NA) * 2048); };
      pv2d ap[2];
      ap[0] = *a_ptr(0);
      pv2d nt[NB];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int u = g / NA, t = g % NA;
        if (t == 0) {
#pragma unroll
          for (int i = 0; i < NB; ++i) nt[i] = *nat_ptr(u, i);
        }
        if (g + 1 < NG) ap[(g + 1) & 1] = *a_ptr(g + 1);
        if (ALIGNED) {
          // this group's share of the LDS-DMA requests (Theta first: the stage-end wait counts on the order)
#pragma unroll
          for (int q = 0; q < TCH + 4; ++q) {
            if ((NG >= 8 ? q : q * NG / 8) != g) continue;
            if (q < TCH) { if (more) pj_glds16(p.theta + (tp(q) + (LEAN ? min(kts + 2 * fq, kmax) : kct)), lds0 + (unsigned)(TBASE + ts1 * TBYTES + (TCH * wave + q) * 1024)); }
            else if (ZRING == 3 ? more2 : more)
              pj_glds16(zbase + (zp(q - TCH) + (LEAN ? min(kzs + 2 * fq, kmax) : kcz)), lds0 + (unsigned)((ZRING == 3 ? zs2 : zs1) * PJ_ZBYTES + (4 * wave + q - TCH) * 1024));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          pv4d& a4 = TRP ? acc[i][t] : acc[t][i];
          a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[g & 1].x, nt[i].x, a4, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          pv4d& a4 = TRP ? acc[i][t] : acc[t][i];
          a4 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[g & 1].y, nt[i].y, a4, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      pv2d rp[2][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rp[0][r] = *rep_ptr(0, r);
      pv2d nt[NB];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int u = g / NA, t = g % NA;
        if (t == 0) {
#pragma unroll
          for (int i = 0; i < NB; ++i) nt[i] = *nat_ptr(u, i);
        }
        if (g + 1 < NG) {
#pragma unroll
          for (int r = 0; r < 4; ++r) rp[(g + 1) & 1][r] = *rep_ptr(g + 1, r);
        }
        if (ALIGNED) {
          // this group's share of the LDS-DMA requests (Theta first: the stage-end wait counts on the order)
#pragma unroll
          for (int q = 0; q < TCH + 4; ++q) {
            if ((NG >= 8 ? q : q * NG / 8) != g) continue;
            if (q < TCH) { if (more) pj_glds16(p.theta + (tp(q) + (LEAN ? min(kts + 2 * fq, kmax) : kct)), lds0 + (unsigned)(TBASE + ts1 * TBYTES + (TCH * wave + q) * 1024)); }
            else if (ZRING == 3 ? more2 : more)
              pj_glds16(zbase + (zp(q - TCH) + (LEAN ? min(kzs + 2 * fq, kmax) : kcz)), lds0 + (unsigned)((ZRING == 3 ? zs2 : zs1) * PJ_ZBYTES + (4 * wave + q - TCH) * 1024));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            pv4d& a4 = TRP ? acc[i][t] : acc[t][i];
            a4[r] = TRP ? __builtin_amdgcn_mfma_f64_4x4x4f64(rp[g & 1][r].x, nt[i].x, a4[r], 0, 0, 0)
                        : __builtin_amdgcn_mfma_f64_4x4x4f64(rp[g & 1][r].x, nt[i].x, a4[r], 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < NB; ++i) {
            pv4d& a4 = TRP ? acc[i][t] : acc[t][i];
            a4[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(rp[g & 1][r].y, nt[i].y, a4[r], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
    if (last) {
      // ---- epilogue of tile (br, cg).  f64 C/D layout: D[i = (lane >> 4) + 4 * reg][j = lane & 15] ---------------
      const int64_t r0 = br * PJ_ROWS + 32 * wave;
      const int rows_left = (int)min((int64_t)32, p.N - r0);      // rows of this wave's 32 that exist (<= 0: none)
      if constexpr (!TRP) {
        // i = data row (lk + 4 reg), j = column (li)
        if (cg == 0 || teamed) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int64_t row = r0 + 16 * (e >> 2) + lk + 4 * (e & 3);
            const double y = (p.ycol >= 0 && row < p.N) ? p.Z[row * p.ldz + p.ycol] : 0.0;
            yv[e] = y;
            if (FAM == FAM_POISSON) {
              const int yi = (int)y;
              const bool small_count = (double)yi == y && (unsigned)yi < (unsigned)PJT_NFACT;
              cp[e] = small_count ? tab[PJT_LFACT + (small_count ? yi : 0)] : pj_lgamma1p_call(y);
              __builtin_amdgcn_sched_barrier(0);   // one at a time
            } else if (FAM == FAM_LINREG) {
              cp[e] = fma(-(y * y), parg, clin);       // the row's own part of the likelihood (loglik)
              yv[e] = 2.0 * y;
            } else {
              cp[e] = clin;
            }
          }
        }
        // (the row means are formed by the centring pass from the stored values: no state crosses the column groups)
#pragma unroll
        for (int tc = 0; tc < NCT; ++tc) {
          const int col = cg * COLS + 16 * tc + li;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int64_t row = r0 + 16 * (e >> 2) + lk + 4 * (e & 3);
            if (col < S && row < p.N) p.out[row * p.ldo + col] = loglik<FAM>(acc[e >> 2][tc][e & 3], yv[e], parg, cp[e], tab);
          }
        }
      } else if constexpr (MODE == PMODE_WRITE) {
        // WRITE on the 128-column tile, transposed product: i = column (lk + 4 reg within column tile tc), j = data row (li
        // within row tile tr).  A lane holds 2 data rows x 64 columns: the per-row state (response, row constant, output
        // pointer) is 2 values per lane instead of 8 -- what lets the 128 accumulator registers of this tile fit.  A store
        // instruction (fixed tr, tc, reg) covers 16 rows x 4 consecutive columns (32-byte pieces); the four registers of a
        // column tile complete the rows' 128-byte lines back to back, and the write-back L2 hands HBM whole lines.
        double* orow[2];
        bool rok[2];
        double y2[2], c2[2];                     // (formed per tile: nothing of the epilogue is held across the k loop)
#pragma unroll
        for (int tr = 0; tr < 2; ++tr) {
          const int64_t row = r0 + 16 * tr + li;
          rok[tr] = row < p.N;
          const double y = (p.ycol >= 0 && rok[tr]) ? p.Z[row * p.ldz + p.ycol] : 0.0;
          y2[tr] = y;
          if (FAM == FAM_POISSON) {
            const int yi = (int)y;
            const bool small_count = (double)yi == y && (unsigned)yi < (unsigned)PJT_NFACT;
            c2[tr] = small_count ? tab[PJT_LFACT + (small_count ? yi : 0)] : pj_lgamma1p_call(y);
            __builtin_amdgcn_sched_barrier(0);   // one at a time
          } else if (FAM == FAM_LINREG) {
            c2[tr] = fma(-(y * y), parg, clin);       // the row's own part of the likelihood (loglik)
            y2[tr] = 2.0 * y;
          } else {
            c2[tr] = clin;
          }
          orow[tr] = p.out + (rok[tr] ? row : 0) * p.ldo + cg * COLS + lk;
        }
        if constexpr (FAM != FAM_LINREG) {
#pragma unroll
          for (int tc = 0; tc < NCT; ++tc) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const bool cvalid = cg * COLS + 16 * tc + lk + 4 * r < S;
#pragma unroll
              for (int tr = 0; tr < 2; ++tr) {
                const double v = loglik<FAM>(acc[tr][tc][r], y2[tr], parg, c2[tr], tab);
                if (cvalid && rok[tr]) orow[tr][16 * tc + 4 * r] = v;
              }
              // (transcendental epilogues: two values per scheduling region, as in the column sums of this tile)
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        } else {
          // 16-byte stores: lane rows lk and lk ^ 1 exchange one register each (v_permlane16_swap), after which a lane holds
          // two ADJACENT columns -- even lk: (lk + 4 r, lk + 4 r + 1), odd lk: (lk - 1 + 4 r2, lk + 4 r2) -- and a store
          // instruction covers 16 rows x 8 consecutive columns (64-byte pieces), half as many instructions.  On one box, N = 2M,
          // D = 301, S = 256: 56.2 TFLOP/s against 54.0 with 8-byte stores (59.3 with the stores left out; the 64-column tile
          // of round 3: 52).  The transcendental families measured no gain from it (four values per scheduling region) and keep
          // the 8-byte form.
          const bool pair_ok = (p.ldo & 1) == 0 && ((uintptr_t)p.out & 15) == 0;
          const bool odd = (lk & 1) != 0;
#pragma unroll
          for (int tc = 0; tc < NCT; ++tc) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int cbase = 16 * tc + 4 * (2 * h + (odd ? 1 : 0)) - (odd ? 1 : 0);     // relative to orow (which carries + lk)
              const bool c0ok = cg * COLS + lk + cbase < S, c1ok = cg * COLS + lk + cbase + 1 < S;
#pragma unroll
              for (int tr = 0; tr < 2; ++tr) {
                double a = loglik<FAM>(acc[tr][tc][2 * h], y2[tr], parg, c2[tr], tab);
                double b = loglik<FAM>(acc[tr][tc][2 * h + 1], y2[tr], parg, c2[tr], tab);
                if (pair_ok) {
                  pj_swap16(a, b);
                  if (rok[tr]) {
                    if (c1ok) *(pv2d*)(orow[tr] + cbase) = (pv2d){a, b};
                    else if (c0ok) orow[tr][cbase] = a;
                  }
                } else if (rok[tr]) {
                  if (cg * COLS + 16 * tc + lk + 8 * h < S) orow[tr][16 * tc + 8 * h] = a;
                  if (cg * COLS + 16 * tc + lk + 8 * h + 4 < S) orow[tr][16 * tc + 8 * h + 4] = b;
                }
              }
            }
          }
        }
      } else {
        // i = column (lk + 4 reg within column tile tc), j = data row (li within row tile tr)
        // SELECT: every column group of a row stands alone -- shift = the group's first column, moments about it -- and
        // leaves {shift, sum, sum of squares, dot with the residual} in p.part; select_combine_kernel forms the row's
        // correlation from the groups' records.  Nothing is carried from one column group to the next and the kernel has
        // no arg-max of its own: the per-row state held in registers across the MFMA loop (24 VGPRs) and the in-kernel
        // correlation code are what made the transcendental instantiations spill (logistic 56 VGPRs, Poisson 12).
        // COLSUM keeps its three loop-carried values (response, constant, shift).
        constexpr bool SEL = MODE == PMODE_SELECT;
        double syv[2], scp[2], spiv[2], rs[2], rq[2], rd[2];
        // (the 128-column tile has no registers to carry the response across the k loop either: it re-reads it per tile)
        constexpr bool LOCAL = SEL || NCT > 4;
        double* const yq = LOCAL ? syv : yv;
        double* const cq = LOCAL ? scp : cp;
        double* const pq = LOCAL ? spiv : piv;
        if (LOCAL || cg == 0 || teamed) {
#pragma unroll
          for (int tr = 0; tr < 2; ++tr) {
            // (the row test on 32-bit numbers -- rows left in the matrix from this wave's first one, a scalar -- so that no 64-bit
            // lane value has to live across the k loop)
            const int rl = 16 * tr + li;
            const double y = (p.ycol >= 0 && rl < rows_left) ? p.Z[(r0 + rl) * p.ldz + p.ycol] : 0.0;
            yq[tr] = (FAM == FAM_LINREG) ? 2.0 * y : y;      // (loglik_shifted)
            cq[tr] = (FAM == FAM_POISSON) ? 0.0 : clin;   // (Poisson: gammaln(y + 1) is constant along the row and cancels in value - shift)
            rs[tr] = 0.0; rq[tr] = 0.0; rd[tr] = 0.0;
          }
        }
        {
#pragma unroll
          for (int tr = 0; tr < 2; ++tr) {
            // per-row shift: the row's value in the group's first column (lane group lk == 0, register 0 of column tile
            // 0), handed to the four lanes that share the row.  Sums, squares and dot products are accumulated on
            // (ll - shift): the one-pass moments then cancel on the scale of the row's SPREAD, not of |ll| (rows with
            // |mean| >> spread: a concentrated posterior, saturated logistic rows) -- the accuracy of the reference's
            // centre-then-norm order (sparsevi.py:49-51) without a second pass.
            // COLSUM only needs SOME shift that is constant along the row (the centring correction removes it); its
            // column groups may sit in different workgroups, so it takes one every workgroup can form from the row
            // alone: the likelihood at a zero linear predictor.
            // (logistic: -log 2 for every row; Poisson: y log(log 2) - log 2 -- formed per tile, nothing held across the k loop)
            if (MODE == PMODE_COLSUM && FAM == FAM_LOGISTIC) pq[tr] = -0.693147180559945309;
            else if (MODE == PMODE_COLSUM && FAM == FAM_POISSON) pq[tr] = fma(yq[tr], -0.366512920581664327, -0.693147180559945309);
            else if (MODE == PMODE_COLSUM) pq[tr] = 0.0;            // (linear regression: closed form of value - shift, loglik_shifted)
            else {
              const double m0 = acc[tr][0][0];
              const double l0 = (FAM == FAM_LINREG) ? (yq[tr] - m0) * m0 : loglik<FAM>(m0, yq[tr], parg, cq[tr], tab);
              pq[tr] = __shfl(l0, li, BCX_WAVE);
            }
          }
        }
        // 64 columns at a time, fenced: with all NCT column tiles in one scheduling region the compiler keeps every
        // likelihood value of the tile live at once (the 128 x 128 tile then spills)
        // Logistic column sums: sum_n -log1p(u_n) = -log prod_n (1 + u_n), u = exp(-|m|) <= 1 -- the 32 rows of the wave's
        // tile share ONE logarithm per column instead of 32 log1p series: a lane multiplies the factors of its two rows,
        // the lanes of a DPP row multiply theirs in the same transposed butterfly that adds the linear parts
        // (min(m, 0) - shift), and the lane that ends up with a column takes log(product) <= log 2^32.  Per value:
        // exp + two operations instead of exp + log1p (24 -> ~16 fp64 instructions; they share the MFMA datapath).  The
        // product of 32 factors in [1, 2] carries a relative error of a few ulp, i.e. an absolute error of its logarithm
        // of ~1e-15 against a sum of 32 terms of up to log 2 -- tighter than adding 32 rounded log1p values.
        constexpr bool PRODF = MODE == PMODE_COLSUM && FAM == FAM_LOGISTIC;
#pragma unroll
        for (int h = 0; h < NCT / 4; ++h) {
          double cs[16];               // COLSUM: this lane's two rows of column (tc, r) of the half, index 4 tc + r
          double cpd[PRODF ? 16 : 1];  // logistic COLSUM: the product of (1 + u) over the lane's two rows
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) {
            const int tc = 4 * h + t4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int col = cg * COLS + 16 * tc + lk + 4 * r;
              const bool cvalid = col < S;
              const double rsd = (MODE == PMODE_SELECT && cvalid) ? p.resid[col] : 0.0;
              double csum = 0.0, cprod = 1.0;
#pragma unroll
              for (int tr = 0; tr < 2; ++tr) {
                const bool ok = cvalid && r0 + 16 * tr + li < p.N;
                if (PRODF) {
                  const double m = acc[tr][tc][r];
                  const double lin = (pjm_hi(m) < 0 ? m : 0.0) - pq[tr];           // min(m, 0) - shift
                  const double f = 1.0 + pjm_exp_nonpos(-fabs(m), tab);
                  csum += ok ? lin : 0.0;
                  cprod *= ok ? f : 1.0;
                } else {
                  const double v = ok ? loglik_shifted<FAM, MODE>(acc[tr][tc][r], yq[tr], parg, cq[tr], pq[tr], tab) : 0.0;
                  if (MODE == PMODE_COLSUM) csum += v;
                  else { rs[tr] += v; rq[tr] += v * v; rd[tr] += v * rsd; }
                }
              }
              cs[4 * t4 + r] = csum;
              if (PRODF) cpd[4 * t4 + r] = cprod;
              // the 128-column tile with a transcendental epilogue: two likelihood values (one column, the lane's two rows)
              // per scheduling region -- given more, the compiler interleaves many evaluations and runs out of registers.
              // (SELECT on the 64-column tile used to be fenced the same way; with the table forms it is not, and went
              // from 39.9 to 46.1 TFLOP/s for the logistic family.)
              if (NCT > 4 && FAM != FAM_LINREG) __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (MODE == PMODE_COLSUM) {
            // Sum over the 16 lanes (data rows) of a DPP row, all 16 columns at once: a transposed butterfly -- at each
            // level a lane keeps the half of its values whose index bit matches its lane bit and adds the partner's
            // partial sums of that half (partners: mirror, half mirror, xor 2, xor 1 -- the lane bit flips each time),
            // so lane li ends with the total of column index li: 15 exchanges instead of 64, then ONE LDS update per
            // lane instead of 16 dependent read-add-write round trips.
            double c8[8], c4[4], c2[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) c8[e] = pj_fold<0x140>(cs[e], cs[e + 8], (li & 8) != 0);
#pragma unroll
            for (int e = 0; e < 4; ++e) c4[e] = pj_fold<0x141>(c8[e], c8[e + 4], (li & 4) != 0);
#pragma unroll
            for (int e = 0; e < 2; ++e) c2[e] = pj_fold<0x4E>(c4[e], c4[e + 2], (li & 2) != 0);
            double tot = pj_fold<0xB1>(c2[0], c2[1], (li & 1) != 0);
            if (PRODF) {
              double p8[8], p4[4], p2[2];
#pragma unroll
              for (int e = 0; e < 8; ++e) p8[e] = pj_foldm<0x140>(cpd[e], cpd[PRODF ? e + 8 : 0], (li & 8) != 0);
#pragma unroll
              for (int e = 0; e < 4; ++e) p4[e] = pj_foldm<0x141>(p8[e], p8[e + 4], (li & 4) != 0);
#pragma unroll
              for (int e = 0; e < 2; ++e) p2[e] = pj_foldm<0x4E>(p4[e], p4[e + 2], (li & 2) != 0);
              const double ptot = pj_foldm<0xB1>(p2[0], p2[1], (li & 1) != 0);       // in [1, 2^32] (or NaN, as the sum would be)
              tot -= pjm_log_pos(ptot, tab);
            }
            const int col = cg * COLS + 64 * h + 16 * (li >> 2) + lk + 4 * (li & 3);
            if (col < S) colacc[col - cacc_0] += (FAM == FAM_LINREG) ? tot * parg : tot;
          }
          if (NCT > 4) __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == PMODE_SELECT) {
#pragma unroll
          for (int tr = 0; tr < 2; ++tr) {
            // the four lane groups hold disjoint columns of the same row
            double s1 = rs[tr], s2 = rq[tr], sd = rd[tr];
            s1 += bcx_xor16_f64(s1); s1 += bcx_xor32_f64(s1);
            s2 += bcx_xor16_f64(s2); s2 += bcx_xor32_f64(s2);
            sd += bcx_xor16_f64(sd); sd += bcx_xor32_f64(sd);
            const int rl = 16 * tr + li;
            // this column group's share of the row's moments, about the group's own shift (its first column)
            if (lk == 0 && rl < rows_left) {
              double* rec = p.part + ((size_t)cg * (size_t)p.N + (size_t)(r0 + rl)) * 4;
              *(pv2d*)rec = (pv2d){pq[tr], s1};
              *(pv2d*)(rec + 2) = (pv2d){s2, sd};
            }
          }
        }
      }
    }
    if (!more) break;
    if (ALIGNED) {
      // everything but the Z requests of the stage after next has to have landed (requests complete in issue order;
      // after WRITE's epilogue the queue also holds stores, which do not: drain it)
      if (ZRING == 3 && more2 && !(last && !TRP)) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!plain(n1)) zero_fill(n1, zs1, ts1);
    } else {
      park_regs(n1, zs1, ts1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    cur = n1; n1 = n2; n2 = advance(n2);
    zs = zs1; ts = ts1;
  }
  if (MODE == PMODE_COLSUM) {
    __syncthreads();
    double* outp = p.colpart + (size_t)blockIdx.x * S;
    const double* ca = (const double*)(pj_lds + PJ_STAGING_BYTES(NCT));
    for (int c = tid; c < S; c += blockDim.x) {
      const int k = c - cacc_0;                    // (columns of other team members: zero)
      outp[c] = (k >= 0 && k < cacc_n) ? ((ca[k] + ca[(size_t)cacc_n + k]) + ca[2 * (size_t)cacc_n + k]) + ca[3 * (size_t)cacc_n + k] : 0.0;
    }
  }
}

// out[n][:] -= mean(out[n][:])    (lls -= lls.mean(axis=1)[:, None], projector.py:21)
// One wave per row at a time: the row is read once (16-byte accesses when the rows allow them; up to 512 columns stay in
// registers between the sum and the subtraction, longer rows are re-read from L2), summed in a fixed order (lane-strided
// partial sums, then the wave butterfly) and written back.  (The projection kernel used to hand over row sums; forming
// them here costs no traffic and leaves WRITE without state that crosses its column groups.)
template <bool VEC>
__global__ __launch_bounds__(256) void center_kernel(double* out, int64_t ldo, int64_t N, int S) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t n = wave; n < N; n += nwaves) {
    double* row = out + n * ldo;
    if (VEC && S <= 512) {
      pv2d v[4];
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 2 * lane + 128 * t;
        v[t] = c < S ? *(const pv2d*)(row + c) : (pv2d){0.0, 0.0};
        acc += v[t].x + v[t].y;
      }
      const double m = wave_allsum(acc) / (double)S;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = 2 * lane + 128 * t;
        if (c < S) { v[t].x -= m; v[t].y -= m; *(pv2d*)(row + c) = v[t]; }
      }
    } else {
      double acc = 0.0;
      for (int c = lane; c < S; c += 64) acc += row[c];
      const double m = wave_allsum(acc) / (double)S;
      for (int c = lane; c < S; c += 64) row[c] -= m;
    }
  }
}

// out[n] = sum_c rows[n][c]^2: a wave per row (the zero-vector filter of the subsample branch, hilbert.py:19-22).
__global__ __launch_bounds__(256) void row_sumsq_kernel(const double* rows, int64_t ld, int64_t N, int S, double* out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
  for (int64_t n = wave; n < N; n += nwaves) {
    const double* row = rows + n * ld;
    double acc = 0.0;
    for (int c = lane; c < S; c += 64) acc = fma(row[c], row[c], acc);
    acc = wave_allsum(acc);
    if (lane == 0) out[n] = acc;
  }
}

// A handful of rows (the coreset points SparseVI projects at every ADAM step, sparsevi.py:38-39): one workgroup of sixteen
// waves per row, x_row in LDS.  A WAVE takes a sample at a time (four in flight): its lanes read the sample's parameter row
// as consecutive 16-byte pieces (D = 301: three loads per lane, 1 KiB contiguous per instruction -- a thread per sample walked
// 64 rows 2.4 KB apart with every load: 13 us), the partial products meet in a wave sum, one lane evaluates the likelihood.
// Row mean over the workgroup, centred values stored: one ~5 us launch where the tiled MFMA kernel (one 128-row block, D / 16
// barrier-separated stages) takes 35 us plus 5 us for the centring pass; the call sits on the critical path of every ADAM
// step.  Same arithmetic per value (loglik<FAM>); the dot product is summed lane-wise (values 2 lane + 128 i), then over lanes.
#define PJ_SMALL_ROWS 32
template <int FAM, bool AL>
__global__ __launch_bounds__(1024) void proj_small_kernel(ProjArgs p, int center) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pj_lds[];
  __shared__ double scratch[BCX_SCRATCH];
  __shared__ double svals[1024];                                       // S <= 1024
  double* xs = (double*)pj_lds;                                        // D (+1) doubles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, D = p.D, S = p.S;
  const int Dp = (D + 1) & ~1;
  constexpr int NTAB = FAM == FAM_POISSON ? PJT_DOUBLES : FAM == FAM_LOGISTIC ? PJT_DOUBLES_LOGISTIC : 0;
  double* tabw = xs + Dp;
  const pj_tab_t tab = (pj_tab_t)tabw;
  const int64_t row = blockIdx.x;
  const double* z = p.Z + row * p.ldz;
  for (int k = tid; k < Dp; k += 1024) xs[k] = k < D ? z[k] : 0.0;
  if (NTAB) for (int k = tid; k < NTAB; k += 1024) tabw[k] = p.tab[k];
  const double y = p.ycol >= 0 ? z[p.ycol] : 0.0;
  const double clin = (FAM == FAM_LINREG) ? -0.5 * log(2.0 * 3.14159265358979323846 * p.param) : 0.0;
  const double parg = (FAM == FAM_LINREG) ? 1.0 / (2.0 * p.param) : p.param;
  double yv = y, c0 = clin;
  if (FAM == FAM_POISSON) {
    const int yi = (int)y;
    const bool small_count = (double)yi == y && (unsigned)yi < (unsigned)PJT_NFACT;
    c0 = small_count ? p.tab[PJT_LFACT + (small_count ? yi : 0)] : pj_lgamma1p_call(y);
  } else if (FAM == FAM_LINREG) {
    c0 = fma(-(y * y), parg, clin);
    yv = 2.0 * y;
  }
  // the samples of this workgroup: all of them when the row is centred here, a slice (gridDim.y of them) when it is left raw
  const int per = ((S + (int)gridDim.y - 1) / (int)gridDim.y + 3) & ~3;
  const int s_begin = min(S, (int)blockIdx.y * per), s_end = min(S, s_begin + per);
  __syncthreads();
  for (int s0 = s_begin + 4 * wave; s0 < s_end; s0 += 64) {            // (wave-uniform trip count)
    double m[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c = 2 * lane; c < Dp; c += 128) {
      const pv2d x = *(const pv2d*)(xs + c);                           // (the pad column's x is 0; its theta may be anything: masked)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int sidx = s0 + q < s_end ? s0 + q : s_end - 1;
        const double* th = p.theta + (size_t)sidx * p.ldt + c;
        pv2d t;
        if (AL) t = *(const pv2d*)th;
        else { t.x = th[0]; t.y = c + 1 < D ? th[1] : 0.0; }
        if (c + 1 >= D) t.y = 0.0;
        m[q] = fma(x.x, t.x, m[q]);
        m[q] = fma(x.y, t.y, m[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = wave_allsum(m[q]);
    const double mine = lane == 0 ? m[0] : lane == 1 ? m[1] : lane == 2 ? m[2] : m[3];
    if (lane < 4 && s0 + lane < s_end) svals[s0 + lane - s_begin] = loglik<FAM>(mine, yv, parg, c0, tab);
  }
  __syncthreads();
  const int mine_n = s_end - s_begin;
  double part[1] = {tid < mine_n ? svals[tid] : 0.0};
  const double v = part[0];
  double mean = 0.0;
  if (center) {                                                        // (gridDim.y == 1)
    block_allsum<1>(part, scratch);
    mean = part[0] / (double)S;
  }
  if (tid < mine_n) p.out[row * p.ldo + s_begin + tid] = v - mean;
}

// ---- a few hundred rows (the coreset points of SparseVI once the coreset has grown: sparsevi.py:38-39 at k = 33 .. 4096) -------
// The tiled kernel above starts at 128-row blocks with D / 16 barrier-separated LDS stages: for 300 rows it runs three
// workgroups per column tile for 56 us, on the critical path of every ADAM step.  Here a workgroup forms a 32 x 32 block of
// Z Theta^T -- one 16 x 16 tile per wave on v_mfma_f64_16x16x4_f64 -- with both operands straight from global memory (they are
// L2-resident: k x D and S x D doubles): lane (i = lane % 16, g = lane / 16) feeds row i's values 8 g .. 8 g + 7 of each run
// of 32 of the inner dimension, one per MFMA step (any assignment of the inner index to steps serves, as long as both
// operands use the same one), i.e. four 16-byte loads per operand and run; three runs in flight.  9 us for 300 x 301 x 256.
// The inner index is summed in ANOTHER order than in the tiled kernel (last bits differ: with the tiled kernel's order and
// 8-byte loads it ran 14 us and still did not reproduce it), so only the *_points entries use it: for rows that every shard
// projects alike -- the coreset points -- while data rows keep the tiled kernel whatever their shard's size, and a row-sharded
// build reproduces the single-shard one bit for bit (tests/test_gpu_sharded.py).  Raw log-likelihoods (the caller centres).
#define PJ_MID_ROWS 4096
typedef double pjm4d __attribute__((ext_vector_type(4)));
template <int FAM, bool AL>
static __device__ __forceinline__ void proj_mid_body(const ProjArgs& p, const int bx, const int by) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pj_lds[];
  constexpr int NTAB = FAM == FAM_POISSON ? PJT_DOUBLES : FAM == FAM_LOGISTIC ? PJT_DOUBLES_LOGISTIC : 0;
  double* tabw = (double*)pj_lds;
  const pj_tab_t tab = (pj_tab_t)tabw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4, rb = wave >> 1, cb = wave & 1;
  const int D = p.D, S = p.S;
  if (NTAB) { for (int k = tid; k < NTAB; k += 256) tabw[k] = p.tab[k]; __syncthreads(); }
  const int64_t arow = (int64_t)bx * 32 + rb * 16 + li;
  const int bcol = by * 32 + cb * 16 + li;
  const double* zp = p.Z + (arow < p.N ? arow : 0) * p.ldz;
  const double* tp = p.theta + (size_t)(bcol < S ? bcol : 0) * p.ldt;
  const bool aok = arow < p.N, bok = bcol < S;
  pjm4d acc = (pjm4d){0.0, 0.0, 0.0, 0.0};
  auto fetch = [&](int kb, double (&xa)[8], double (&xb)[8]) {
    const int k0 = kb + 8 * lk;
    if (AL && k0 + 8 <= D) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const pv2d u = *(const pv2d*)(zp + k0 + 2 * q), v = *(const pv2d*)(tp + k0 + 2 * q);
        xa[2 * q] = u.x; xa[2 * q + 1] = u.y; xb[2 * q] = v.x; xb[2 * q + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const bool ok = k0 + q < D;
        xa[q] = ok ? zp[ok ? k0 + q : 0] : 0.0;
        xb[q] = ok ? tp[ok ? k0 + q : 0] : 0.0;
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
  double xa[3][8], xb[3][8];                       // three runs of the inner dimension in flight
  fetch(0, xa[0], xb[0]);
  if (32 < D) fetch(32, xa[1], xb[1]);
  for (int kb = 0; kb < D; kb += 96) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int kc = kb + 32 * u;
      if (kc < D) {
        if (kc + 64 < D) fetch(kc + 64, xa[(u + 2) % 3], xb[(u + 2) % 3]);
#pragma unroll
        for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[u][t], xb[u][t], acc, 0, 0, 0);
      }
    }
  }
  // accumulator layout: register r holds (row 16 rb + lane / 16 + 4 r, column 16 cb + lane % 16)
  const double clin = (FAM == FAM_LINREG) ? -0.5 * log(2.0 * 3.14159265358979323846 * p.param) : 0.0;
  const double parg = (FAM == FAM_LINREG) ? 1.0 / (2.0 * p.param) : p.param;
  const int col = by * 32 + cb * 16 + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t row = (int64_t)bx * 32 + rb * 16 + lk + 4 * r;
    if (row < p.N && col < S) {
      const double y = p.ycol >= 0 ? p.Z[row * p.ldz + p.ycol] : 0.0;
      double yv = y, c0 = clin;
      if (FAM == FAM_POISSON) {
        const int yi = (int)y;
        const bool small_count = (double)yi == y && (unsigned)yi < (unsigned)PJT_NFACT;
        c0 = small_count ? tab[PJT_LFACT + (small_count ? yi : 0)] : pj_lgamma1p_call(y);
      } else if (FAM == FAM_LINREG) {
        c0 = fma(-(y * y), parg, clin);
        yv = 2.0 * y;
      }
      p.out[row * p.ldo + col] = loglik<FAM>(acc[r], yv, parg, c0, tab);
    }
  }
}

template <int FAM, bool AL>
__global__ __launch_bounds__(256) void proj_mid_kernel(ProjArgs p) { proj_mid_body<FAM, AL>(p, blockIdx.x, blockIdx.y); }

// SparseVI's ADAM step needs, at fresh draws, the column sums of the whole data set -- in closed form for the linear-regression
// family (csrc/moments_quad.h) -- AND the projection of the coreset points.  Both only read the draws, and each is a chain of
// dependent round trips to memory on a fraction of the chip: ONE launch, the first nq workgroups the closed form, the rest
// the 32 x 32 blocks of the points' projection (16 + 9 us one after the other).  The same arithmetic as the separate launches.
template <bool ALP, bool ALQ>
__global__ __launch_bounds__(256) void proj_mid_quad_kernel(ProjArgs p, MqArgs q, int nq, int gx) {
  if ((int)blockIdx.x < nq) { moments_quad_body<ALQ>(q, blockIdx.x, nq); return; }
  const int b = blockIdx.x - nq;
  proj_mid_body<FAM_LINREG, ALP>(p, b % gx, b / gx);
}

// colsum[s] = sum over the workgroup partials in a fixed order: one workgroup per 64 columns, four
// partial-index segments per column combined 0..3 (8 independent loads in flight per thread).
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const double* part, int nparts, int S, double* colsum) {
  __shared__ double seg[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
  double acc = 0.0;
  if (c < S) {
    int b = sg;
    for (; b + 28 < nparts; b += 32) {
      double v[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = part[(size_t)(b + 4 * t) * S + c];
#pragma unroll
      for (int t = 0; t < 8; ++t) acc += v[t];
    }
    for (; b < nparts; b += 4) acc += part[(size_t)b * S + c];
  }
  seg[sg][threadIdx.x & 63] = acc;
  __syncthreads();
  if (sg == 0 && c < S) colsum[c] = ((seg[0][threadIdx.x] + seg[1][threadIdx.x]) + seg[2][threadIdx.x]) + seg[3][threadIdx.x];
}
// the centring correction  sum_n (ll[n][s] - mean_n) = raw[s] - (1/S) sum_s' raw[s']
__global__ __launch_bounds__(256) void colsum_center_kernel(int S, double* colsum) {
  __shared__ double scratch[BCX_SCRATCH];
  double tot[1] = {0.0};
  for (int c = threadIdx.x; c < S; c += blockDim.x) tot[0] += colsum[c];
  block_allsum<1>(tot, scratch);
  const double corr = tot[0] / (double)S;
  for (int c = threadIdx.x; c < S; c += blockDim.x) colsum[c] -= corr;
}

// SELECT on teams: the column groups of a row were processed by different workgroups, each leaving {shift p_g, s1_g, s2_g,
// sd_g} = its first column's value and the sums of (ll - p_g), (ll - p_g)^2, (ll - p_g) resid over its n_g columns.  With
// delta_g = mean - p_g (differences of close numbers: every p_g is a value of the row, so the cancellation stays on the scale
// of the row's spread):   mean - p_0 = sum_g (n_g (p_g - p_0) + s1_g) / S,
//   ||ll - mean||^2 = sum_g (s2_g - 2 delta_g s1_g + n_g delta_g^2),    (ll - mean) . resid = sum_g (sd_g - delta_g R_g)
// with R_g the group's residual sum -- the one-group case is the formula of the projection kernel's own epilogue.
__global__ __launch_bounds__(256) void select_combine_kernel(const double* part, int64_t N, int ngc, int cols, int S, const double* resid,
                                                             double* best_val, int64_t* best_idx) {
  __shared__ double Rg[64];
  __shared__ double sv[4];
  __shared__ long long si[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int g = wave; g < ngc; g += 4) {                 // residual sum of each column group, fixed order
    double a = 0.0;
    for (int c = g * cols + lane; c < min(S, (g + 1) * cols); c += 64) a += resid[c];
    a = wave_allsum(a);
    if (lane == 0) Rg[g] = a;
  }
  __syncthreads();
  double bestv = -INFINITY;
  long long besti = 0x7fffffffffffffffLL;
  for (int64_t row = (int64_t)blockIdx.x * 256 + tid; row < N; row += (int64_t)gridDim.x * 256) {
    const double p0 = part[(size_t)row * 4];
    double msum = 0.0;
    for (int g = 0; g < ngc; ++g) {
      const double* rec = part + ((size_t)g * (size_t)N + (size_t)row) * 4;
      msum += (double)min(cols, S - g * cols) * (rec[0] - p0) + rec[1];
    }
    const double mrel = msum / (double)S;               // mean - p_0
    double nrm2 = 0.0, dot = 0.0;
    for (int g = 0; g < ngc; ++g) {
      const double* rec = part + ((size_t)g * (size_t)N + (size_t)row) * 4;
      const double dl = mrel - (rec[0] - p0);
      nrm2 += rec[2] - 2.0 * dl * rec[1] + (double)min(cols, S - g * cols) * dl * dl;
      dot += rec[3] - dl * Rg[g];
    }
    const double corr = nrm2 > 0.0 ? dot / sqrt(nrm2) / (double)S : __builtin_nan("");
    if (corr_better(corr, row, bestv, besti)) { bestv = corr; besti = row; }
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const double ov = __shfl_xor(bestv, off, BCX_WAVE);
    const long long oi = __shfl_xor(besti, off, BCX_WAVE);
    if (corr_better(ov, oi, bestv, besti)) { bestv = ov; besti = oi; }
  }
  if (lane == 0) { sv[wave] = bestv; si[wave] = besti; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (corr_better(sv[w], si[w], bestv, besti)) { bestv = sv[w]; besti = si[w]; }
    best_val[blockIdx.x] = bestv;
    best_idx[blockIdx.x] = besti;
  }
}

__global__ __launch_bounds__(256) void select_final_kernel(const double* bv, const int64_t* bi, int nparts, double* out_val, int64_t* out_idx) {
  __shared__ double sv[256];
  __shared__ long long si[256];
  double v = -INFINITY; long long i = 0x7fffffffffffffffLL;
  for (int b = threadIdx.x; b < nparts; b += blockDim.x)
    if (corr_better(bv[b], bi[b], v, i)) { v = bv[b]; i = bi[b]; }
  sv[threadIdx.x] = v; si[threadIdx.x] = i;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int t = 1; t < 256; ++t)
      if (corr_better(sv[t], si[t], v, i)) { v = sv[t]; i = si[t]; }
    *out_val = v; *out_idx = i;
  }
}

// ---- C ABI --------------------------------------------------------------------------------------
static thread_local std::string g_proj_err;
extern "C" const char* bcx_project_last_error(void) { return g_proj_err.c_str(); }
void bcx_project_set_error(const std::string& msg) { g_proj_err = msg; }   // (moments.hip reports through the same string)

#define PROJ_HIP(call)                                                            \
  do {                                                                            \
    hipError_t _e = (call);                                                       \
    if (_e != hipSuccess) {                                                       \
      g_proj_err = std::string(#call) + ": " + hipGetErrorString(_e);             \
      return BCX_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

// Persistent launch: as many workgroups as are resident at once (2 per CU: 64 KiB of staging LDS each), every one
// striding over the 128-row blocks -- with more workgroups than that the last wave of blocks leaves most CUs idle.
static int proj_grid(int64_t N) {
  // resident workgroups of the CURRENT device (looked up per device id, once; races write the same value)
  static std::atomic<int> resident_of[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  int resident = resident_of[dev].load(std::memory_order_relaxed);
  if (!resident) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    resident = std::min(2 * cus, 2048);
    resident_of[dev].store(resident, std::memory_order_relaxed);
  }
  const int64_t blocks = (N + PJ_ROWS - 1) / PJ_ROWS;
  return (int)std::max<int64_t>(1, std::min<int64_t>(blocks, resident));
}

// The likelihood tables of proj_math.h in device memory: built once per device (from long double, on the host), never freed.
static const double* proj_tables() {
  static std::mutex mu;
  static double* tabs[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tabs[dev]) {
    double host[PJT_DOUBLES];
    pjm_fill_tables(host);
    double* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(host)) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (hipMemcpy(d, host, sizeof(host), hipMemcpyHostToDevice) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d); return nullptr; }
    tabs[dev] = d;
  }
  return tabs[dev];
}

// ---- measurement: hipEvents around the projection kernel alone, on the stream it runs on (bcx_project_profile) ----
namespace {
struct ProjProfile {
  bool on = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t used = 0;
  double ms = 0.0, flops = 0.0;
  int64_t launches = 0;
};
thread_local ProjProfile g_prof;
}  // namespace

template <int FAM, int MODE, int NCT, bool ALIGNED_ONLY = false> static int launch_one(bool aligned, dim3 grid, size_t shmem, hipStream_t st, const ProjArgs& p) {
  const bool timed = g_prof.on;
  if (timed) {
    if (g_prof.used == g_prof.ev.size()) {
      hipEvent_t a, b;
      PROJ_HIP(hipEventCreate(&a));
      PROJ_HIP(hipEventCreate(&b));
      g_prof.ev.emplace_back(a, b);
    }
    PROJ_HIP(hipEventRecord(g_prof.ev[g_prof.used].first, st));
  }
if (aligned) {
    PROJ_HIP(hipFuncSetAttribute((const void*)proj_kernel<FAM, MODE, true, NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL((proj_kernel<FAM, MODE, true, NCT>), grid, dim3(256), shmem, st, p);
  } else if constexpr (!ALIGNED_ONLY) {
    PROJ_HIP(hipFuncSetAttribute((const void*)proj_kernel<FAM, MODE, false, NCT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL((proj_kernel<FAM, MODE, false, NCT>), grid, dim3(256), shmem, st, p);
  }
  PROJ_HIP(hipGetLastError());
  if (timed) {
    PROJ_HIP(hipEventRecord(g_prof.ev[g_prof.used].second, st));
    g_prof.used++;
    g_prof.flops += 2.0 * (double)p.N * p.D * p.S;
  }
  return BCX_OK;
}

// on != 0: start timing every projection-kernel launch of this host thread; on == 0: stop.
extern "C" int bcx_project_profile(int32_t on) {
  g_prof.on = on != 0;
  if (on) { g_prof.used = 0; g_prof.ms = 0.0; g_prof.flops = 0.0; g_prof.launches = 0; }
  return BCX_OK;
}
// Totals since bcx_project_profile(1): kernel milliseconds (synchronises on the recorded events), launches and the
// algorithmic flops 2 N D S of those launches.
extern "C" int bcx_project_profile_read(double* ms_total, int64_t* launches, double* flops) {
  for (size_t i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    PROJ_HIP(hipEventSynchronize(g_prof.ev[i].second));
    PROJ_HIP(hipEventElapsedTime(&ms, g_prof.ev[i].first, g_prof.ev[i].second));
    g_prof.ms += ms;
    g_prof.launches++;
  }
  g_prof.used = 0;
  if (ms_total) *ms_total = g_prof.ms;
  if (launches) *launches = g_prof.launches;
  if (flops) *flops = g_prof.flops;
  return BCX_OK;
}

// Width of the workgroup tile: 128 columns (NCT = 8) for COLSUM of the linear-regression family when that does not add
// padded columns over 64-wide groups (54 against 50 TFLOP/s at the configs[4] shard shape).  Everything else keeps 64:
// SELECT's per-row moments and the transcendental epilogues (logistic, Poisson) spill with 128 accumulator VGPRs and
// measured slower (logistic D=300: 29 against 35 TFLOP/s), WRITE holds eight rows per lane.
static int proj_nct(int mode, int family, int S, bool aligned, int D) {
  // SELECT on the 128-column tile spilled ~200 registers in every family in round 2.  Since the row moments moved to
  // per-column-group records the linear-regression instantiation is the faster one: N = 5M, D = 301, S = 256
  // 13.80 against 15.08 ms per call (55.8 against 51.1 TFLOP/s); since round 5 (one request offset + scalar strides instead
  // of four row pointers and TCH offsets, row tests on 32-bit numbers) it fits 250 VGPRs without scratch (rounds 3-4: 24
  // VGPRs of request state parked in scratch around the loop).  The transcendental families stay at 64 columns
  // (logistic SELECT at 128: 231 spilled VGPRs).  BCX_PROJ_SEL_NCT=4 selects the 64-column tile (dev).
  static const bool sel8 = [] { const char* e = bcx_dev_env("BCX_PROJ_SEL_NCT"); return !(e && atoi(e) == 4); }();
  // (round 4, with the table forms: logistic / Poisson SELECT on the 128-column tile still spill 225 / 208 VGPRs and run at 16 /
  // 32 TFLOP/s against 48 / 45 on the 64-column tile.  Tried for the registers they would need: the residual in LDS instead
  // of per-column global loads (no change in the spills; Poisson + 1 %, linreg - 0.5 %), and recomputing the request pointers
  // at every stage instead of carrying them (frees 12 VGPRs and the parked ones of the other 128-column instantiations, but
  // costs 2-4 % in every instantiation).  Neither is in.)
  if (mode == PMODE_SELECT && family == FAM_LINREG && aligned && sel8) return (S + 127) / 128 * 128 == (S + 63) / 64 * 64 ? 8 : 4;
  // WRITE on the 128-column tile (round 4, transposed product: two data rows per lane; 16-byte aligned rows only): on one
  // box, N = 2M, S = 256, D = 300 / 301: linreg 56.3 against 51.5 TFLOP/s, logistic 51.6 / 48.6, Poisson 48.3 / 43.9.  Short
  // rows (D < 64) are bound by the stores of the N x S output, and there the 64-column tile's 128-byte pieces are the
  // better ones (logistic D = 10, S = 512: 1.22 against 1.25 ms).  BCX_PROJ_WRITE_NCT=4 selects the 64-column tile (dev).
  static const bool wr8 = [] { const char* e = bcx_dev_env("BCX_PROJ_WRITE_NCT"); return !(e && atoi(e) == 4); }();
  if (mode == PMODE_WRITE && aligned && wr8 && D >= 64) return (S + 127) / 128 * 128 == (S + 63) / 64 * 64 ? 8 : 4;
  if (mode != PMODE_COLSUM) return 4;
  // the 128-column tile exists for the column sums of the linear-regression family and, on 16-byte aligned rows, of the
  // logistic one (round 3: with the series' constants in scalar registers and 32-bit Theta offsets it fits 255 VGPRs;
  // +8 % at D = 300) and the Poisson one (round 4, below); the unaligned transcendental instantiations are not built
  // Poisson COLSUM on the 128-column tile (249 VGPRs, no scratch since round 5; it parked 16 VGPRs before) is the faster one
  // on 16-byte aligned rows: N = 2M, D = 301, S = 256 6.15 against 6.72 ms (50.1 against 45.9 TFLOP/s).  BCX_PROJ_POIS_NCT=4: dev.
  static const bool pois8 = [] { const char* e = bcx_dev_env("BCX_PROJ_POIS_NCT"); return !(e && atoi(e) == 4); }();
  if ((family == FAM_POISSON && !(pois8 && aligned)) || (family == FAM_LOGISTIC && !aligned)) return 4;
  static const int forced = [] { const char* e = bcx_dev_env("BCX_PROJ_NCT"); return e ? atoi(e) : 0; }();   // dev knob
  if (forced == 4 || forced == 8) return forced;
  return (S + 127) / 128 * 128 == (S + 63) / 64 * 64 ? 8 : 4;
}
// XCD teams (COLSUM, WRITE) need a grid that covers the 8 XCDs evenly; workgroups of an XCD that do not fill a team stay
// idle (at most a fifth of them).  Returns the team size = number of column groups, 0 for the one-workgroup walk.
static int proj_team(int mode, int family, int S, int grid, bool aligned, int D) {
  static const bool no_team = bcx_dev_env("BCX_PROJ_NO_TEAM") != nullptr;   // dev knob
  const int cols = 16 * proj_nct(mode, family, S, aligned, D), ngc = (S + cols - 1) / cols;
  return (!no_team && ngc > 1 && grid % 8 == 0 && grid / 8 >= 4 * ngc) ? ngc : 0;
}
// Grid and team size of one launch.  Large problems: the persistent grid of proj_grid, teams where they fit.  Small ones
// (all tiles resident at once: row blocks x column groups <= resident workgroups -- the few coreset points SparseVI
// projects beside the full data set at every ADAM step, sparsevi.py:35-41): one workgroup per TILE, as teams of one row
// block each, so the column groups of a row block run side by side instead of one workgroup walking them (k = 4 points,
// S = 256: 132 -> 35 us per call; that call sits on the critical path of every ADAM step).
static void proj_plan(int mode, int family, int64_t N, int S, bool aligned, int D, int* grid, int* team) {
  *grid = proj_grid(N);
  *team = proj_team(mode, family, S, *grid, aligned, D);
  static const bool no_team = bcx_dev_env("BCX_PROJ_NO_TEAM") != nullptr;   // dev knob
  const int cols = 16 * proj_nct(mode, family, S, aligned, D), ngc = (S + cols - 1) / cols;
  const int64_t nblk = (N + PJ_ROWS - 1) / PJ_ROWS;
  if (!no_team && *team == 0 && ngc > 1 && N > 0 && nblk * ngc <= 2 * 256) {
    const int per = (int)((nblk + 7) / 8);                     // teams (= row blocks) per XCD
    *grid = 8 * per * ngc;
    *team = ngc;
  }
}
static bool proj_aligned(const ProjArgs& p) {
  // 16-byte requests need 16-byte aligned rows: even leading dimensions and aligned bases (else 8-byte loads)
  return ((uintptr_t)p.Z % 16 == 0) && ((uintptr_t)p.theta % 16 == 0) && p.ldz % 2 == 0 && p.ldt % 2 == 0;
}
template <int MODE> static int launch_family(int family, dim3 grid, size_t extra_lds, hipStream_t st, const ProjArgs& p_in) {
  ProjArgs p = p_in;
  const int nct = proj_nct(MODE, family, p.S, proj_aligned(p), p.D);
  const size_t tab_bytes = family == FAM_POISSON ? PJT_BYTES(PJT_DOUBLES) : family == FAM_LOGISTIC ? PJT_BYTES(PJT_DOUBLES_LOGISTIC) : 0;
  const size_t shmem = (nct == 8 ? PJ_STAGING_BYTES(8) : PJ_STAGING_BYTES(4)) + extra_lds + tab_bytes;
  if (tab_bytes && !(p.tab = proj_tables())) { g_proj_err = "bcx_project: no device memory for the likelihood tables"; return BCX_ERR_NOMEM; }
  if (shmem > 160 * 1024) { g_proj_err = "bcx_project: S too large for the column-sum accumulators (S <= 3072)"; return BCX_ERR_ARG; }
  const bool aligned = proj_aligned(p);
  if constexpr (MODE == PMODE_SELECT) {
    if (nct == 8 && family == FAM_LINREG) return launch_one<FAM_LINREG, MODE, 8, true>(true, grid, shmem, st, p);
  }
  if constexpr (MODE == PMODE_WRITE) {
    if (nct == 8 && family == FAM_LINREG) return launch_one<FAM_LINREG, MODE, 8, true>(true, grid, shmem, st, p);
    if (nct == 8 && family == FAM_LOGISTIC) return launch_one<FAM_LOGISTIC, MODE, 8, true>(true, grid, shmem, st, p);
    if (nct == 8 && family == FAM_POISSON) return launch_one<FAM_POISSON, MODE, 8, true>(true, grid, shmem, st, p);
  }
  if constexpr (MODE == PMODE_COLSUM) {
    if (nct == 8 && family == FAM_LINREG) return launch_one<FAM_LINREG, MODE, 8>(aligned, grid, shmem, st, p);
    if (nct == 8 && family == FAM_LOGISTIC) return launch_one<FAM_LOGISTIC, MODE, 8, true>(true, grid, shmem, st, p);
    if (nct == 8 && family == FAM_POISSON) return launch_one<FAM_POISSON, MODE, 8, true>(true, grid, shmem, st, p);
  }
  switch (family) {
    case FAM_LOGISTIC: return launch_one<FAM_LOGISTIC, MODE, 4>(aligned, grid, shmem, st, p);
    case FAM_POISSON: return launch_one<FAM_POISSON, MODE, 4>(aligned, grid, shmem, st, p);
    case FAM_LINREG: return launch_one<FAM_LINREG, MODE, 4>(aligned, grid, shmem, st, p);
    default: g_proj_err = "unknown likelihood family"; return BCX_ERR_ARG;
  }
}

static int fill(ProjArgs& p, int family, const void* Z, int64_t N, int64_t ldz, int D, int ycol, const void* theta,
                int S, int ldt, double param) {
  if (!Z || !theta || N < 0 || D < 1 || S < 1 || ldz < D || ldt < D || S > 4096 || (int64_t)(S + 128) * ldt >= (int64_t)1 << 31) {
    // (the kernel forms Theta offsets (column) * ldt in 32 bits, columns up to S rounded to the tile width)
    g_proj_err = "bcx_project: bad arguments";
    return BCX_ERR_ARG;
  }
  if (family != FAM_LOGISTIC && (ycol < 0 || ycol >= ldz)) { g_proj_err = "bcx_project: response column required"; return BCX_ERR_ARG; }
  p.Z = (const double*)Z; p.theta = (const double*)theta; p.N = N; p.ldz = ldz; p.ldt = ldt; p.D = D; p.S = S;
  p.ycol = family == FAM_LOGISTIC ? -1 : ycol; p.param = param;
  p.out = nullptr; p.ldo = 0; p.rowsum = nullptr; p.colpart = nullptr; p.resid = nullptr; p.resid_sum = 0.0;
  p.best_val = nullptr; p.best_idx = nullptr; p.team = 0; p.part = nullptr; p.tab = nullptr;
  return BCX_OK;
}

// vecs (N x S) into out_dev, centred by a second pass over the rows (center != 0) or left as the raw log-likelihoods
static int project_write(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                         int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                         void* out_dev, int64_t ldo, void* rowsum_dev, bool center, bool points = false) {
  ProjArgs p;
  int rc = fill(p, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param);
  if (rc) return rc;
  if (!out_dev || ldo < S) { g_proj_err = "bcx_project_write: bad output"; return BCX_ERR_ARG; }
  if (N == 0) return BCX_OK;
  p.out = (double*)out_dev; p.ldo = ldo; p.rowsum = (double*)rowsum_dev;
  hipStream_t st = (hipStream_t)stream;
  static const bool no_small = bcx_dev_env("BCX_PROJ_NO_SMALL") != nullptr;    // dev: the tiled kernel for every N
  if (N <= PJ_SMALL_ROWS && S <= 1024 && D <= 4096 && !no_small) {
    // a handful of rows: one workgroup per row, likelihood and centring in the same launch (proj_small_kernel)
    const size_t tabd = family == FAM_POISSON ? PJT_DOUBLES : family == FAM_LOGISTIC ? PJT_DOUBLES_LOGISTIC : 0;
    if (tabd && !(p.tab = proj_tables())) { g_proj_err = "bcx_project: no device memory for the likelihood tables"; return BCX_ERR_NOMEM; }
    const size_t lds = ((size_t)((D + 1) & ~1) + tabd) * sizeof(double);
    const bool al = ((uintptr_t)p.theta % 16 == 0) && p.ldt % 2 == 0;
    const int cen = center ? 1 : 0;
    // raw rows need no sum over the samples: four workgroups per row, every wave then has all its loads in flight at once
    const dim3 sgrid((unsigned)N, center ? 1u : (unsigned)std::min(4, (S + 63) / 64));
#define PJ_SMALL(F)                                                                                                   \
    do {                                                                                                                \
      if (al) hipLaunchKernelGGL((proj_small_kernel<F, true>), sgrid, dim3(1024), lds, st, p, cen);                     \
      else hipLaunchKernelGGL((proj_small_kernel<F, false>), sgrid, dim3(1024), lds, st, p, cen);                      \
    } while (0)
    if (family == FAM_LOGISTIC) PJ_SMALL(FAM_LOGISTIC); else if (family == FAM_POISSON) PJ_SMALL(FAM_POISSON); else PJ_SMALL(FAM_LINREG);
#undef PJ_SMALL
    PROJ_HIP(hipGetLastError());
    return BCX_OK;
  }
  static const bool no_mid = bcx_dev_env("BCX_PROJ_NO_MID") != nullptr;        // dev: the tiled kernel beyond 32 rows
  if (points && N <= PJ_MID_ROWS && !no_mid) {
    // a few hundred rows that every shard projects alike (the coreset points): 32 x 32 blocks of the product straight from
    // L2-resident operands (proj_mid_kernel)
    const size_t tabd = family == FAM_POISSON ? PJT_DOUBLES : family == FAM_LOGISTIC ? PJT_DOUBLES_LOGISTIC : 0;
    if (tabd && !(p.tab = proj_tables())) { g_proj_err = "bcx_project: no device memory for the likelihood tables"; return BCX_ERR_NOMEM; }
    const bool al = ((uintptr_t)p.theta % 16 == 0) && p.ldt % 2 == 0 && ((uintptr_t)p.Z % 16 == 0) && p.ldz % 2 == 0;
    const dim3 mgrid((unsigned)((N + 31) / 32), (unsigned)((S + 31) / 32));
#define PJ_MID(F)                                                                                                     \
    do {                                                                                                                \
      if (al) hipLaunchKernelGGL((proj_mid_kernel<F, true>), mgrid, dim3(256), tabd * sizeof(double), st, p);           \
      else hipLaunchKernelGGL((proj_mid_kernel<F, false>), mgrid, dim3(256), tabd * sizeof(double), st, p);            \
    } while (0)
    if (family == FAM_LOGISTIC) PJ_MID(FAM_LOGISTIC); else if (family == FAM_POISSON) PJ_MID(FAM_POISSON); else PJ_MID(FAM_LINREG);
#undef PJ_MID
    PROJ_HIP(hipGetLastError());
  } else {
    int wgrid = 0;
    proj_plan(PMODE_WRITE, family, N, S, proj_aligned(p), D, &wgrid, &p.team);
    if ((rc = launch_family<PMODE_WRITE>(family, dim3(wgrid), 0, st, p))) return rc;
  }
  if (!center) return BCX_OK;
  const int g = (int)std::min<int64_t>((N + 3) / 4, 8192);
  if (S % 2 == 0 && ldo % 2 == 0 && (uintptr_t)p.out % 16 == 0)
    hipLaunchKernelGGL(center_kernel<true>, dim3(g), dim3(256), 0, st, p.out, ldo, N, S);
  else
    hipLaunchKernelGGL(center_kernel<false>, dim3(g), dim3(256), 0, st, p.out, ldo, N, S);
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}

// rowsum_dev (N doubles) is no longer used: the centring pass forms the means.
extern "C" int bcx_project_write(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                 int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                 void* out_dev, int64_t ldo, void* rowsum_dev) {
  return project_write(stream, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param, out_dev, ldo, rowsum_dev, true);
}
// The raw log-likelihoods loglik(z_n, theta_s), NOT centred: for a consumer that centres the rows itself while it reads
// them (bcx_load_rows_flags with BCX_LOAD_CENTER_ROWS) -- the N x S matrix is then written once and read once.
extern "C" int bcx_project_write_raw(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                     int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                     void* out_dev, int64_t ldo) {
  return project_write(stream, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param, out_dev, ldo, nullptr, false);
}

// The same two for the POINTS of a coreset (sparsevi.py:38-39): rows that every shard holds and projects alike, so the kernel
// may be chosen by their number -- up to 4096 of them take proj_mid_kernel (9 us where the tiled kernel takes 56 at 300 rows).
extern "C" int bcx_project_write_points(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                        int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                        void* out_dev, int64_t ldo, int32_t center) {
  return project_write(stream, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param, out_dev, ldo, nullptr, center != 0, true);
}

// bcx_project_colsum_moments_at and bcx_project_write_points (raw rows) at the same draws, as ONE launch when the points take
// the 32 x 32-block kernel (more than PJ_SMALL_ROWS and at most 4096 of them, linear-regression family); otherwise the two
// calls one after the other.  The results are those of the two calls, bit for bit.
extern "C" int bcx_project_points_colsum_moments(void* stream, const void* Zc_dev, int64_t Nc, int64_t ldzc, int32_t D, int32_t ycol,
                                                 const void* theta_dev, int32_t S, int32_t ldt, double sigsq, void* out_dev, int64_t ldo,
                                                 const void* M_dev, int64_t ldm, int32_t ycol_m, void* colsum_dev, void* work_dev,
                                                 const void* tbar_dev) {
  static const bool apart = bcx_dev_env("BCX_PROJ_NO_MERGE") != nullptr;      // dev: always the two launches
  const bool one = !apart && tbar_dev && Nc > PJ_SMALL_ROWS && Nc <= PJ_MID_ROWS && bcx_dev_env("BCX_PROJ_NO_MID") == nullptr &&
                   bcx_dev_env("BCX_PROJ_NO_SMALL") == nullptr;
  if (!one) {
    const int rc = bcx_project_colsum_moments_at(stream, M_dev, ldm, D, ycol_m, theta_dev, S, ldt, sigsq, colsum_dev, work_dev, tbar_dev);
    if (rc) return rc;
    return bcx_project_write_points(stream, FAM_LINREG, Zc_dev, Nc, ldzc, D, ycol, theta_dev, S, ldt, sigsq, out_dev, ldo, 0);
  }
  ProjArgs p;
  int rc = fill(p, FAM_LINREG, Zc_dev, Nc, ldzc, D, ycol, theta_dev, S, ldt, sigsq);
  if (rc) return rc;
  if (!out_dev || ldo < S) { g_proj_err = "bcx_project_points_colsum_moments: bad output"; return BCX_ERR_ARG; }
  if (!M_dev || !colsum_dev || !work_dev || D >= 1024 || ycol_m < 0 || ycol_m >= ldm || ldm < D) {
    g_proj_err = "bcx_project_points_colsum_moments: bad arguments (the moments of bcx_project_moments, D < 1024)";
    return BCX_ERR_ARG;
  }
  p.out = (double*)out_dev; p.ldo = ldo; p.rowsum = nullptr;
  MqArgs q;
  mq_plan(D, S, &q.nct, &q.Spad);
  q.M = (const double*)M_dev; q.ldm = ldm; q.D = D; q.ycol = ycol_m; q.theta = (const double*)theta_dev; q.S = S; q.ldt = ldt;
  q.tbar = (const double*)tbar_dev; q.sigsq = sigsq; q.colsum = (double*)colsum_dev; q.work = (double*)work_dev; q.dbg = 0;
  const int nq = q.nct * (q.Spad / 16);
  const int gx = (int)((Nc + 31) / 32), gy = (S + 31) / 32;
  const bool alq = ((uintptr_t)theta_dev % 16 == 0) && ldt % 2 == 0;
  const bool alp = alq && ((uintptr_t)p.Z % 16 == 0) && p.ldz % 2 == 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)(nq + gx * gy));
  if (alp) hipLaunchKernelGGL((proj_mid_quad_kernel<true, true>), grid, dim3(256), 0, st, p, q, nq, gx);
  else if (alq) hipLaunchKernelGGL((proj_mid_quad_kernel<false, true>), grid, dim3(256), 0, st, p, q, nq, gx);
  else hipLaunchKernelGGL((proj_mid_quad_kernel<false, false>), grid, dim3(256), 0, st, p, q, nq, gx);
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}

// colsum_dev[s] = sum_n vecs[n][s] without materialising vecs.  work_dev: 2048 * S doubles.
extern "C" int bcx_project_colsum(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                  int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                  void* colsum_dev, void* work_dev) {
  ProjArgs p;
  int rc = fill(p, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param);
  if (rc) return rc;
  if (!colsum_dev || !work_dev) { g_proj_err = "bcx_project_colsum: bad output"; return BCX_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  int grid = 0;
  p.colpart = (double*)work_dev;
  proj_plan(PMODE_COLSUM, family, N, S, proj_aligned(p), D, &grid, &p.team);
  const size_t cacc = p.team ? (size_t)16 * proj_nct(PMODE_COLSUM, family, S, proj_aligned(p), D) : (size_t)S;   // accumulators per wave
  if ((rc = launch_family<PMODE_COLSUM>(family, dim3(grid), 4 * cacc * sizeof(double), st, p))) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((S + 63) / 64), dim3(256), 0, st, p.colpart, grid, S, (double*)colsum_dev);
  hipLaunchKernelGGL(colsum_center_kernel, dim3(1), dim3(256), 0, st, S, (double*)colsum_dev);
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}

// arg-max_n of vecs[n].resid / ||vecs[n]|| / S  (first maximum), result to result_dev = {double value, int64 row}.
// work_dev: 2048 doubles + 2048 int64.
// Scratch of the select step: per row and column group four doubles {shift, sum, sum of squares, dot with the residual},
// after 2048 doubles + 2048 int64 for the reduction of the arg-max.
extern "C" int64_t bcx_project_select_scratch_bytes(int32_t family, int64_t N, int32_t S) {
  if (N < 0 || S < 1) return -1;
  const int cols = 64, ngc = (S + cols - 1) / cols;     // (the narrowest column group any instantiation uses)
  return (int64_t)(4096 * sizeof(double)) + (int64_t)ngc * N * 4 * (int64_t)sizeof(double);
}

static int project_select(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                          int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                          const void* resid_dev, double resid_sum, void* result_dev, void* work_dev, void* part_dev) {
  ProjArgs p;
  int rc = fill(p, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param);
  if (rc) return rc;
  if (!resid_dev || !result_dev || !work_dev) { g_proj_err = "bcx_project_select: bad output"; return BCX_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  int grid = 0;
  proj_plan(PMODE_SELECT, family, N, S, proj_aligned(p), D, &grid, &p.team);
  p.resid = (const double*)resid_dev; p.resid_sum = resid_sum;
  p.best_val = (double*)work_dev; p.best_idx = (int64_t*)((double*)work_dev + 2048);
  // The kernel leaves every column group's share of the row moments (32 bytes per row and group); the arg-max is taken
  // by select_combine_kernel.  On XCD teams the column groups of a row block run side by side (Z streamed once instead
  // of once per column group), otherwise one workgroup walks them; the records are the same.
  const int cols = 16 * proj_nct(PMODE_SELECT, family, S, proj_aligned(p), D), ngc = (S + cols - 1) / cols;
  void* part = part_dev;
  bool own = false;
  if (!part && N > 0) {
    // (callers of the round-2 entry point: stream-ordered scratch; a caller-owned buffer does not compete with the
    // framework's allocator -- bcx_project_select_ws)
    if (hipMallocAsync(&part, (size_t)ngc * (size_t)N * 4 * sizeof(double), st) != hipSuccess) {
      (void)hipGetLastError();
      g_proj_err = "bcx_project_select: no scratch for the row moments (" + std::to_string((size_t)ngc * (size_t)N * 32) + " bytes)";
      return BCX_ERR_NOMEM;
    }
    own = true;
  }
  p.part = (double*)part;
  if (N > 0 && (rc = launch_family<PMODE_SELECT>(family, dim3(grid), 0, st, p))) { if (own) (void)hipFreeAsync(part, st); return rc; }
  const int nparts = (int)std::max<int64_t>(1, std::min<int64_t>((N + 255) / 256, 512));
  hipLaunchKernelGGL(select_combine_kernel, dim3(nparts), dim3(256), 0, st, (const double*)part, N, ngc, cols, S, p.resid, p.best_val,
                     p.best_idx);
  hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(256), 0, st, p.best_val, p.best_idx, nparts, (double*)result_dev,
                     (int64_t*)((double*)result_dev + 1));
  PROJ_HIP(hipGetLastError());
  if (own) PROJ_HIP(hipFreeAsync(part, st));
  return BCX_OK;
}

// arg-max_n of vecs[n].resid / ||vecs[n]|| / S  (first maximum), result to result_dev = {double value, int64 row}.
// work_dev: 2048 doubles + 2048 int64 (the row-moment scratch is taken from the stream-ordered allocator).
extern "C" int bcx_project_select(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                  int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                  const void* resid_dev, double resid_sum, void* result_dev, void* work_dev) {
  return project_select(stream, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param, resid_dev, resid_sum, result_dev, work_dev,
                        nullptr);
}
// The same with ALL scratch supplied by the caller: work_dev holds bcx_project_select_scratch_bytes(family, N, S) bytes.
extern "C" int bcx_project_select_ws(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                     int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                     const void* resid_dev, double resid_sum, void* result_dev, void* work_dev, int64_t work_bytes) {
  if (work_bytes < bcx_project_select_scratch_bytes(family, N, S)) { g_proj_err = "bcx_project_select_ws: scratch too small"; return BCX_ERR_ARG; }
  return project_select(stream, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param, resid_dev, resid_sum, result_dev, work_dev,
                        work_dev ? (char*)work_dev + 4096 * sizeof(double) : nullptr);
}

// rows_dev (N x ld doubles, S used per row) -= its row means, in place: projector.py:21 for rows that were written raw
// (the same kernel bcx_project_write centres with).
extern "C" int bcx_center_rows(void* stream, void* rows_dev, int64_t N, int32_t S, int64_t ld) {
  if (!rows_dev || N < 0 || S < 1 || ld < S) { g_proj_err = "bcx_center_rows: bad arguments"; return BCX_ERR_ARG; }
  if (N == 0) return BCX_OK;
  hipStream_t st = (hipStream_t)stream;
  const int g = (int)std::min<int64_t>((N + 3) / 4, 8192);
  if (S % 2 == 0 && ld % 2 == 0 && (uintptr_t)rows_dev % 16 == 0)
    hipLaunchKernelGGL(center_kernel<true>, dim3(g), dim3(256), 0, st, (double*)rows_dev, ld, N, (int)S);
  else
    hipLaunchKernelGGL(center_kernel<false>, dim3(g), dim3(256), 0, st, (double*)rows_dev, ld, N, (int)S);
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}
// out_dev[n] = sum of squares of row n of rows_dev (N x ld doubles, S used per row): hilbert.py:19-22 drops the rows where it is 0.
extern "C" int bcx_row_sumsq(void* stream, const void* rows_dev, int64_t N, int32_t S, int64_t ld, void* out_dev) {
  if (!rows_dev || !out_dev || N < 0 || S < 1 || ld < S) { g_proj_err = "bcx_row_sumsq: bad arguments"; return BCX_ERR_ARG; }
  if (N == 0) return BCX_OK;
  const int g = (int)std::min<int64_t>((N + 3) / 4, 8192);
  hipLaunchKernelGGL(row_sumsq_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const double*)rows_dev, ld, N, (int)S, (double*)out_dev);
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}
