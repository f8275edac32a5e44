// proj.hip -- device-native log-likelihood projection:
//   vecs[n][s] = loglik(z_n, theta_s) - mean_s' loglik(z_n, theta_s')        projector.py:19-21
// for the three example likelihoods of the reference
//   logistic regression   examples/common/model_lr.py:25-32
//   Poisson (softplus)    examples/common/model_poiss.py:25-38
//   Gaussian linear regr. examples/common/model_linreg.py:4-10
// as one fused kernel: Z . Theta^T on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, one wave per
// 16-row tile, all S columns in 16-column tiles) with the likelihood as the epilogue, and three
// consumers that never need the N x S matrix twice:
//   WRITE   : store the uncentred values + per-row sums (a second elementwise pass subtracts the mean)
//   COLSUM  : only the column sums  sum_n vecs[n][s]        (SparseVI gradient, sparsevi.py:70-74)
//   SELECT  : per row  corr_n = vecs[n].resid / ||vecs[n]|| / S  and its arg-max   (sparsevi.py:44-55)
// Arithmetic is fp64 throughout (the selection compares correlations to ~1e-7).
#include <algorithm>
#include <string>
#include <utility>
#include <vector>
#include "bcx_internal.h"
#include "dev_util.h"

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
  double* rowsum;       // WRITE: N
  double* colpart;      // COLSUM: gridDim.x x S partial column sums
  const double* resid;  // SELECT: S
  double resid_sum;     // SELECT: sum_s resid[s]
  double* best_val;     // SELECT: gridDim.x
  int64_t* best_idx;    // SELECT: gridDim.x
};

template <int FAM> __device__ __forceinline__ double loglik(double m, double y, double param, double c0) {
  if (FAM == FAM_LOGISTIC) {
    const double t = -m;                                   // model_lr.py:28
    return t < 100.0 ? -log1p(exp(t)) : -t;                // model_lr.py:29-31
  } else if (FAM == FAM_POISSON) {
    double s = m;                                          // model_poiss.py:25-30
    if (s > -100.0) s = log(fmax(s, 0.0) + log1p(exp(-fabs(s))));
    return y * s - c0 - exp(s);                            // model_poiss.py:38  (c0 = gammaln(y+1))
  } else {
    return c0 - (y * y - 2.0 * m * y + m * m) / (2.0 * param);   // model_linreg.py:10 (c0 = -0.5 log(2 pi sigsq))
  }
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
// it as 2 x 4 MFMA tiles (64 accumulator VGPRs).  The k (feature) dimension advances in stages of 16 values.  Per
// stage the workgroup brings 128 x 16 values of Z and 64 x 16 values of Theta from global memory -- one 16-byte
// piece per thread and load, eight consecutive threads reading one 128-byte line of a row -- into registers WHILE
// the MFMAs of the previous stage run, then parks them in the other half of a double-buffered LDS region; every
// operand is fetched from L2 once per workgroup (Theta used to be fetched once per WAVE, straight into the MFMA
// operand registers, with the load latency exposed to the matrix pipe) and one barrier separates two stages.
// The stage sequence runs across tile boundaries (next column group / next row block), so the first loads of a new
// tile are in flight during the epilogue of the previous one.
// LDS layout: one block per (16-row operand tile, 8-k step u), holding the 16-byte operand pieces
// {X[row li][k0 + 2 lk], X[row li][k0 + 2 lk + 1]} at byte lk * 256 + shift(lk) * 16 + li * 16 -- one ds_read_b128 feeds
// k-slot lk of two consecutive MFMA steps, every read of the compute loop is one per-lane base register plus an
// immediate offset.  ds_read_b128 is serviced in four fixed 16-lane groups that mix two lk values -- {0-3, 12-15 of lk 0
// with 4-11 of lk 1}, ... (MI355X_MICROARCH.md, LDS) -- so the shift is chosen per operand kind to keep every group on 16
// distinct 16-byte slots of the 256-byte bank row:
//   natural operand (lane (li, lk) reads piece [lk][li])                      shift = (0, 0, 2, 2), block 1088 B
//   replicated operand (lane reads piece [lk][4 r + (lane & 3)], 4 rows x 4)   shift = (0, 4, 4, 8), block 1184 B
// The block sizes are picked so that the eight lanes of a ds_write_b128 group (the eight (u, lk) pieces of one row: they
// loaded one 128-byte line of it) land on four different 16-byte slots of the 128-byte store row: 2-way, which a
// 13-cycle store absorbs.  (With one 288-byte stride for everything, as in the first LDS version, half of the LDS
// cycles were read conflicts: SQ_LDS_BANK_CONFLICT.)  Which matrix is natural and which replicated depends on the
// orientation (TRP): Z natural / Theta replicated for COLSUM and SELECT, the other way round for WRITE.
#define PJ_KC 16
#define PJ_ROWS 128
#define PJ_COLS 64
#define PJ_NAT_BLK 1088
#define PJ_REP_BLK 1184
#define PJ_ZBLKS (PJ_ROWS / 16 * (PJ_KC / 8))
#define PJ_TBLKS (PJ_COLS / 16 * (PJ_KC / 8))
#define PJ_STAGE_BYTES (PJ_ZBLKS * PJ_REP_BLK + PJ_TBLKS * PJ_NAT_BLK)   /* the larger of the two orientations */
#define PJ_STAGING_BYTES (2 * PJ_STAGE_BYTES)
__device__ __forceinline__ unsigned pj_nat_row(int lk) { return (unsigned)(lk * 256 + 32 * (lk >> 1)); }
__device__ __forceinline__ unsigned pj_rep_row(int lk) { return (unsigned)(lk * 256 + 64 * ((lk + 1) >> 1)); }

typedef double pv2d __attribute__((ext_vector_type(2)));

// NumPy's arg-max order on correlations: NaN (0/0 of a zero vector) beats every number, the first one wins;
// otherwise larger value, then lower row (sparsevi.py:55 `corrs.argmax()`).
__device__ __forceinline__ bool corr_better(double a, long long ia, double b, long long ib) {
  const bool an = a != a, bn = b != b;
  if (an || bn) return an && (!bn || ia < ib);
  return a > b || (a == b && ia < ib);
}

// Zero-fill of a 16-byte piece {X[k], X[k+1]} beyond D and for rows that do not exist.  The loads themselves are
// branch-free (address clamped into the row) and the mask is applied only when the piece is parked in LDS a stage
// later, so the six loads of a stage are issued back to back with no consumer in between (with the mask next to the
// load, or with branches, the compiler waited for each load before issuing the next one).  k is even.  ALIGNED (16-byte
// aligned rows, even leading dimension >= D): the piece at the last even k < D may read element D of an odd-D row --
// inside the row's padding.
__device__ __forceinline__ pv2d mask_piece(pv2d v, bool valid, int k, int D) {
  v.x = (valid && k < D) ? v.x : 0.0;
  v.y = (valid && k + 1 < D) ? v.y : 0.0;
  return v;
}

template <int FAM, int MODE, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void proj_kernel(ProjArgs p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pj_lds[];   // staging (2 stages) | COLSUM: 4 x S column sums
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int S = p.S, D = p.D;
  const int ngc = (S + PJ_COLS - 1) / PJ_COLS;
  const int nst = (D + PJ_KC - 1) / PJ_KC;
  const int64_t nblk = (p.N + PJ_ROWS - 1) / PJ_ROWS;
  double* colacc = (double*)(pj_lds + PJ_STAGING_BYTES) + (size_t)wave * S;
  if (MODE == PMODE_COLSUM) {
    for (int c = lane; c < S; c += 64) colacc[c] = 0.0;
  }
  double bestv = -INFINITY;
  long long besti = 0x7fffffffffffffffLL;
  const double clin = (FAM == FAM_LINREG) ? -0.5 * log(2.0 * 3.14159265358979323846 * p.param) : 0.0;

  // Orientation of the wave's 32 rows x 64 columns (see the compute loop): COLSUM / SELECT compute the transposed product.
  constexpr bool TRP = MODE != PMODE_WRITE;
  // global -> LDS assignment of this thread: piece q = tid & 7 (u = q >> 2, lk = q & 3) of rows (tid >> 3) + 32 j
  constexpr int ZBLK = TRP ? PJ_NAT_BLK : PJ_REP_BLK, TBLK = TRP ? PJ_REP_BLK : PJ_NAT_BLK, ZBYTES = PJ_ZBLKS * ZBLK;
  const int q = tid & 7, grow = tid >> 3;
  const unsigned zst_off = (unsigned)((q >> 2) * ZBLK + (TRP ? pj_nat_row(q & 3) : pj_rep_row(q & 3)) + (grow & 15) * 16);
  const unsigned tst_off = (unsigned)((q >> 2) * TBLK + (TRP ? pj_rep_row(q & 3) : pj_nat_row(q & 3)) + (grow & 15) * 16);

  int64_t br = blockIdx.x;
  if (br >= nblk) {
    if (MODE == PMODE_COLSUM) {
      __syncthreads();
      double* outp = p.colpart + (size_t)blockIdx.x * S;
      for (int c = tid; c < S; c += blockDim.x) outp[c] = 0.0;
    }
    if (MODE == PMODE_SELECT && tid == 0) { p.best_val[blockIdx.x] = -INFINITY; p.best_idx[blockIdx.x] = besti; }
    return;
  }
  int cg = 0, s = 0;
  pv2d zreg[4], treg[2];
  // The prefetch stream keeps, per thread, the row pointers of the tile it is loading (they change once per tile = every
  // nst stages) and one bit per row / column saying whether it exists; a stage then costs one index clamp and six
  // pointer adds, and parking is six plain stores unless the stage is the last of the k range or the tile hangs over
  // the edge of the matrix (only then pieces need zero-filling).  The first version recomputed 64-bit row addresses and
  // four selects per piece every stage: fetch + park took a quarter of the kernel's time.
  const double* zp[4];
  const double* tp[2];
  unsigned fvalid = 0;                       // bits 0-3: Z rows, bits 4-5: Theta columns of the tile being fetched
  const int kmax = ALIGNED ? ((D - 1) & ~1) : (D - 1);
  auto set_tile = [&](int64_t fbr, int fcg) {
    fvalid = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t row = fbr * PJ_ROWS + grow + 32 * j;
      zp[j] = p.Z + (row < p.N ? row : p.N - 1) * p.ldz;
      fvalid |= (row < p.N ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = fcg * PJ_COLS + grow + 32 * j;
      tp[j] = p.theta + (size_t)(col < S ? col : S - 1) * p.ldt;
      fvalid |= (col < S ? 1u : 0u) << (4 + j);
    }
  };
  auto fetch = [&](int fs) {
    const int k = fs * PJ_KC + 2 * q;
    const int kc = min(k, kmax);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (ALIGNED) zreg[j] = *(const pv2d*)(zp[j] + kc);
      else { zreg[j].x = zp[j][kc]; zreg[j].y = zp[j][min(k + 1, D - 1)]; }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (ALIGNED) treg[j] = *(const pv2d*)(tp[j] + kc);
      else { treg[j].x = tp[j][kc]; treg[j].y = tp[j][min(k + 1, D - 1)]; }
    }
  };
  auto park = [&](int par, int fs) {      // (fs: the stage of the fetch it completes)
    unsigned char* base = pj_lds + par * PJ_STAGE_BYTES;
    // wave-uniform: every row / column of the tile exists and the stage lies inside the k range for all eight pieces
    const bool plain = fs < nst - 1 && __all(fvalid == 0x3fu);
    if (!plain) {
      const int k = fs * PJ_KC + 2 * q;
#pragma unroll
      for (int j = 0; j < 4; ++j) zreg[j] = mask_piece(zreg[j], (fvalid >> j) & 1u, k, D);
#pragma unroll
      for (int j = 0; j < 2; ++j) treg[j] = mask_piece(treg[j], (fvalid >> (4 + j)) & 1u, k, D);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)      // rows grow + 32 j: 16-row tile (grow >> 4) + 2 j
      *(pv2d*)(base + ((grow >> 4) + 2 * j) * (PJ_KC / 8) * ZBLK + zst_off) = zreg[j];
#pragma unroll
    for (int j = 0; j < 2; ++j)
      *(pv2d*)(base + ZBYTES + ((grow >> 4) + 2 * j) * (PJ_KC / 8) * TBLK + tst_off) = treg[j];
  };

  set_tile(br, 0);
  fetch(0);
  park(0, 0);
  __syncthreads();
  int par = 0;
  // Orientation of the wave's 32 rows x 64 columns on the MFMA tiles.  WRITE: A = Z, B = Theta -- a lane holds 8 rows
  // x 4 columns, the 16 lanes of a DPP row hold 16 consecutive columns of one data row (128-byte stores).
  // COLSUM / SELECT: A = Theta, B = Z (the transposed product) -- a lane holds only 2 data rows (li, li + 16) x 16
  // columns, so the per-row state of the epilogue (response, shift, three moments) is 2 values per lane instead
  // of 8: what makes the kernel fit 256 registers at two waves per SIMD.
  pv4d acc[2][4];                    // [row tile][column tile]
  double yv[TRP ? 2 : 8], cp[TRP ? 2 : 8];
  double piv[2], rs[TRP ? 2 : 8], rq[2], rd[2];
  while (true) {
    // coordinates of the stage after this one
    int ns = s + 1, ncg = cg;
    int64_t nbr = br;
    if (ns == nst) { ns = 0; if (++ncg == ngc) { ncg = 0; nbr += gridDim.x; } }
    const bool more = nbr < nblk;
    const bool last = s == nst - 1;
    // the next stage's loads fly while this stage's MFMAs run; across a tile boundary they are issued after the
    // epilogue instead (keeps the 12 prefetch registers out of the epilogue's live set)
    if (more && !last) fetch(ns);
    if (s == 0) {
#pragma unroll
      for (int tr = 0; tr < 2; ++tr)
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) acc[tr][tc] = (pv4d){0.0, 0.0, 0.0, 0.0};
    }
    {
      // v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction (lanes 16k + 4b + i hold A_b[i][k],
      // lanes 16k + 4b + j hold B_b[k][j], D_b[i][j] comes back in lane 16i + 4b + j).  It sustains 72 TFLOP/s on this
      // chip where the 16x16x4 form tops out at 47.6 (tools/probe/mfma_f64_peak.hip), so the 16x16 tile is built from it:
      // the "B" operand is a natural 16-wide tile (block b = its columns 4b .. 4b+3), the "A" operand four rows replicated
      // into all four blocks -- D is then a 4 x 16 strip, and register r of the old 16x16 accumulator IS the strip of rows
      // 4r .. 4r+3, so accumulator layout and epilogue are those of the 16x16x4 version.
      const unsigned char* base = pj_lds + par * PJ_STAGE_BYTES;
      const unsigned nat_off = pj_nat_row(lk) + li * 16;                 // piece [lk][li]
      const unsigned rep_off = pj_rep_row(lk) + (lane & 3) * 16;         // piece [lk][4 r + (lane & 3)], + 64 r
      // Operand registers are double-buffered by hand: the four replicated pieces of group g + 1 are requested before the
      // 16 MFMAs of group g are issued (an LDS read takes ~130 cycles, four MFMAs 64), and within a group the .x MFMAs of
      // all accumulators precede their dependent .y MFMAs.  A group = one column tile (TRP) / one row tile (WRITE) of one
      // 8-k step; NG groups per stage.
      constexpr int NU = PJ_KC / 8, NG = TRP ? 4 * NU : 2 * NU;
      auto nat_ptr = [&](int u, int t) {      // natural operand tile t of 8-k step u
        return TRP ? (const pv2d*)(base + ((2 * wave + t) * NU + u) * ZBLK + nat_off)
                   : (const pv2d*)(base + ZBYTES + (t * NU + u) * TBLK + nat_off);
      };
      auto rep_ptr = [&](int g, int r) {      // replicated operand, group g = (u, tile), rows 4 r .. 4 r + 3
        const int u = TRP ? g / 4 : g / 2, t = TRP ? g % 4 : g % 2;
        return TRP ? (const pv2d*)(base + ZBYTES + (t * NU + u) * TBLK + rep_off + 64 * r)
                   : (const pv2d*)(base + ((2 * wave + t) * NU + u) * ZBLK + rep_off + 64 * r);
      };
      pv2d rp[2][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) rp[0][r] = *rep_ptr(0, r);
      pv2d nt[TRP ? 2 : 4];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int u = TRP ? g / 4 : g / 2, t = TRP ? g % 4 : g % 2;
        if (t == 0) {
#pragma unroll
          for (int i = 0; i < (TRP ? 2 : 4); ++i) nt[i] = *nat_ptr(u, i);
        }
        if (g + 1 < NG) {
#pragma unroll
          for (int r = 0; r < 4; ++r) rp[(g + 1) & 1][r] = *rep_ptr(g + 1, r);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < (TRP ? 2 : 4); ++i) {
            pv4d& a4 = TRP ? acc[i][t] : acc[t][i];
            a4[r] = TRP ? __builtin_amdgcn_mfma_f64_4x4x4f64(rp[g & 1][r].x, nt[i].x, a4[r], 0, 0, 0)
                        : __builtin_amdgcn_mfma_f64_4x4x4f64(rp[g & 1][r].x, nt[i].x, a4[r], 0, 0, 0);
          }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int i = 0; i < (TRP ? 2 : 4); ++i) {
            pv4d& a4 = TRP ? acc[i][t] : acc[t][i];
            a4[r] = __builtin_amdgcn_mfma_f64_4x4x4f64(rp[g & 1][r].y, nt[i].y, a4[r], 0, 0, 0);
          }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (last) {
      // ---- epilogue of tile (br, cg).  f64 C/D layout: D[i = (lane >> 4) + 4 * reg][j = lane & 15] ---------------
      const int64_t r0 = br * PJ_ROWS + 32 * wave;
      if (!TRP) {
        // i = data row (lk + 4 reg), j = column (li)
        if (cg == 0) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int64_t row = r0 + 16 * (e >> 2) + lk + 4 * (e & 3);
            const double y = (p.ycol >= 0 && row < p.N) ? p.Z[row * p.ldz + p.ycol] : 0.0;
            yv[e] = y;
            cp[e] = (FAM == FAM_POISSON) ? lgamma(y + 1.0) : clin;
            rs[e] = 0.0;
            if (FAM == FAM_POISSON) __builtin_amdgcn_sched_barrier(0);   // one lgamma at a time
          }
        }
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
          const int col = cg * PJ_COLS + 16 * tc + li;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int64_t row = r0 + 16 * (e >> 2) + lk + 4 * (e & 3);
            const bool ok = col < S && row < p.N;
            const double ll = ok ? loglik<FAM>(acc[e >> 2][tc][e & 3], yv[e], p.param, cp[e]) : 0.0;
            if (ok) p.out[row * p.ldo + col] = ll;
            rs[e] += ll;
          }
        }
        if (cg == ngc - 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const double t = row16_sum(rs[e]);
            const int64_t row = r0 + 16 * (e >> 2) + lk + 4 * (e & 3);
            if (li == 0 && row < p.N) p.rowsum[row] = t;
          }
        }
      } else {
        // i = column (lk + 4 reg within column tile tc), j = data row (li within row tile tr)
        if (cg == 0) {
#pragma unroll
          for (int tr = 0; tr < 2; ++tr) {
            const int64_t row = r0 + 16 * tr + li;
            const double y = (p.ycol >= 0 && row < p.N) ? p.Z[row * p.ldz + p.ycol] : 0.0;
            yv[tr] = y;
            cp[tr] = (FAM == FAM_POISSON) ? lgamma(y + 1.0) : clin;
            rs[tr] = 0.0; rq[tr] = 0.0; rd[tr] = 0.0;
            // per-row shift: the row's value in column 0 (lane group lk == 0, register 0 of column tile 0), handed to
            // the four lanes that share the row.  Sums, squares and dot products are accumulated on (ll - shift): the
            // one-pass moments then cancel on the scale of the row's SPREAD, not of |ll| (rows with |mean| >> spread:
            // a concentrated posterior, saturated logistic rows) -- the accuracy of the reference's centre-then-norm
            // order (sparsevi.py:49-51) without a second pass.
            const double l0 = loglik<FAM>(acc[tr][0][0], yv[tr], p.param, cp[tr]);
            piv[tr] = __shfl(l0, li, BCX_WAVE);
          }
        }
#pragma unroll
        for (int tc = 0; tc < 4; ++tc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int col = cg * PJ_COLS + 16 * tc + lk + 4 * r;
            const bool cvalid = col < S;
            const double rsd = (MODE == PMODE_SELECT && cvalid) ? p.resid[col] : 0.0;
            double csum = 0.0;
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) {
              const bool ok = cvalid && r0 + 16 * tr + li < p.N;
              const double v = ok ? loglik<FAM>(acc[tr][tc][r], yv[tr], p.param, cp[tr]) - piv[tr] : 0.0;
              if (MODE == PMODE_COLSUM) csum += v;
              else { rs[tr] += v; rq[tr] += v * v; rd[tr] += v * rsd; }
            }
            if (MODE == PMODE_COLSUM) {
              csum = row16_sum(csum);                      // the 16 lanes (data rows) that hold this column
              if (li == 0 && cvalid) colacc[col] += csum;
            }
          }
        }
        if (MODE == PMODE_SELECT && cg == ngc - 1) {
#pragma unroll
          for (int tr = 0; tr < 2; ++tr) {
            // the four lane groups hold disjoint columns of the same row
            double s1 = rs[tr], s2 = rq[tr], sd = rd[tr];
            s1 += bcx_xor16_f64(s1); s1 += bcx_xor32_f64(s1);
            s2 += bcx_xor16_f64(s2); s2 += bcx_xor32_f64(s2);
            sd += bcx_xor16_f64(sd); sd += bcx_xor32_f64(sd);
            const long long row = r0 + 16 * tr + li;
            const double mean = s1 / (double)S;                         // mean of (ll - shift)
            const double dot = sd - mean * p.resid_sum;                 // (ll - mean ll) . resid
            const double nrm2 = s2 - (double)S * mean * mean;           // ||ll - mean ll||^2
            // a row that is constant over the samples has shifted values 0 exactly: 0/0 = NaN as in NumPy
            const double corr = nrm2 > 0.0 ? dot / sqrt(nrm2) / (double)S : __builtin_nan("");
            if (row < p.N && corr_better(corr, row, bestv, besti)) { bestv = corr; besti = row; }
          }
        }
      }
    }
    if (!more) break;
    if (last) { set_tile(nbr, ncg); fetch(ns); }
    park(par ^ 1, ns);
    __syncthreads();
    par ^= 1;
    s = ns; cg = ncg; br = nbr;
  }
  if (MODE == PMODE_COLSUM) {
    __syncthreads();
    double* outp = p.colpart + (size_t)blockIdx.x * S;
    const double* ca = (const double*)(pj_lds + PJ_STAGING_BYTES);
    for (int c = tid; c < S; c += blockDim.x)
      outp[c] = ((ca[c] + ca[(size_t)S + c]) + ca[2 * (size_t)S + c]) + ca[3 * (size_t)S + c];
  }
  if (MODE == PMODE_SELECT) {
    // arg-max over the workgroup
    __shared__ double sv[4];
    __shared__ long long si[4];
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
      p.best_val[blockIdx.x] = bestv;
      p.best_idx[blockIdx.x] = besti;
    }
  }
}

// out[n][s] -= rowsum[n] / S    (lls -= lls.mean(axis=1)[:, None], projector.py:21)
__global__ __launch_bounds__(256) void center_kernel(double* out, int64_t ldo, const double* rowsum, int64_t N, int S) {
  const int64_t total = N * (int64_t)S;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t n = i / S;
    const int s = (int)(i - n * S);
    out[n * ldo + s] -= rowsum[n] / (double)S;
  }
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

#define PROJ_HIP(call)                                                            \
  do {                                                                            \
    hipError_t _e = (call);                                                       \
    if (_e != hipSuccess) {                                                       \
      g_proj_err = std::string(#call) + ": " + hipGetErrorString(_e);             \
      return BCX_ERR_HIP;                                                         \
    }                                                                             \
  } while (0)

// Persistent launch: as many workgroups as are resident at once (2 per CU: 54 KiB of staging LDS each), every one
// striding over the 128-row blocks -- with more workgroups than that the last wave of blocks leaves most CUs idle.
static int proj_grid(int64_t N) {
  static int resident = 0;
  if (!resident) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    resident = std::min(2 * cus, 2048);
  }
  const int64_t blocks = (N + PJ_ROWS - 1) / PJ_ROWS;
  return (int)std::max<int64_t>(1, std::min<int64_t>(blocks, resident));
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

template <int FAM, int MODE> static int launch_one(bool aligned, dim3 grid, size_t shmem, hipStream_t st, const ProjArgs& p) {
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
    PROJ_HIP(hipFuncSetAttribute((const void*)proj_kernel<FAM, MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL((proj_kernel<FAM, MODE, true>), grid, dim3(256), shmem, st, p);
  } else {
    PROJ_HIP(hipFuncSetAttribute((const void*)proj_kernel<FAM, MODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    hipLaunchKernelGGL((proj_kernel<FAM, MODE, false>), grid, dim3(256), shmem, st, p);
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

template <int MODE> static int launch_family(int family, dim3 grid, size_t extra_lds, hipStream_t st, const ProjArgs& p) {
  const size_t shmem = PJ_STAGING_BYTES + extra_lds;
  if (shmem > 160 * 1024) { g_proj_err = "bcx_project: S too large for the column-sum accumulators (S <= 3328)"; return BCX_ERR_ARG; }
  // 16-byte loads need 16-byte aligned rows: even leading dimensions and aligned bases (else 8-byte loads)
  const bool aligned = ((uintptr_t)p.Z % 16 == 0) && ((uintptr_t)p.theta % 16 == 0) && p.ldz % 2 == 0 && p.ldt % 2 == 0;
  switch (family) {
    case FAM_LOGISTIC: return launch_one<FAM_LOGISTIC, MODE>(aligned, grid, shmem, st, p);
    case FAM_POISSON: return launch_one<FAM_POISSON, MODE>(aligned, grid, shmem, st, p);
    case FAM_LINREG: return launch_one<FAM_LINREG, MODE>(aligned, grid, shmem, st, p);
    default: g_proj_err = "unknown likelihood family"; return BCX_ERR_ARG;
  }
}

static int fill(ProjArgs& p, int family, const void* Z, int64_t N, int64_t ldz, int D, int ycol, const void* theta,
                int S, int ldt, double param) {
  if (!Z || !theta || N < 0 || D < 1 || S < 1 || ldz < D || ldt < D || S > 4096) {
    g_proj_err = "bcx_project: bad arguments";
    return BCX_ERR_ARG;
  }
  if (family != FAM_LOGISTIC && (ycol < 0 || ycol >= ldz)) { g_proj_err = "bcx_project: response column required"; return BCX_ERR_ARG; }
  p.Z = (const double*)Z; p.theta = (const double*)theta; p.N = N; p.ldz = ldz; p.ldt = ldt; p.D = D; p.S = S;
  p.ycol = family == FAM_LOGISTIC ? -1 : ycol; p.param = param;
  p.out = nullptr; p.ldo = 0; p.rowsum = nullptr; p.colpart = nullptr; p.resid = nullptr; p.resid_sum = 0.0;
  p.best_val = nullptr; p.best_idx = nullptr;
  return BCX_OK;
}

// vecs (N x S, centred) into out_dev; rowsum_dev is N doubles of scratch.
extern "C" int bcx_project_write(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                 int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                 void* out_dev, int64_t ldo, void* rowsum_dev) {
  ProjArgs p;
  int rc = fill(p, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param);
  if (rc) return rc;
  if (!out_dev || !rowsum_dev || ldo < S) { g_proj_err = "bcx_project_write: bad output"; return BCX_ERR_ARG; }
  if (N == 0) return BCX_OK;
  p.out = (double*)out_dev; p.ldo = ldo; p.rowsum = (double*)rowsum_dev;
  hipStream_t st = (hipStream_t)stream;
  if ((rc = launch_family<PMODE_WRITE>(family, dim3(proj_grid(N)), 0, st, p))) return rc;
  const int64_t total = N * (int64_t)S;
  const int g = (int)std::min<int64_t>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(center_kernel, dim3(g), dim3(256), 0, st, p.out, ldo, p.rowsum, N, S);
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
  const int grid = proj_grid(N);
  p.colpart = (double*)work_dev;
  if ((rc = launch_family<PMODE_COLSUM>(family, dim3(grid), 4 * (size_t)S * sizeof(double), st, p))) return rc;
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((S + 63) / 64), dim3(256), 0, st, p.colpart, grid, S, (double*)colsum_dev);
  hipLaunchKernelGGL(colsum_center_kernel, dim3(1), dim3(256), 0, st, S, (double*)colsum_dev);
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}

// arg-max_n of vecs[n].resid / ||vecs[n]|| / S  (first maximum), result to result_dev = {double value, int64 row}.
// work_dev: 2048 doubles + 2048 int64.
extern "C" int bcx_project_select(void* stream, int32_t family, const void* Z_dev, int64_t N, int64_t ldz, int32_t D,
                                  int32_t ycol, const void* theta_dev, int32_t S, int32_t ldt, double param,
                                  const void* resid_dev, double resid_sum, void* result_dev, void* work_dev) {
  ProjArgs p;
  int rc = fill(p, family, Z_dev, N, ldz, D, ycol, theta_dev, S, ldt, param);
  if (rc) return rc;
  if (!resid_dev || !result_dev || !work_dev) { g_proj_err = "bcx_project_select: bad output"; return BCX_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const int grid = proj_grid(N);
  p.resid = (const double*)resid_dev; p.resid_sum = resid_sum;
  p.best_val = (double*)work_dev; p.best_idx = (int64_t*)((double*)work_dev + 2048);
  if ((rc = launch_family<PMODE_SELECT>(family, dim3(grid), 0, st, p))) return rc;
  hipLaunchKernelGGL(select_final_kernel, dim3(1), dim3(256), 0, st, p.best_val, p.best_idx, grid, (double*)result_dev,
                     (int64_t*)((double*)result_dev + 1));
  PROJ_HIP(hipGetLastError());
  return BCX_OK;
}
