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
#include "scan_core.h"

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

// Everything a scan launch is decided by (arguments, kernel variant, launch width): bcx_launch_scan below and the
// several-iterations-per-launch form of persist.hip take the same plan.
int bcx_scan_plan(bcx_solver* s, int exact, ScanArgs* ap, ScanPlan* pl) {
  ScanArgs& a = *ap;
  // storage fp64            : fp64 kernel over the normalised rows
  // storage fp32 / fp16, exact == 0: fp32-accumulating kernel (interval scan)
  // storage fp32 / fp16, exact == 1: fp64 kernel over the RAW rows (A64) when they are resident, else the
  //                           low-precision kernel again (resolve then takes its arg-max as is)
  const int d = s->cfg.d;
  const int sd = s->cfg.store_dtype;
  const bool raw64 = exact && sd != BCX_F64 && s->A64 != nullptr;
  const bool f64 = (sd == BCX_F64) || raw64;
  const bool f16 = !f64 && sd == BCX_F16;
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
  a.out = partial_view(s->partials, grid);
  pl->G = G; pl->CH = CH; pl->UR = ur; pl->grid = grid;
  pl->f64 = f64; pl->f16 = f16; pl->dual = dual; pl->long_rows = long_rows;
  return BCX_OK;
}

int bcx_launch_scan(bcx_solver* s, int exact) {
  ScanArgs a;
  ScanPlan pl;
  int rc = bcx_scan_plan(s, exact, &a, &pl);
  if (rc != BCX_OK) return rc;
  const int G = pl.G, CH = pl.CH, ur = pl.UR, grid = pl.grid;
  const bool f64 = pl.f64, f16 = pl.f16, dual = pl.dual, long_rows = pl.long_rows;
  s->n_partials = grid;                     // resolve reads exactly this launch's partials
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

