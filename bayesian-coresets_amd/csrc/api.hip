// api.hip -- host side of the C ABI declared in include/bcx.h.
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <cmath>
#include "bcx_internal.h"

static thread_local std::string g_create_err;
namespace {   // pinned upload buffers, one pair per device (defined with the upload code below)
void bounce_acquire(int device);
void bounce_release(int device);
}

extern "C" const char* bcx_version(void) { return "bcx 0.1 gfx950"; }

extern "C" const char* bcx_last_error(const bcx_solver* s) { return s ? s->err.c_str() : g_create_err.c_str(); }

template <typename T> static hipError_t dev_alloc(T** p, size_t count, bool zero = true) {
  *p = nullptr;
  if (count == 0) count = 1;
  hipError_t e = hipMalloc((void**)p, count * sizeof(T));
  if (e != hipSuccess) return e;
  if (zero) e = hipMemset(*p, 0, count * sizeof(T));
  return e;
}

static int round_up(int x, int m) { return (x + m - 1) / m * m; }

static void free_all(bcx_solver* s) {
  void* ptrs[] = {s->An, s->A64, s->norms, s->chunk_sums, s->staging, s->st, s->b, s->bn, s->xw, s->q64, s->qst,
                  s->tmp, s->partials, s->rec_local, s->act_idx, s->act_w, s->act_rows, s->act_norm, s->gram,
                  s->hinv, s->hinv_lo, s->cvec, s->plist, s->ppos, s->nn_x, s->nn_z, s->nn_wv, s->nn_tmp, s->nn_flag, s->nn_wbak, s->nn_xr, s->tr_sel, s->tr_err,
                  s->tr_status};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (size_t w = 0; w < s->peer_mbox.size(); ++w)
    if (s->peer_mbox[w] && s->peer_mbox[w] != s->mbox) (void)hipIpcCloseMemHandle(s->peer_mbox[w]);
  void* xptrs[] = {s->mbox, s->peer_tab, s->xseq, s->xprobe, s->rec_gather, s->grid_counter, s->fin_part, s->gram_work, s->warm_buf, s->pflags, s->pdbg};
  for (void* p : xptrs)
    if (p) (void)hipFree(p);
  for (auto& ev : s->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
}

extern "C" int bcx_create(const bcx_config* cfg, bcx_solver** out) {
  if (!cfg || !out) { g_create_err = "bcx_create: null argument"; return BCX_ERR_ARG; }
  *out = nullptr;
  if (cfg->alg < BCX_ALG_GIGA || cfg->alg > BCX_ALG_OMP || cfg->d < 1 || cfg->n_local < 0 ||
      cfg->world_size < 1 || cfg->rank < 0 || cfg->rank >= cfg->world_size ||
      (cfg->store_dtype != BCX_F32 && cfg->store_dtype != BCX_F64 && cfg->store_dtype != BCX_F16)) {
    g_create_err = "bcx_create: invalid configuration";
    return BCX_ERR_ARG;
  }
  const int maxd = BCX_MAX_D;
  if (cfg->d > maxd) { g_create_err = "bcx_create: d exceeds the supported row length (" + std::to_string(maxd) + ")"; return BCX_ERR_ARG; }
  if (cfg->n_local >= (int64_t)0x7fffffff) { g_create_err = "bcx_create: n_local must be < 2^31 per shard"; return BCX_ERR_ARG; }
  if (cfg->row_offset % BCX_CHUNK_ROWS != 0) {
    g_create_err = "bcx_create: row_offset must be a multiple of the chunk size (1024 rows)";
    return BCX_ERR_ARG;
  }
  bcx_solver* s = new bcx_solver();
  s->cfg = *cfg;
  if (s->cfg.refresh_every == 0) s->cfg.refresh_every = 64;
  hipError_t e = hipSetDevice(cfg->device);
  if (e != hipSuccess) { g_create_err = std::string("hipSetDevice: ") + hipGetErrorString(e); delete s; return BCX_ERR_HIP; }
  const int d = cfg->d;
  const int64_t n = cfg->n_local;
  s->elem = cfg->store_dtype == BCX_F32 ? 4 : (cfg->store_dtype == BCX_F16 ? 2 : 8);
  s->qelem = cfg->store_dtype == BCX_F64 ? 8 : 4;   // the query the scan reads: fp32 for fp32/fp16 rows
  s->ld = round_up(d, 16 / s->elem);
  s->ld64 = round_up(d, 2);
  s->n_chunks = (n + BCX_CHUNK_ROWS - 1) / BCX_CHUNK_ROWS;
  bool ok = true;
  auto chk = [&](hipError_t r) { if (r != hipSuccess && ok) { ok = false; g_create_err = std::string("hipMalloc: ") + hipGetErrorString(r); } };
  chk(dev_alloc((char**)&s->An, (size_t)n * s->ld * s->elem));
  if (cfg->keep_exact_rows && cfg->store_dtype != BCX_F64) chk(dev_alloc(&s->A64, (size_t)n * s->ld64));
  chk(dev_alloc(&s->norms, (size_t)n));
  chk(dev_alloc(&s->chunk_sums, (size_t)s->n_chunks * (d + 1)));
  chk(dev_alloc(&s->st, 1));
  chk(dev_alloc(&s->b, (size_t)d));
  chk(dev_alloc(&s->bn, (size_t)d));
  chk(dev_alloc(&s->xw, (size_t)d));
  chk(dev_alloc(&s->q64, (size_t)2 * s->ld64));
  chk(dev_alloc((char**)&s->qst, (size_t)2 * s->ld * s->qelem));
  chk(dev_alloc(&s->tmp, (size_t)20 * d));
  s->n_partials = 0;
  chk(dev_alloc((char**)&s->partials, (size_t)bcx_scan_grid(s) * BCX_PARTIAL_BYTES));
  chk(dev_alloc(&s->rec_local, (size_t)(d + BCX_REC_HDR)));
  if (!ok) { free_all(s); delete s; return BCX_ERR_NOMEM; }
  bounce_acquire(cfg->device);
  *out = s;
  return BCX_OK;
}

extern "C" int bcx_destroy(bcx_solver* s) {
  if (!s) return BCX_OK;
  (void)hipSetDevice(s->cfg.device);
  (void)hipDeviceSynchronize();
  free_all(s);
  bounce_release(s->cfg.device);
  delete s;
  return BCX_OK;
}

extern "C" int bcx_set_stream(bcx_solver* s, void* hip_stream) {
  if (!s) return BCX_ERR_ARG;
  s->stream = (hipStream_t)hip_stream;
  return BCX_OK;
}

static int read_state(bcx_solver* s, DevState* h);

// ---- host -> device upload --------------------------------------------------------------------------
// The GPU never touches the caller's memory: rows are copied by host threads into two pinned bounce buffers that
// the library owns (hipHostMalloc, allocated once per process) and DMA'd from there, the memcpy of one buffer
// overlapping the DMA of the other.  (Round 1 pinned the caller's ndarray with hipHostRegister for the duration
// of the copy; on a fresh box that path died intermittently with "Memory access fault by GPU ... Reason: Unknown"
// at an address inside the registered heap range -- a user-pointer mapping of glibc heap pages is only as stable
// as the heap is -- and took the whole GPU test run with it.)  On return the caller's buffer is no longer in use.
#include <mutex>
#include <thread>
namespace {
constexpr size_t kBounceBytes = (size_t)32 << 20;
constexpr int kMaxDevices = 64;
// One pair of pinned buffers and one lock PER DEVICE: uploads to different GPUs of one process run side by side (round 2
// had a single process-wide pair: solvers on different devices queued behind each other); the pair is freed when the
// last solver of its device is destroyed.
struct Bounce {
  std::mutex mu;
  void* buf[2] = {nullptr, nullptr};
  int users = 0;
};
Bounce g_bounce[kMaxDevices];
Bounce& bounce_of(int device) { return g_bounce[(device >= 0 && device < kMaxDevices) ? device : 0]; }
void bounce_acquire(int device) {
  Bounce& b = bounce_of(device);
  std::lock_guard<std::mutex> lock(b.mu);
  b.users += 1;
}
void bounce_release(int device) {
  Bounce& b = bounce_of(device);
  std::lock_guard<std::mutex> lock(b.mu);
  if (--b.users <= 0) {
    b.users = 0;
    for (int i = 0; i < 2; ++i)
      if (b.buf[i]) { (void)hipHostFree(b.buf[i]); b.buf[i] = nullptr; }
  }
}

void copy_rows_mt(char* dst, size_t dpitch, const char* src, size_t spitch, size_t width, int64_t rows) {
  const size_t bytes = (size_t)rows * width;
  int nt = (int)std::min<size_t>(8, bytes / ((size_t)2 << 20));      // one thread per 2 MiB, at most 8
  const unsigned hw = std::thread::hardware_concurrency();
  if (hw && (unsigned)nt > hw) nt = (int)hw;
  auto work = [=](int64_t r0, int64_t r1) {
    if (dpitch == width && spitch == width) { memcpy(dst + (size_t)r0 * width, src + (size_t)r0 * width, (size_t)(r1 - r0) * width); return; }
    for (int64_t r = r0; r < r1; ++r) memcpy(dst + (size_t)r * dpitch, src + (size_t)r * spitch, width);
  };
  if (nt <= 1) { work(0, rows); return; }
  std::vector<std::thread> th;
  const int64_t per = (rows + nt - 1) / nt;
  for (int t = 1; t < nt; ++t) {
    const int64_t r0 = std::min<int64_t>(rows, t * per), r1 = std::min<int64_t>(rows, (t + 1) * per);
    if (r0 < r1) th.emplace_back(work, r0, r1);
  }
  work(0, std::min<int64_t>(rows, per));
  for (auto& t : th) t.join();
}
}  // namespace

static int upload_host_rows(bcx_solver* s, void* dst_dev, size_t dpitch, const void* src, size_t spitch, size_t width,
                            int64_t rows) {
  if (rows <= 0) return BCX_OK;
  Bounce& g_b = bounce_of(s->cfg.device);
  std::lock_guard<std::mutex> lock(g_b.mu);
  for (int i = 0; i < 2; ++i)
    if (!g_b.buf[i]) BCX_HIP(hipHostMalloc(&g_b.buf[i], kBounceBytes, hipHostMallocPortable));
  if (width > kBounceBytes) { s->err = "bcx_load_rows: a row exceeds the upload buffer"; return BCX_ERR_ARG; }
  hipEvent_t done[2] = {nullptr, nullptr};     // (per call: events belong to the device that is current now)
  bool busy[2] = {false, false};
  for (int i = 0; i < 2; ++i) {
    const hipError_t ee = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    if (ee != hipSuccess) {
      if (i == 1) (void)hipEventDestroy(done[0]);
      s->err = std::string("upload: ") + hipGetErrorString(ee);
      return BCX_ERR_HIP;
    }
  }
  const int64_t rows_per = std::max<int64_t>(1, (int64_t)(kBounceBytes / width));
  int which = 0;
  int rc = BCX_OK;
  for (int64_t r = 0; r < rows && rc == BCX_OK; r += rows_per, which ^= 1) {
    const int64_t m = std::min(rows_per, rows - r);
    if (busy[which]) {
      if (hipEventSynchronize(done[which]) != hipSuccess) { s->err = "upload: event wait failed"; rc = BCX_ERR_HIP; break; }
      busy[which] = false;
    }
    copy_rows_mt((char*)g_b.buf[which], width, (const char*)src + (size_t)r * spitch, spitch, width, m);
    hipError_t e = hipMemcpy2DAsync((char*)dst_dev + (size_t)r * dpitch, dpitch, g_b.buf[which], width, width, (size_t)m,
                                    hipMemcpyHostToDevice, s->stream);
    if (e == hipSuccess) e = hipEventRecord(done[which], s->stream);
    if (e != hipSuccess) { s->err = std::string("upload: ") + hipGetErrorString(e); rc = BCX_ERR_HIP; break; }
    busy[which] = true;
  }
  // the bounce buffers are shared by every solver of the device: leave them idle
  for (int i = 0; i < 2; ++i) {
    if (busy[i]) (void)hipEventSynchronize(done[i]);
    (void)hipEventDestroy(done[i]);
  }
  return rc;
}


static int load_rows(bcx_solver* s, const void* src, int32_t src_is_device, int32_t src_dtype, int64_t row_begin,
                     int64_t rows, int64_t ld, int32_t flags);

extern "C" int bcx_load_rows(bcx_solver* s, const void* src, int32_t src_is_device, int32_t src_dtype,
                             int64_t row_begin, int64_t rows, int64_t ld) {
  return load_rows(s, src, src_is_device, src_dtype, row_begin, rows, ld, 0);
}

extern "C" int bcx_load_rows_flags(bcx_solver* s, const void* src, int32_t src_is_device, int32_t src_dtype,
                                   int64_t row_begin, int64_t rows, int64_t ld, int32_t flags) {
  if (flags & ~BCX_LOAD_CENTER_ROWS) { if (s) s->err = "bcx_load_rows_flags: unknown flag"; return BCX_ERR_ARG; }
  return load_rows(s, src, src_is_device, src_dtype, row_begin, rows, ld, flags);
}

static int load_rows(bcx_solver* s, const void* src, int32_t src_is_device, int32_t src_dtype, int64_t row_begin,
                     int64_t rows, int64_t ld, int32_t flags) {
  if (!s || (!src && rows > 0)) return BCX_ERR_ARG;
  const int center = (flags & BCX_LOAD_CENTER_ROWS) ? 1 : 0;
  const int d = s->cfg.d;
  if (rows == 0) return BCX_OK;
  if (row_begin < 0 || rows < 0 || row_begin + rows > s->cfg.n_local || ld < d ||
      (src_dtype != BCX_F32 && src_dtype != BCX_F64)) {
    s->err = "bcx_load_rows: bad range, leading dimension or dtype";
    return BCX_ERR_ARG;
  }
  if (row_begin % BCX_CHUNK_ROWS != 0 || (rows % BCX_CHUNK_ROWS != 0 && row_begin + rows != s->cfg.n_local)) {
    s->err = "bcx_load_rows: pieces must start and end on 1024-row chunk boundaries (except the last)";
    return BCX_ERR_ARG;
  }
  BCX_HIP(hipSetDevice(s->cfg.device));
  if (row_begin == 0) {
    // a (re)load from the first row starts a new matrix: forget the previous one's zero-row flag and state.
    // Unconditionally -- a matrix whose bcx_finalize reported BCX_ERR_ZERO_ROW never became `finalized`, and its
    // flag must not outlive it (SparseVI reloads one engine with a fresh projection every step).
    if (s->finalized || s->rows_loaded > 0) {
      DevState h;
      int rc0 = read_state(s, &h);
      if (rc0 != BCX_OK) return rc0;
      h.zero_row = 0; h.k = 0; h.np = 0; h.hvalid = 1; h.omp_ill = 0; h.limit = 0; h.active = 0; h.halt = HALT_NONE;
      BCX_HIP(hipMemcpy(s->st, &h, sizeof h, hipMemcpyHostToDevice));
      s->finalized = false;
    }
    s->rows_loaded = 0;
  }
  const size_t esz = src_dtype == BCX_F64 ? 8 : 4;
  if (src_is_device) {
    int rc = bcx_launch_ingest(s, src, src_dtype, ld, row_begin, rows, center);
    if (rc != BCX_OK) return rc;
  } else if (s->A64 && src_dtype == BCX_F64) {
    // host fp64 rows go straight to their final place; the ingest kernel then works in place
    double* dst = s->A64 + (size_t)row_begin * s->ld64;
    int rc = upload_host_rows(s, dst, (size_t)s->ld64 * 8, src, (size_t)ld * 8, (size_t)d * 8, rows);
    if (rc != BCX_OK) return rc;
    rc = bcx_launch_ingest(s, dst, BCX_F64, s->ld64, row_begin, rows, center);
    if (rc != BCX_OK) return rc;
  } else {
    // stage through a device buffer in pieces of <= 256 MiB
    const int64_t piece = std::max<int64_t>(BCX_CHUNK_ROWS,
                                            ((int64_t)(256u << 20) / (int64_t)(d * esz)) / BCX_CHUNK_ROWS * BCX_CHUNK_ROWS);
    const size_t need = (size_t)std::min(piece, rows) * d * esz;
    if (s->staging_bytes < need) {
      if (s->staging) BCX_HIP(hipFree(s->staging));
      s->staging = nullptr; s->staging_bytes = 0;
      BCX_HIP(hipMalloc(&s->staging, need));
      s->staging_bytes = need;
    }
    for (int64_t r = 0; r < rows; r += piece) {
      const int64_t m = std::min(piece, rows - r);
      int rc = upload_host_rows(s, s->staging, (size_t)d * esz, (const char*)src + (size_t)r * ld * esz, (size_t)ld * esz,
                                (size_t)d * esz, m);
      if (rc != BCX_OK) return rc;
      rc = bcx_launch_ingest(s, s->staging, src_dtype, d, row_begin + r, m, center);
      if (rc != BCX_OK) return rc;
      BCX_HIP(hipStreamSynchronize(s->stream));  // staging buffer is reused
    }
  }
  s->rows_loaded += rows;
  return BCX_OK;
}

extern "C" int bcx_chunk_sums(bcx_solver* s, const void** dev_ptr, int64_t* n_chunks, int64_t* chunk_rows) {
  if (!s) return BCX_ERR_ARG;
  if (dev_ptr) *dev_ptr = s->chunk_sums;
  if (n_chunks) *n_chunks = s->n_chunks;
  if (chunk_rows) *chunk_rows = BCX_CHUNK_ROWS;
  return BCX_OK;
}

extern "C" int bcx_export_chunk_sums(bcx_solver* s, void* dst_dev, int64_t cap_chunks) {
  if (!s || !dst_dev || cap_chunks < s->n_chunks) return BCX_ERR_ARG;
  BCX_HIP(hipMemcpyAsync(dst_dev, s->chunk_sums, (size_t)s->n_chunks * (s->cfg.d + 1) * 8, hipMemcpyDeviceToDevice,
                         s->stream));
  return BCX_OK;
}

static int read_state(bcx_solver* s, DevState* h) {
  BCX_HIP(hipStreamSynchronize(s->stream));
  BCX_HIP(hipMemcpy(h, s->st, sizeof(DevState), hipMemcpyDeviceToHost));
  return BCX_OK;
}

extern "C" int bcx_finalize(bcx_solver* s, const double* b_host, const void* gathered_sums_dev, int64_t n_gathered) {
  if (!s) return BCX_ERR_ARG;
  BCX_HIP(hipSetDevice(s->cfg.device));
  if (s->rows_loaded < s->cfg.n_local) { s->err = "bcx_finalize: not all rows were loaded"; return BCX_ERR_STATE; }
  if (b_host) BCX_HIP(hipMemcpyAsync(s->b, b_host, (size_t)s->cfg.d * 8, hipMemcpyHostToDevice, s->stream));
  int rc = bcx_launch_finalize(s, b_host != nullptr, (const double*)gathered_sums_dev, n_gathered);
  if (rc != BCX_OK) return rc;
  DevState h;
  rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  if (h.zero_row != 0) {
    char buf[128];
    snprintf(buf, sizeof buf, "A must not have any 0 columns (local row %d has zero norm)", h.zero_row - 1);
    s->err = buf;
    return BCX_ERR_ZERO_ROW;
  }
  if (s->cfg.alg == BCX_ALG_GIGA && h.bnorm == 0.0) { s->err = "norm of b must be > 0"; return BCX_ERR_ZERO_B; }
  s->finalized = true;
  return BCX_OK;
}

// ---- capacity management -----------------------------------------------------------------
template <typename T> static int grow(bcx_solver* s, T** p, size_t old_count, size_t new_count) {
  T* q = nullptr;
  BCX_HIP(dev_alloc(&q, new_count));
  if (*p && old_count) BCX_HIP(hipMemcpy(q, *p, old_count * sizeof(T), hipMemcpyDeviceToDevice));
  if (*p) BCX_HIP(hipFree(*p));
  *p = q;
  return BCX_OK;
}

static int ensure_slots(bcx_solver* s, int64_t need) {
  if (need <= s->cap) return BCX_OK;
  int64_t ncap = std::max<int64_t>(need, std::max<int64_t>(256, s->cap * 2));
  const size_t d = (size_t)s->cfg.d;
  int rc;
  if ((rc = grow(s, &s->act_idx, (size_t)s->cap, (size_t)ncap))) return rc;
  if ((rc = grow(s, &s->act_w, (size_t)s->cap, (size_t)ncap))) return rc;
  if ((rc = grow(s, &s->act_norm, (size_t)s->cap, (size_t)ncap))) return rc;
  if ((rc = grow(s, &s->act_rows, (size_t)s->cap * d, (size_t)ncap * d))) return rc;
  s->cap = ncap;
  return BCX_OK;
}

int bcx_ensure_gram(bcx_solver* s, int64_t need) {
  if (need <= s->gram_cap) return BCX_OK;
  const int64_t ncap = std::max<int64_t>((need + 63) / 64 * 64, std::max<int64_t>(256, s->gram_cap * 2));
  const size_t oc = (size_t)s->gram_cap, nc = (size_t)ncap;
  // gram / hinv change their leading dimension: copy row by row
  double* g2 = nullptr; double* h2 = nullptr; double* l2 = nullptr;
  BCX_HIP(dev_alloc(&g2, nc * nc));
  BCX_HIP(dev_alloc(&h2, nc * nc));
  BCX_HIP(dev_alloc(&l2, nc * nc));
  if (oc) {
    BCX_HIP(hipMemcpy2D(g2, nc * 8, s->gram, oc * 8, oc * 8, oc, hipMemcpyDeviceToDevice));
    BCX_HIP(hipMemcpy2D(h2, nc * 8, s->hinv, oc * 8, oc * 8, oc, hipMemcpyDeviceToDevice));
    BCX_HIP(hipMemcpy2D(l2, nc * 8, s->hinv_lo, oc * 8, oc * 8, oc, hipMemcpyDeviceToDevice));
    BCX_HIP(hipFree(s->gram));
    BCX_HIP(hipFree(s->hinv));
    BCX_HIP(hipFree(s->hinv_lo));
  }
  s->gram = g2; s->hinv = h2; s->hinv_lo = l2;
  int rc;
  if ((rc = grow(s, &s->cvec, oc, nc))) return rc;
  if ((rc = grow(s, &s->plist, oc, nc))) return rc;
  if ((rc = grow(s, &s->ppos, oc, nc))) return rc;
  if ((rc = grow(s, &s->nn_x, oc, nc))) return rc;
  if ((rc = grow(s, &s->nn_z, 0, nc))) return rc;
  if ((rc = grow(s, &s->nn_wv, 0, nc))) return rc;
  if ((rc = grow(s, &s->nn_tmp, 0, 16 * nc))) return rc;   // t0..t3 + the exchange ring of grid_lh.h (GRID_RING = 16 buffers)
  if ((rc = grow(s, &s->nn_flag, 0, nc))) return rc;
  if ((rc = grow(s, &s->nn_wbak, 0, nc))) return rc;
  if ((rc = grow(s, &s->nn_xr, 0, 2 * nc))) return rc;
  s->gram_cap = ncap;
  return BCX_OK;
}

static int ensure_trace(bcx_solver* s, int64_t need) {
  if (need <= s->trace_cap) return BCX_OK;
  int64_t ncap = std::max<int64_t>(need, 1024);
  int rc;
  if ((rc = grow(s, &s->tr_sel, 0, (size_t)ncap))) return rc;
  if ((rc = grow(s, &s->tr_err, 0, (size_t)ncap))) return rc;
  if ((rc = grow(s, &s->tr_status, 0, (size_t)ncap))) return rc;
  s->trace_cap = ncap;
  return BCX_OK;
}

// ---- build ---------------------------------------------------------------------------------
extern "C" int bcx_build_begin(bcx_solver* s, int64_t itrs, double tol, int32_t* skip) {
  if (!s || !skip) return BCX_ERR_ARG;
  if (!s->finalized) { s->err = "bcx_build_begin: call bcx_finalize first"; return BCX_ERR_STATE; }
  BCX_HIP(hipSetDevice(s->cfg.device));
  *skip = 0;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  if (h.limit || s->cfg.n_global == 0 || itrs <= 0) { *skip = 1; return BCX_OK; }   // snnls.py:32-38
  s->k_ub = h.k;
  if ((rc = ensure_slots(s, (int64_t)h.k + itrs))) return rc;
  if (s->cfg.alg == BCX_ALG_OMP) {
    if ((rc = bcx_ensure_gram(s, (int64_t)h.k + itrs))) return rc;
    if (!s->grid_counter && dev_alloc(&s->grid_counter, 32 /* = BCX_GRID_WORDS, nnls_common.h */) != hipSuccess) { s->err = "grid counter allocation failed"; return BCX_ERR_NOMEM; }
    BCX_HIP(hipMemsetAsync(s->grid_counter, 0, 2 * sizeof(unsigned long long), s->stream));
    s->grid_epoch = 0;
    s->grid_dirty = false;
  }
  if ((rc = ensure_trace(s, itrs))) return rc;
  return bcx_launch_begin(s, itrs, tol);
}

// Event pairs around the scan launch, on every prof_every-th launch: a pair costs 7-12 us of stream time
// (measured: GIGA N=1M d=256 171 -> 159 us per iteration without them), so the bench samples.
// (weight: greedy iterations the bracketed launch covers -- 1, or a whole batch of them for persist.hip, which times EVERY
// launch: a pair per ~100 iterations costs nothing)
int bcx_prof_begin(bcx_solver* s, int weight, bool every_launch) {
  if (!s->profile) return BCX_OK;
  s->prof_now = every_launch || (s->prof_tick++ % s->prof_every) == 0;
  if (!s->prof_now) return BCX_OK;
  if (s->prof_used == s->prof_events.size()) {
    hipEvent_t a, b;
    BCX_HIP(hipEventCreate(&a));
    BCX_HIP(hipEventCreate(&b));
    s->prof_events.emplace_back(a, b);
    s->prof_weight.push_back(1);
  }
  s->prof_weight[s->prof_used] = weight;
  BCX_HIP(hipEventRecord(s->prof_events[s->prof_used].first, s->stream));
  return BCX_OK;
}
int bcx_prof_end(bcx_solver* s) {
  if (!s->profile || !s->prof_now) return BCX_OK;
  BCX_HIP(hipEventRecord(s->prof_events[s->prof_used].second, s->stream));
  s->prof_used++;
  return BCX_OK;
}
static int prof_begin(bcx_solver* s) { return bcx_prof_begin(s, 1, false); }
static int prof_end(bcx_solver* s) { return bcx_prof_end(s); }

static int step_scan(bcx_solver* s, void* send_dev, int exact, bool with_tail) {
  if (!s->finalized) { s->err = "solver not finalized"; return BCX_ERR_STATE; }
  double* send = send_dev ? (double*)send_dev : s->rec_local;
  int rc;
  if (exact && (rc = bcx_launch_resume_exact(s))) return rc;
  if ((rc = prof_begin(s))) return rc;
  if ((rc = bcx_launch_scan(s, exact))) return rc;
  if ((rc = prof_end(s))) return rc;
  if (with_tail) return bcx_launch_tail(s, exact);   // resolve + apply in one launch (single shard, GIGA/FW)
  return bcx_launch_resolve(s, send, exact);
}

extern "C" int bcx_step_scan(bcx_solver* s, void* send_dev) { return s ? step_scan(s, send_dev, 0, false) : BCX_ERR_ARG; }
extern "C" int bcx_step_scan_exact(bcx_solver* s, void* send_dev) { return s ? step_scan(s, send_dev, 1, false) : BCX_ERR_ARG; }

extern "C" int bcx_step_apply(bcx_solver* s, const void* recv_dev) {
  if (!s) return BCX_ERR_ARG;
  return bcx_launch_apply(s, recv_dev ? (const double*)recv_dev : s->rec_local);
}

// one whole iteration without the host: single shard, or row shards with an attached peer mailbox
static int enqueue_one(bcx_solver* s, int exact) {
  if (s->cfg.world_size != 1) {
    if (!s->exchange_ready) {
      s->err = "bcx_build_enqueue on a row shard needs bcx_exchange_attach (or drive bcx_step_scan / bcx_step_apply)";
      return BCX_ERR_ARG;
    }
    int rc;
    if (exact && (rc = bcx_launch_resume_exact(s))) return rc;
    if ((rc = prof_begin(s))) return rc;
    if ((rc = bcx_launch_scan(s, exact))) return rc;
    if ((rc = prof_end(s))) return rc;
    return bcx_launch_tail_exchange(s, exact);
  }
  if (s->cfg.alg == BCX_ALG_OMP) {
    // scan, then the fused resolve + OMP step (omp_lh.hip); where that does not apply: resolve_kernel + the multi-kernel step
    int rc;
    if (exact && (rc = bcx_launch_resume_exact(s))) return rc;
    if ((rc = prof_begin(s))) return rc;
    if ((rc = bcx_launch_scan(s, exact))) return rc;
    if ((rc = prof_end(s))) return rc;
    rc = bcx_launch_omp_fused(s, exact);
    if (rc != 1) return rc;
    if ((rc = bcx_launch_resolve(s, s->rec_local, exact))) return rc;
    return bcx_launch_apply(s, s->rec_local);
  }
  return step_scan(s, nullptr, exact, true);
}

extern "C" int bcx_build_enqueue(bcx_solver* s, int64_t itrs) {
  if (!s) return BCX_ERR_ARG;
  if (!s->finalized) { s->err = "solver not finalized"; return BCX_ERR_STATE; }
  int64_t i = 0;
  // single shard, GIGA / FW, a few GB of rows: batches of iterations per launch with the tail's workgroup resident beside
  // the scan's (persist.hip); everything else, and what that form declines: one launch per kernel
  bool batches = s->cfg.world_size == 1 && s->cfg.alg != BCX_ALG_OMP;
  while (i < itrs) {
    if (batches) {
      int64_t covered = 0;
      const int rc = bcx_launch_persist(s, itrs - i, &covered);
      if (rc == BCX_OK) { i += covered; continue; }
      if (rc != 1) return rc;
      batches = false;
    }
    const int rc = enqueue_one(s, 0);
    if (rc != BCX_OK) return rc;
    ++i;
  }
  return BCX_OK;
}

extern "C" int bcx_build_enqueue_exact(bcx_solver* s) {
  if (!s) return BCX_ERR_ARG;
  if (!s->finalized) { s->err = "solver not finalized"; return BCX_ERR_STATE; }
  return enqueue_one(s, 1);
}

// ---- peer mailbox --------------------------------------------------------------------------------
static int mailbox_alloc(bcx_solver* s) {
  if (s->mbox) return BCX_OK;
  const int world = s->cfg.world_size;
  const size_t bytes = bcx_mailbox_slot_offset(world) + 2 * (size_t)world * (s->cfg.d + BCX_REC_HDR) * sizeof(double);
  // fine-grained: stores from peer GPUs become visible to a kernel that is already running here
  hipError_t e = hipExtMallocWithFlags(&s->mbox, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    s->mbox = nullptr;
    s->err = std::string("peer mailbox: fine-grained allocation failed: ") + hipGetErrorString(e);
    return BCX_ERR_HIP;
  }
  s->mbox_bytes = bytes;
  BCX_HIP(hipMemset(s->mbox, 0, bytes));
  BCX_HIP(hipDeviceSynchronize());
  return BCX_OK;
}

extern "C" int bcx_exchange_export(bcx_solver* s, void* handle_out, int32_t handle_bytes) {
  if (!s || !handle_out) return BCX_ERR_ARG;
  if (handle_bytes < (int32_t)sizeof(hipIpcMemHandle_t)) { s->err = "bcx_exchange_export: handle buffer too small"; return BCX_ERR_ARG; }
  if ((size_t)s->cfg.d * sizeof(double) > 144 * 1024) {
    // the exchange kernels stage one record (d doubles) in LDS; longer rows use the host-driven all-gather exchange
    s->err = "peer mailbox: rows of more than 18432 values exchange their records through the all-gather";
    return BCX_ERR_ARG;
  }
  BCX_HIP(hipSetDevice(s->cfg.device));
  int rc = mailbox_alloc(s);
  if (rc != BCX_OK) return rc;
  hipIpcMemHandle_t h;
  BCX_HIP(hipIpcGetMemHandle(&h, s->mbox));
  memset(handle_out, 0, handle_bytes);
  memcpy(handle_out, &h, sizeof(h));
  return BCX_OK;
}

extern "C" int bcx_exchange_attach(bcx_solver* s, const void* handles, int32_t handle_bytes, double timeout_s) {
  if (!s || !handles) return BCX_ERR_ARG;
  if (s->exchange_ready) return BCX_OK;
  if (!s->mbox) { s->err = "bcx_exchange_attach: call bcx_exchange_export first"; return BCX_ERR_STATE; }
  if (handle_bytes < (int32_t)sizeof(hipIpcMemHandle_t)) return BCX_ERR_ARG;
  BCX_HIP(hipSetDevice(s->cfg.device));
  const int world = s->cfg.world_size;
  if (world > BCX_APPLY_THREADS) { s->err = "peer mailbox supports at most 256 shards"; return BCX_ERR_ARG; }
  s->peer_mbox.assign(world, nullptr);
  for (int w = 0; w < world; ++w) {
    if (w == s->cfg.rank) { s->peer_mbox[w] = s->mbox; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)w * handle_bytes, sizeof(h));
    hipError_t e = hipIpcOpenMemHandle(&s->peer_mbox[w], h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      s->peer_mbox[w] = nullptr;
      s->err = "peer mailbox: cannot map shard " + std::to_string(w) + ": " + hipGetErrorString(e);
      for (int v = 0; v < w; ++v)
        if (v != s->cfg.rank && s->peer_mbox[v]) { (void)hipIpcCloseMemHandle(s->peer_mbox[v]); s->peer_mbox[v] = nullptr; }
      s->peer_mbox.clear();
      return BCX_ERR_HIP;
    }
  }
  hipError_t e;
  if ((e = dev_alloc(&s->peer_tab, (size_t)world)) != hipSuccess || (e = dev_alloc(&s->xseq, 8)) != hipSuccess ||
      (e = dev_alloc(&s->xprobe, 1)) != hipSuccess ||
      (e = dev_alloc(&s->rec_gather, (size_t)world * (s->cfg.d + BCX_REC_HDR))) != hipSuccess) {
    s->err = std::string("peer mailbox: ") + hipGetErrorString(e);
    return BCX_ERR_NOMEM;
  }
  BCX_HIP(hipMemcpy(s->peer_tab, s->peer_mbox.data(), (size_t)world * sizeof(void*), hipMemcpyHostToDevice));
  if (timeout_s > 0.0) s->exchange_timeout_s = timeout_s;
  s->exchange_ready = true;
  return BCX_OK;
}

extern "C" int bcx_exchange_probe(bcx_solver* s, int32_t* result) {
  if (!s || !result) return BCX_ERR_ARG;
  if (!s->exchange_ready) { s->err = "bcx_exchange_probe: no peer mailbox attached"; return BCX_ERR_STATE; }
  BCX_HIP(hipSetDevice(s->cfg.device));
  int rc = bcx_launch_exchange_probe(s);
  if (rc != BCX_OK) return rc;
  BCX_HIP(hipStreamSynchronize(s->stream));
  BCX_HIP(hipMemcpy(result, s->xprobe, sizeof(int32_t), hipMemcpyDeviceToHost));
  return BCX_OK;
}

extern "C" int bcx_exchange_set_timeout(bcx_solver* s, double timeout_s) {
  if (!s || !(timeout_s > 0.0)) return BCX_ERR_ARG;
  s->exchange_timeout_s = timeout_s;
  return BCX_OK;
}

// Device-side timing of the record exchange since the last reset: n exchanges, mean / max (microseconds) of the wait for
// the slowest peer's record after this shard posted its own, and of the whole exchange step.
extern "C" int bcx_exchange_stats(bcx_solver* s, int64_t* n, double* wait_us_mean, double* wait_us_max, double* total_us_mean,
                                  double* total_us_max, int32_t reset) {
  if (!s) return BCX_ERR_ARG;
  unsigned long long h[5] = {0, 0, 0, 0, 0};
  if (s->xseq) {
    BCX_HIP(hipSetDevice(s->cfg.device));
    BCX_HIP(hipStreamSynchronize(s->stream));
    BCX_HIP(hipMemcpy(h, s->xseq + 1, sizeof h, hipMemcpyDeviceToHost));
    if (reset) BCX_HIP(hipMemset(s->xseq + 1, 0, sizeof h));
  }
  const double per = h[0] ? 1.0 / (double)h[0] : 0.0, us = 1.0 / 100.0;   // wall_clock64: 100 MHz
  if (n) *n = (int64_t)h[0];
  if (wait_us_mean) *wait_us_mean = (double)h[1] * per * us;
  if (wait_us_max) *wait_us_max = (double)h[2] * us;
  if (total_us_mean) *total_us_mean = (double)h[3] * per * us;
  if (total_us_max) *total_us_max = (double)h[4] * us;
  return BCX_OK;
}

extern "C" int bcx_exchange_disable(bcx_solver* s) {
  if (!s) return BCX_ERR_ARG;
  s->exchange_ready = false;   // mappings stay until bcx_destroy; the host-driven step_scan / step_apply path is used again
  return BCX_OK;
}

extern "C" int bcx_build_poll(bcx_solver* s, int64_t* n_done, int32_t* need_exact, int32_t* limit) {
  if (!s) return BCX_ERR_ARG;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  if (h.halt == HALT_GRID_TIMEOUT) { s->err = "OMP step: grid barrier timed out"; return BCX_ERR_STATE; }
  if (h.halt == HALT_EXCHANGE_TIMEOUT) {
    // which peers' flags never arrived (recorded by the waiting lanes, csrc/resolve.hip mailbox_exchange)
    unsigned long long late[2] = {0, 0};
    std::string who;
    if (s->xseq && hipMemcpy(late, s->xseq + 6, sizeof late, hipMemcpyDeviceToHost) == hipSuccess) {
      for (int r = 0; r < s->cfg.world_size && r < 64; ++r)
        if (late[0] >> r & 1ull) who += (who.empty() ? "" : ", ") + std::to_string(r);
      (void)hipMemset(s->xseq + 6, 0, sizeof late);
    } else (void)hipGetLastError();
    s->err = "peer mailbox: rank " + std::to_string(s->cfg.rank) + " got no record from rank(s) " + (who.empty() ? "?" : who) +
             " within " + std::to_string(s->exchange_timeout_s) + " s (exchange #" + std::to_string(late[1]) + ")";
    return BCX_ERR_EXCHANGE;
  }
  if (n_done) *n_done = h.it;
  if (need_exact) *need_exact = (h.halt == HALT_NEED_EXACT);
  if (limit) *limit = h.limit;
  if (s->profile) {
    // launches issued after the state machine stopped (end of call, latch) return at once: keep only launches that
    // did the scan -- one that "read" the shard faster than 1.5x the HBM peak cannot have (a whole batch may consist
    // of such launches, e.g. everything enqueued after a latch, so the test is absolute, not relative to the batch)
    const double min_ms = (double)s->cfg.n_local * s->cfg.d * s->elem / 12.0e9;
    for (size_t i = 0; i < s->prof_used; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, s->prof_events[i].first, s->prof_events[i].second) != hipSuccess) continue;
      const int w = s->prof_weight[i];
      // (a launch of several iterations counts only when the call ran all of its iterations: one that stopped on the way
      //  -- latch, exact redo -- covered fewer than w)
      if (w > 1 && h.it != h.itrs) continue;
      if ((double)ms >= min_ms * w) { s->prof_ms += ms; s->prof_launches += w; }
    }
    s->prof_used = 0;
  }
  return BCX_OK;
}

extern "C" int bcx_build_trace(bcx_solver* s, int64_t* sel, double* err, int32_t* status, int64_t cap, int64_t* n_out) {
  if (!s) return BCX_ERR_ARG;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  const int64_t n = std::min<int64_t>(h.it, cap);
  if (n > 0) {
    if (sel) BCX_HIP(hipMemcpy(sel, s->tr_sel, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (err) BCX_HIP(hipMemcpy(err, s->tr_err, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (status) BCX_HIP(hipMemcpy(status, s->tr_status, (size_t)n * 4, hipMemcpyDeviceToHost));
  }
  if (n_out) *n_out = n;
  return BCX_OK;
}

// ---- read-out ------------------------------------------------------------------------------
extern "C" int bcx_active_count(bcx_solver* s, int64_t* k) {
  if (!s || !k) return BCX_ERR_ARG;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  *k = h.k;
  return BCX_OK;
}

extern "C" int bcx_get_weights(bcx_solver* s, int64_t* idx, double* w, int64_t cap, int64_t* k) {
  if (!s || !k) return BCX_ERR_ARG;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  const int64_t n = std::min<int64_t>(h.k, cap);
  if (n > 0) {
    if (idx) BCX_HIP(hipMemcpy(idx, s->act_idx, (size_t)n * 8, hipMemcpyDeviceToHost));
    if (w) BCX_HIP(hipMemcpy(w, s->act_w, (size_t)n * 8, hipMemcpyDeviceToHost));
  }
  *k = h.k;
  return BCX_OK;
}

extern "C" int bcx_error(bcx_solver* s, double* err) {
  if (!s || !err) return BCX_ERR_ARG;
  if (!s->finalized) { s->err = "solver not finalized"; return BCX_ERR_STATE; }
  int rc = bcx_launch_error_refresh(s);
  if (rc != BCX_OK) return rc;
  DevState h;
  if ((rc = read_state(s, &h))) return rc;
  *err = h.err;
  return BCX_OK;
}

extern "C" int bcx_reached_numeric_limit(bcx_solver* s, int32_t* limit) {
  if (!s || !limit) return BCX_ERR_ARG;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  *limit = h.limit;
  return BCX_OK;
}

extern "C" int bcx_reset(bcx_solver* s) {
  if (!s) return BCX_ERR_ARG;
  if (!s->finalized) return BCX_OK;
  // w = 0, reached_numeric_limit = False (snnls.py:18-20); b, norms and the matrix stay
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  h.k = 0; h.np = 0; h.hvalid = 1; h.omp_ill = 0; h.limit = 0; h.retried = 0; h.active = 0; h.halt = HALT_NONE; h.since_refresh = 0; h.exact_mode = 0;
  h.err = h.bnorm; h.nw = 1.0; h.it = 0; h.itrs = 0;
  BCX_HIP(hipMemcpy(s->st, &h, sizeof h, hipMemcpyHostToDevice));
  BCX_HIP(hipMemset(s->xw, 0, (size_t)s->cfg.d * 8));
  // error() with an empty list must give ||b|| computed the same way as after finalize
  return bcx_launch_error_refresh(s);
}

extern "C" int bcx_set_check_monotone(bcx_solver* s, int32_t on) {
  if (!s) return BCX_ERR_ARG;
  BCX_HIP(hipSetDevice(s->cfg.device));
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  h.no_monotone = on ? 0 : 1;
  BCX_HIP(hipMemcpy(s->st, &h, sizeof h, hipMemcpyHostToDevice));
  return BCX_OK;
}

extern "C" int bcx_optimize(bcx_solver* s, double tol, int32_t* accepted) {
  if (!s || !accepted) return BCX_ERR_ARG;
  if (!s->finalized) { s->err = "solver not finalized"; return BCX_ERR_STATE; }
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  if ((rc = ensure_slots(s, std::max<int64_t>(h.k, 1)))) return rc;
  if ((rc = bcx_ensure_gram(s, std::max<int64_t>(h.k, 1)))) return rc;
  if ((rc = bcx_launch_optimize(s, tol))) return rc;
  DevState h2;
  if ((rc = read_state(s, &h2))) return rc;
  if (h2.halt == HALT_GRID_TIMEOUT) { s->err = "optimize: grid barrier timed out"; return BCX_ERR_STATE; }
  *accepted = h2.limit ? 0 : 1;
  return BCX_OK;
}

// ---- introspection / measurement ---------------------------------------------------------------
extern "C" int bcx_get_vector(bcx_solver* s, int32_t which, double* out) {
  if (!s || !out) return BCX_ERR_ARG;
  const double* src = nullptr;
  switch (which) {
    case 0: src = s->b; break;
    case 1: src = s->xw; break;
    case 2: src = s->q64; break;
    case 3: src = s->q64 + s->ld64; break;
    default: return BCX_ERR_ARG;
  }
  BCX_HIP(hipStreamSynchronize(s->stream));
  BCX_HIP(hipMemcpy(out, src, (size_t)s->cfg.d * 8, hipMemcpyDeviceToHost));
  return BCX_OK;
}

extern "C" int bcx_get_norms(bcx_solver* s, int64_t begin, int64_t count, double* out) {
  if (!s || !out || begin < 0 || begin + count > s->cfg.n_local) return BCX_ERR_ARG;
  BCX_HIP(hipStreamSynchronize(s->stream));
  BCX_HIP(hipMemcpy(out, s->norms + begin, (size_t)count * 8, hipMemcpyDeviceToHost));
  return BCX_OK;
}

extern "C" int bcx_time_scan(bcx_solver* s, int32_t reps, int32_t exact, double* ms_per_launch, double* bytes_per_launch) {
  if (!s || reps < 1) return BCX_ERR_ARG;
  if (!s->finalized) { s->err = "solver not finalized"; return BCX_ERR_STATE; }
  BCX_HIP(hipSetDevice(s->cfg.device));
  // the scan only runs while the state machine is active: force it for the measurement
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  DevState forced = h;
  forced.active = 1;
  BCX_HIP(hipMemcpy(s->st, &forced, sizeof forced, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  BCX_HIP(hipEventCreate(&e0));
  BCX_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3 && rc == BCX_OK; ++i) rc = bcx_launch_scan(s, exact);
  if (rc == BCX_OK) (void)hipEventRecord(e0, s->stream);
  for (int i = 0; i < reps && rc == BCX_OK; ++i) rc = bcx_launch_scan(s, exact);
  if (rc != BCX_OK) {
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipMemcpy(s->st, &h, sizeof h, hipMemcpyHostToDevice);
    return rc;
  }
  BCX_HIP(hipEventRecord(e1, s->stream));
  BCX_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  BCX_HIP(hipEventElapsedTime(&ms, e0, e1));
  BCX_HIP(hipEventDestroy(e0));
  BCX_HIP(hipEventDestroy(e1));
  BCX_HIP(hipMemcpy(s->st, &h, sizeof h, hipMemcpyHostToDevice));
  if (ms_per_launch) *ms_per_launch = (double)ms / reps;
  const bool raw64 = exact && s->cfg.store_dtype != BCX_F64 && s->A64;
  if (bytes_per_launch) *bytes_per_launch = (double)s->cfg.n_local * s->cfg.d * (raw64 ? 8.0 : (double)s->elem);
  return BCX_OK;
}

// One N-way correlation scan with a caller-supplied query: arg-max_n An[n] . q (first maximum) and the
// exact fp64 score.  This is the select step of SparseVI on already projected vectors
// (sparsevi.py:49-56: corrs = vecs.dot(resid)/||vecs||/S, argmax).  Single-query algorithms only.
extern "C" int bcx_argmax_correlation(bcx_solver* s, const double* query_host, int64_t* idx, double* score) {
  if (!s || !query_host || !idx || !score) return BCX_ERR_ARG;
  if (!s->finalized) { s->err = "solver not finalized"; return BCX_ERR_STATE; }
  if (s->cfg.alg == BCX_ALG_GIGA) { s->err = "bcx_argmax_correlation needs a single-query solver (FW/OMP)"; return BCX_ERR_ARG; }
  BCX_HIP(hipSetDevice(s->cfg.device));
  const int d = s->cfg.d;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  std::vector<double> q64((size_t)s->ld64, 0.0);
  double qn = 0.0;
  for (int j = 0; j < d; ++j) { q64[j] = query_host[j]; qn += query_host[j] * query_host[j]; }
  BCX_HIP(hipMemcpy(s->q64, q64.data(), (size_t)s->ld64 * 8, hipMemcpyHostToDevice));
  if (s->cfg.store_dtype == BCX_F64) {
    std::vector<double> qs((size_t)s->ld, 0.0);
    for (int j = 0; j < d; ++j) qs[j] = query_host[j];
    BCX_HIP(hipMemcpy(s->qst, qs.data(), (size_t)s->ld * 8, hipMemcpyHostToDevice));
  } else {
    std::vector<float> qs((size_t)s->ld, 0.f);
    for (int j = 0; j < d; ++j) qs[j] = (float)query_host[j];
    BCX_HIP(hipMemcpy(s->qst, qs.data(), (size_t)s->ld * 4, hipMemcpyHostToDevice));
  }
  DevState forced = h;
  forced.active = 1; forced.exact_mode = 0; forced.qscale = sqrt(qn);
  double rec[BCX_REC_HDR];
  for (int attempt = 0; attempt < 2; ++attempt) {
    forced.exact_mode = attempt;
    BCX_HIP(hipMemcpy(s->st, &forced, sizeof forced, hipMemcpyHostToDevice));
    if ((rc = bcx_launch_scan(s, attempt))) return rc;
    if ((rc = bcx_launch_resolve(s, s->rec_local, attempt))) return rc;
    BCX_HIP(hipStreamSynchronize(s->stream));
    BCX_HIP(hipMemcpy(rec, s->rec_local, sizeof rec, hipMemcpyDeviceToHost));
    if (rec[3] != BCX_REC_OVERFLOW) break;
  }
  BCX_HIP(hipMemcpy(s->st, &h, sizeof h, hipMemcpyHostToDevice));
  if (rec[3] != BCX_REC_VALID) { s->err = "no rows to scan"; return BCX_ERR_STATE; }
  *idx = (int64_t)rec[1];
  *score = rec[0];
  return BCX_OK;
}

extern "C" int bcx_stats(bcx_solver* s, int64_t* exact_fallbacks, int64_t* candidates, int64_t* resolves) {
  if (!s) return BCX_ERR_ARG;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  if (exact_fallbacks) *exact_fallbacks = h.n_exact;
  if (candidates) *candidates = h.n_cand;
  if (resolves) *resolves = h.n_resolved;
  return BCX_OK;
}

// OMP step diagnostics since construction: out4 = {steps, columns that left the passive set, from-scratch re-solves,
// columns that entered beyond the selected one}
extern "C" int bcx_omp_stats(bcx_solver* s, int64_t* out4) {
  if (!s || !out4) return BCX_ERR_ARG;
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  for (int i = 0; i < 4; ++i) out4[i] = h.n_omp[i];
  out4[2] += s->opt_fallbacks;          // optimize() calls that needed the refined solve count as re-solves too
  return BCX_OK;
}

extern "C" int bcx_profile_scan(bcx_solver* s, int32_t on) {
  if (!s) return BCX_ERR_ARG;
  s->profile = on != 0;
  s->prof_every = on > 1 ? on : 1;      // on = N > 1: time every N-th scan launch
  s->prof_tick = 0;
  s->prof_ms = 0.0;
  s->prof_launches = 0;
  s->prof_used = 0;
  return BCX_OK;
}

extern "C" int bcx_profile_read(bcx_solver* s, double* scan_ms_total, int64_t* scan_launches) {
  if (!s) return BCX_ERR_ARG;
  if (scan_ms_total) *scan_ms_total = s->prof_ms;
  if (scan_launches) *scan_launches = s->prof_launches;
  return BCX_OK;
}

// dev builds (-DBCX_TIMING): phase time stamps (100 MHz ticks) of the last merged tail; not in include/bcx.h
extern "C" int bcx_debug_stamps(bcx_solver* s, long long* out32) {
  DevState h;
  int rc = read_state(s, &h);
  if (rc != BCX_OK) return rc;
  for (int i = 0; i < 32; ++i) out32[i] = h.dbg_t[i];
  return BCX_OK;
}

// dev builds: raw copy of the weight back-up buffer (debug digests of nnls_grid.hip with -DBCX_DEBUG_GRID)
extern "C" int bcx_debug_wbak(bcx_solver* s, double* out, int32_t count) {
  if (!s || !s->nn_wbak || count > s->gram_cap) return BCX_ERR_ARG;
  BCX_HIP(hipDeviceSynchronize());
  BCX_HIP(hipMemcpy(out, s->nn_wbak, (size_t)count * sizeof(double), hipMemcpyDeviceToHost));
  return BCX_OK;
}
