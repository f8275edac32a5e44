"""ctypes binding of libbcx.so (the C ABI in include/bcx.h) and the host-side driver of one
row shard.  There is NO CPU fallback: if the library is missing or no GPU is usable every
solver constructor raises."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# a source checkout builds the library in-tree (make -C bayesian-coresets_amd); an installed package carries it inside (setup.py)
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libbcx.so")
if not os.path.exists(LIB_PATH) and os.path.exists(os.path.join(_HERE, "_lib", "libbcx.so")):
    LIB_PATH = os.path.join(_HERE, "_lib", "libbcx.so")

ALG_GIGA, ALG_FW, ALG_OMP = 0, 1, 2
F32, F64, F16 = 0, 1, 2
OK, ERR_ARG, ERR_HIP, ERR_ZERO_ROW, ERR_ZERO_B, ERR_NOMEM, ERR_STATE, ERR_EXCHANGE, ERR_TIMEOUT = 0, -1, -2, -3, -4, -5, -6, -7, -8
IT_OK, IT_FAIL_SELECT, IT_FAIL_REWEIGHT, IT_FAIL_MONOTONE = 0, 1, 2, 3
REC_HDR = 4
LOAD_CENTER_ROWS = 1
MAX_ROW_LENGTH = 1048576      # BCX_MAX_ROW_LENGTH of include/bcx.h (tests/test_abi.py keeps the two equal)
CHUNK_ROWS = 1024

# every symbol include/bcx.h declares (checked by tests/test_abi.py)
SYMBOLS = (
    "bcx_create", "bcx_destroy", "bcx_last_error", "bcx_set_stream", "bcx_load_rows", "bcx_chunk_sums", "bcx_export_chunk_sums",
    "bcx_finalize", "bcx_build_begin", "bcx_step_scan", "bcx_step_apply", "bcx_build_enqueue", "bcx_build_poll",
    "bcx_step_scan_exact", "bcx_build_trace", "bcx_active_count", "bcx_get_weights", "bcx_error", "bcx_optimize",
    "bcx_reset", "bcx_reached_numeric_limit", "bcx_get_vector", "bcx_get_norms", "bcx_argmax_correlation", "bcx_time_scan",
    "bcx_stats", "bcx_profile_scan", "bcx_profile_read", "bcx_version",
    "bcx_project_write", "bcx_project_colsum", "bcx_project_select", "bcx_project_last_error",
    "bcx_build_enqueue_exact", "bcx_exchange_export", "bcx_exchange_attach", "bcx_exchange_probe", "bcx_exchange_disable", "bcx_exchange_set_timeout",
    "bcx_set_check_monotone", "bcx_project_profile", "bcx_project_profile_read", "bcx_exchange_stats", "bcx_load_rows_flags", "bcx_project_write_raw", "bcx_omp_stats", "bcx_project_select_ws", "bcx_project_select_scratch_bytes",
    "bcx_project_moments", "bcx_project_colsum_moments", "bcx_project_moments_scratch_bytes",
    "bcx_project_colsum_moments_scratch_bytes", "bcx_gram", "bcx_gram_scratch_bytes", "bcx_gram_check",
    "bcx_project_colsum_moments_at", "bcx_project_points_colsum_moments", "bcx_linreg_posterior_draw", "bcx_sparsevi_adam_step",
    "bcx_linreg_posterior_apply", "bcx_linreg_posterior_apply_ok",
    "bcx_linreg_posterior_factor", "bcx_linreg_posterior_factor_scratch_bytes", "bcx_linreg_posterior_factor_status",
    "bcx_linreg_posterior_draw_factored",
    "bcx_sparsevi_adam_step_ws", "bcx_sparsevi_adam_scratch_bytes", "bcx_standard_normal", "bcx_column_means",
    "bcx_center_rows", "bcx_row_sumsq", "bcx_project_write_points",
    "bcx_laplace_sampler", "bcx_laplace_sampler_ok", "bcx_laplace_sampler_lds_bytes",
)


class Config(ctypes.Structure):
    _fields_ = [
        ("alg", ctypes.c_int32), ("store_dtype", ctypes.c_int32), ("keep_exact_rows", ctypes.c_int32),
        ("device", ctypes.c_int32), ("d", ctypes.c_int32), ("world_size", ctypes.c_int32),
        ("rank", ctypes.c_int32), ("refresh_every", ctypes.c_int32),
        ("n_local", ctypes.c_int64), ("n_global", ctypes.c_int64), ("row_offset", ctypes.c_int64),
    ]


_lib = None


def ensure_ipc_mode():
    """The ONE place that sets the IPC mode: multi-process GPU work on this platform (RCCL, the peer mailbox's
    hipIpcGetMemHandle) needs dmabuf IPC, i.e. HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the HSA runtime
    initialises.  Called by load(), tests/conftest.py, bench.py and __graft_entry__.py.  A different value chosen by the
    user is respected (BCX_KEEP_IPC_MODE=1) and a setting that can no longer take effect is reported."""
    import logging
    cur = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    if cur == "0":
        return True
    if os.environ.get("BCX_KEEP_IPC_MODE"):
        return False
    initialised = False
    try:
        import sys
        t = sys.modules.get("torch")
        initialised = bool(t is not None and t.cuda.is_initialized())
    except Exception:
        pass
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    if initialised:
        logging.getLogger().warning("bayesiancoresets_amd: the HIP runtime was initialised before HSA_ENABLE_IPC_MODE_LEGACY=0 "
                                    "could be set (it was %r); peer mailboxes / RCCL between processes may fail -- export it "
                                    "before the first GPU call", cur)
    return not initialised


def load():
    """Load libbcx.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libbcx.so not found at %s -- build it with `make -C bayesian-coresets_amd` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback." % LIB_PATH)
    ensure_ipc_mode()
    # PyTorch-ROCm bundles its own HIP / HSA runtime; libbcx.so must bind to THAT copy (same SONAME, already loaded),
    # not pull the system one in next to it -- two HIP runtimes in one process fight over the device ("No HIP GPUs are
    # available" in whichever initialises second).  So torch is imported before the library is opened
    # (BCX_NO_TORCH_PRELOAD=1 skips this for embedders without torch).
    if not os.environ.get("BCX_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, dbl = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_double
    P = ctypes.POINTER
    lib.bcx_version.restype = ctypes.c_char_p
    lib.bcx_last_error.restype = ctypes.c_char_p
    lib.bcx_last_error.argtypes = [vp]
    sigs = {
        "bcx_create": [P(Config), P(vp)],
        "bcx_destroy": [vp],
        "bcx_set_stream": [vp, vp],
        "bcx_load_rows": [vp, vp, i32, i32, i64, i64, i64],
        "bcx_load_rows_flags": [vp, vp, i32, i32, i64, i64, i64, i32],
        "bcx_chunk_sums": [vp, P(vp), P(i64), P(i64)],
        "bcx_export_chunk_sums": [vp, vp, i64],
        "bcx_finalize": [vp, vp, vp, i64],
        "bcx_build_begin": [vp, i64, dbl, P(i32)],
        "bcx_step_scan": [vp, vp],
        "bcx_step_scan_exact": [vp, vp],
        "bcx_step_apply": [vp, vp],
        "bcx_build_enqueue": [vp, i64],
        "bcx_build_poll": [vp, P(i64), P(i32), P(i32)],
        "bcx_build_trace": [vp, vp, vp, vp, i64, P(i64)],
        "bcx_active_count": [vp, P(i64)],
        "bcx_get_weights": [vp, vp, vp, i64, P(i64)],
        "bcx_error": [vp, P(dbl)],
        "bcx_optimize": [vp, dbl, P(i32)],
        "bcx_reset": [vp],
        "bcx_set_check_monotone": [vp, i32],
        "bcx_reached_numeric_limit": [vp, P(i32)],
        "bcx_get_vector": [vp, i32, vp],
        "bcx_get_norms": [vp, i64, i64, vp],
        "bcx_argmax_correlation": [vp, vp, P(i64), P(dbl)],
        "bcx_time_scan": [vp, i32, i32, P(dbl), P(dbl)],
        "bcx_stats": [vp, P(i64), P(i64), P(i64)],
        "bcx_omp_stats": [vp, vp],
        "bcx_profile_scan": [vp, i32],
        "bcx_profile_read": [vp, P(dbl), P(i64)],
        "bcx_build_enqueue_exact": [vp],
        "bcx_exchange_export": [vp, vp, i32],
        "bcx_exchange_attach": [vp, vp, i32, dbl],
        "bcx_exchange_probe": [vp, P(i32)],
        "bcx_exchange_disable": [vp],
        "bcx_exchange_set_timeout": [vp, dbl],
        "bcx_exchange_stats": [vp, P(i64), P(dbl), P(dbl), P(dbl), P(dbl), i32],
    }
    proj_common = [vp, i32, vp, i64, i64, i32, i32, vp, i32, i32, dbl]
    sigs["bcx_project_write"] = proj_common + [vp, i64, vp]
    sigs["bcx_project_write_raw"] = proj_common + [vp, i64]
    sigs["bcx_project_write_points"] = proj_common + [vp, i64, i32]
    sigs["bcx_project_colsum"] = proj_common + [vp, vp]
    sigs["bcx_project_select"] = proj_common + [vp, dbl, vp, vp]
    sigs["bcx_project_select_ws"] = proj_common + [vp, dbl, vp, vp, i64]
    sigs["bcx_project_profile"] = [i32]
    sigs["bcx_project_profile_read"] = [P(dbl), P(i64), P(dbl)]
    sigs["bcx_project_moments"] = [vp, vp, i64, i64, i32, vp, i64, vp, i64]
    sigs["bcx_project_colsum_moments"] = [vp, vp, i64, i32, i32, vp, i32, i32, dbl, vp, vp]
    sigs["bcx_project_colsum_moments_at"] = [vp, vp, i64, i32, i32, vp, i32, i32, dbl, vp, vp, vp]
    sigs["bcx_project_points_colsum_moments"] = [vp, vp, i64, i64, i32, i32, vp, i32, i32, dbl, vp, i64, vp, i64, i32, vp, vp, vp]
    sigs["bcx_linreg_posterior_draw"] = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, dbl, vp, vp, i32, vp, vp]
    sigs["bcx_linreg_posterior_apply"] = [vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, dbl, vp, vp, i32, vp, vp, vp]
    sigs["bcx_linreg_posterior_apply_ok"] = [i32, i32]
    sigs["bcx_sparsevi_adam_step"] = [vp, i32, i32, vp, dbl, vp, i64, vp, vp, vp, vp, i32, dbl, dbl, dbl, vp, i32]
    sigs["bcx_gram"] = [vp, vp, i32, i32, i64, vp, i64, vp, i64]
    sigs["bcx_gram_check"] = [vp, vp]
    sigs["bcx_linreg_posterior_factor"] = [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, dbl, vp, i64, vp, i64, vp, vp]
    sigs["bcx_linreg_posterior_factor_status"] = [vp, i32, vp]
    sigs["bcx_standard_normal"] = [vp, ctypes.c_uint64, ctypes.c_uint64, i64, vp]
    sigs["bcx_column_means"] = [vp, vp, i32, i32, i32, i64, vp, i64]
    sigs["bcx_center_rows"] = [vp, vp, i64, i32, i64]
    sigs["bcx_laplace_sampler"] = [vp, i32, i32, i32, vp, vp, i64, vp, i32, dbl, i32, vp, vp, i32, i32, vp, vp, vp]
    sigs["bcx_laplace_sampler_ok"] = [i32, i32]
    lib.bcx_laplace_sampler_lds_bytes.restype = ctypes.c_int64
    lib.bcx_laplace_sampler_lds_bytes.argtypes = [i32, i32]
    sigs["bcx_row_sumsq"] = [vp, vp, i64, i32, i64, vp]
    sigs["bcx_linreg_posterior_draw_factored"] = [vp, i32, i32, vp, i64, vp, vp, vp, i32, vp, vp]
    sigs["bcx_sparsevi_adam_step_ws"] = [vp, i32, i32, vp, dbl, vp, i64, vp, vp, vp, vp, i32, dbl, dbl, dbl, vp, i32, vp, i64]
    lib.bcx_linreg_posterior_factor_scratch_bytes.restype = ctypes.c_int64
    lib.bcx_linreg_posterior_factor_scratch_bytes.argtypes = [i32]
    lib.bcx_sparsevi_adam_scratch_bytes.restype = ctypes.c_int64
    lib.bcx_sparsevi_adam_scratch_bytes.argtypes = [i32, i32]
    lib.bcx_gram_scratch_bytes.restype = ctypes.c_int64
    lib.bcx_gram_scratch_bytes.argtypes = [i32, i32]
    lib.bcx_project_moments_scratch_bytes.restype = ctypes.c_int64
    lib.bcx_project_moments_scratch_bytes.argtypes = [i64, i32]
    lib.bcx_project_colsum_moments_scratch_bytes.restype = ctypes.c_int64
    lib.bcx_project_colsum_moments_scratch_bytes.argtypes = [i32, i32]
    lib.bcx_project_select_scratch_bytes.restype = ctypes.c_int64
    lib.bcx_project_select_scratch_bytes.argtypes = [i32, i64, i32]
    lib.bcx_project_last_error.restype = ctypes.c_char_p
    lib.bcx_project_last_error.argtypes = []
    for name, args in sigs.items():
        if name.endswith("_scratch_bytes"):
            continue
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int
    _lib = lib
    return lib


class EngineError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bcx error %d: %s" % (code, msg))
        self.code = code


def _current_stream_ptr():
    """Stream torch is currently issuing work on (so torch.distributed collectives and
    torch.cuda events order correctly with the engine's kernels); NULL stream without torch."""
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.current_stream().cuda_stream)
    except Exception:
        pass
    return 0


class Engine(object):
    """One row shard of the device solver.  Thin: every method is one or two ABI calls."""

    def __init__(self, alg, n_local, d, n_global=None, row_offset=0, rank=0, world_size=1, device=0,
                 store_dtype=F32, keep_exact_rows=True, refresh_every=0):
        self.lib = load()
        self.h = ctypes.c_void_p()
        self.cfg = Config(alg=alg, store_dtype=store_dtype, keep_exact_rows=int(bool(keep_exact_rows)),
                          device=device, d=d, world_size=world_size, rank=rank, refresh_every=refresh_every,
                          n_local=n_local, n_global=n_local if n_global is None else n_global,
                          row_offset=row_offset)
        rc = self.lib.bcx_create(ctypes.byref(self.cfg), ctypes.byref(self.h))
        if rc != OK:
            msg = self.lib.bcx_last_error(None).decode()
            self.h = ctypes.c_void_p()
            raise EngineError(rc, msg)
        self.d, self.n_local = d, n_local
        self.rec_len = d + REC_HDR
        self.use_current_stream()

    # -- plumbing ---------------------------------------------------------
    def _check(self, rc):
        if rc != OK:
            raise EngineError(rc, self.lib.bcx_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.bcx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def use_current_stream(self):
        self._check(self.lib.bcx_set_stream(self.h, ctypes.c_void_p(_current_stream_ptr())))

    # -- ingest -----------------------------------------------------------
    def load_host_rows(self, rows, row_begin=0, center=False):
        """rows: C-contiguous (n, >=d) float32/float64 ndarray view (row stride in elements = ld).
        center: the rows are raw log-likelihoods, subtract each row's mean during the pass (projector.py:21)."""
        assert rows.ndim == 2 and rows.strides[1] == rows.itemsize
        dt = F64 if rows.dtype == np.float64 else F32
        ld = rows.strides[0] // rows.itemsize
        self._check(self.lib.bcx_load_rows_flags(self.h, ctypes.c_void_p(rows.ctypes.data), 0, dt, row_begin,
                                                 rows.shape[0], ld, LOAD_CENTER_ROWS if center else 0))

    def load_device_rows(self, ptr, n, ld, is_f64, row_begin=0, center=False):
        self._check(self.lib.bcx_load_rows_flags(self.h, ctypes.c_void_p(ptr), 1, F64 if is_f64 else F32, row_begin, n, ld,
                                                 LOAD_CENTER_ROWS if center else 0))

    def chunk_sums_info(self):
        p, n, r = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.bcx_chunk_sums(self.h, ctypes.byref(p), ctypes.byref(n), ctypes.byref(r)))
        return p.value, n.value, r.value

    def export_chunk_sums(self, dst_ptr, cap_chunks):
        self._check(self.lib.bcx_export_chunk_sums(self.h, ctypes.c_void_p(dst_ptr), cap_chunks))

    def finalize(self, b=None, gathered_ptr=None, n_gathered=0):
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float64)
            assert b.shape == (self.d,)
            bp = ctypes.c_void_p(b.ctypes.data)
        rc = self.lib.bcx_finalize(self.h, bp, ctypes.c_void_p(gathered_ptr) if gathered_ptr else None, n_gathered)
        return rc

    # -- tensor-level protocol used by sharded.ShardedSolver -----------------
    def tensor_device(self):
        import torch
        return torch.device("cuda", self.cfg.device)

    def load_rows_any(self, rows, row_begin=0, center=False):
        try:
            import torch
            if isinstance(rows, torch.Tensor):
                if rows.device.type == "cuda":
                    assert rows.stride(1) == 1
                    self.load_device_rows(rows.data_ptr(), rows.shape[0], rows.stride(0),
                                          rows.element_size() == 8, row_begin, center)
                    return
                rows = rows.numpy()
        except ImportError:
            pass
        self.load_host_rows(np.ascontiguousarray(rows), row_begin, center)

    def export_chunk_sums_tensor(self, t, cap_chunks):
        self.export_chunk_sums(t.data_ptr(), cap_chunks)

    def finalize_any(self, b, gathered, n_gathered):
        return self.finalize(b, gathered.data_ptr() if gathered is not None else None, n_gathered)

    def step_scan_tensor(self, send, exact=False):
        self.step_scan(send.data_ptr(), exact)

    def step_apply_tensor(self, recv):
        self.step_apply(recv.data_ptr())

    # -- build ------------------------------------------------------------
    def build_begin(self, itrs, tol):
        skip = ctypes.c_int32()
        self._check(self.lib.bcx_build_begin(self.h, itrs, tol, ctypes.byref(skip)))
        return bool(skip.value)

    def step_scan(self, send_ptr=None, exact=False):
        fn = self.lib.bcx_step_scan_exact if exact else self.lib.bcx_step_scan
        self._check(fn(self.h, ctypes.c_void_p(send_ptr) if send_ptr else None))

    def step_apply(self, recv_ptr=None):
        self._check(self.lib.bcx_step_apply(self.h, ctypes.c_void_p(recv_ptr) if recv_ptr else None))

    def enqueue(self, itrs):
        self._check(self.lib.bcx_build_enqueue(self.h, itrs))

    def enqueue_exact(self):
        self._check(self.lib.bcx_build_enqueue_exact(self.h))

    # -- peer mailbox (device-side record exchange between row shards) -------
    IPC_HANDLE_BYTES = 64

    def exchange_export(self):
        buf = ctypes.create_string_buffer(self.IPC_HANDLE_BYTES)
        self._check(self.lib.bcx_exchange_export(self.h, buf, self.IPC_HANDLE_BYTES))
        return buf.raw

    def exchange_attach(self, handles, timeout_s=0.0):
        blob = b"".join(handles)
        assert len(blob) == self.IPC_HANDLE_BYTES * len(handles)
        self._check(self.lib.bcx_exchange_attach(self.h, blob, self.IPC_HANDLE_BYTES, float(timeout_s)))

    def exchange_probe(self):
        res = ctypes.c_int32()
        self._check(self.lib.bcx_exchange_probe(self.h, ctypes.byref(res)))
        return res.value

    def exchange_set_timeout(self, timeout_s):
        self._check(self.lib.bcx_exchange_set_timeout(self.h, float(timeout_s)))

    def exchange_stats(self, reset=False):
        """Device-side timing of the record exchanges since the last reset (microseconds)."""
        n = ctypes.c_int64()
        v = [ctypes.c_double() for _ in range(4)]
        self._check(self.lib.bcx_exchange_stats(self.h, ctypes.byref(n), *[ctypes.byref(x) for x in v], int(bool(reset))))
        return {"exchanges": n.value, "wait_us_mean": v[0].value, "wait_us_max": v[1].value,
                "total_us_mean": v[2].value, "total_us_max": v[3].value}

    def exchange_disable(self):
        self._check(self.lib.bcx_exchange_disable(self.h))

    def poll(self):
        n, ne, lim = ctypes.c_int64(), ctypes.c_int32(), ctypes.c_int32()
        self._check(self.lib.bcx_build_poll(self.h, ctypes.byref(n), ctypes.byref(ne), ctypes.byref(lim)))
        return n.value, bool(ne.value), bool(lim.value)

    def trace(self, cap):
        sel = np.empty(cap, dtype=np.int64)
        err = np.empty(cap, dtype=np.float64)
        status = np.empty(cap, dtype=np.int32)
        n = ctypes.c_int64()
        self._check(self.lib.bcx_build_trace(self.h, sel.ctypes.data, err.ctypes.data, status.ctypes.data, cap,
                                             ctypes.byref(n)))
        return sel[:n.value], err[:n.value], status[:n.value]

    def run_build(self, itrs, tol):
        """Single-shard build(): enqueue everything, fall back to the exact scan for the rare
        iterations whose candidate window overflowed.  Returns (sel, err, status) of this call."""
        if self.build_begin(itrs, tol):
            return None
        self.enqueue(itrs)
        while True:
            done, need_exact, limit = self.poll()
            if not need_exact:
                break
            self.step_scan(exact=True)
            self.step_apply()
            done, need_exact, limit = self.poll()
            if need_exact:
                raise EngineError(ERR_STATE, "exact scan did not resolve the iteration")
            if done < itrs and not limit:
                self.enqueue(itrs - done)
        return self.trace(itrs)

    # -- read-out ---------------------------------------------------------
    def sparse_weights(self):
        k = ctypes.c_int64()
        self._check(self.lib.bcx_active_count(self.h, ctypes.byref(k)))
        idx = np.empty(k.value, dtype=np.int64)
        w = np.empty(k.value, dtype=np.float64)
        if k.value:
            self._check(self.lib.bcx_get_weights(self.h, idx.ctypes.data, w.ctypes.data, k.value, ctypes.byref(k)))
        return idx, w

    def error(self):
        e = ctypes.c_double()
        self._check(self.lib.bcx_error(self.h, ctypes.byref(e)))
        return e.value

    def optimize(self, tol):
        acc = ctypes.c_int32()
        self._check(self.lib.bcx_optimize(self.h, tol, ctypes.byref(acc)))
        return bool(acc.value)

    def reset(self):
        self._check(self.lib.bcx_reset(self.h))

    def set_check_monotone(self, on):
        self._check(self.lib.bcx_set_check_monotone(self.h, int(bool(on))))

    def reached_numeric_limit(self):
        v = ctypes.c_int32()
        self._check(self.lib.bcx_reached_numeric_limit(self.h, ctypes.byref(v)))
        return bool(v.value)

    def vector(self, which):
        out = np.empty(self.d, dtype=np.float64)
        self._check(self.lib.bcx_get_vector(self.h, which, out.ctypes.data))
        return out

    def norms(self, begin=0, count=None):
        count = self.n_local - begin if count is None else count
        out = np.empty(count, dtype=np.float64)
        if count:
            self._check(self.lib.bcx_get_norms(self.h, begin, count, out.ctypes.data))
        return out

    def argmax_correlation(self, query):
        q = np.ascontiguousarray(query, dtype=np.float64)
        assert q.shape == (self.d,)
        idx, score = ctypes.c_int64(), ctypes.c_double()
        self._check(self.lib.bcx_argmax_correlation(self.h, ctypes.c_void_p(q.ctypes.data), ctypes.byref(idx),
                                                    ctypes.byref(score)))
        return idx.value, score.value

    # -- measurement ------------------------------------------------------
    def time_scan(self, reps=20, exact=False):
        ms, by = ctypes.c_double(), ctypes.c_double()
        self._check(self.lib.bcx_time_scan(self.h, reps, int(exact), ctypes.byref(ms), ctypes.byref(by)))
        return ms.value, by.value

    def stats(self):
        a, b, c = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.bcx_stats(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return {"exact_fallbacks": a.value, "candidates": b.value, "resolves": c.value}

    def omp_stats(self):
        out = (ctypes.c_int64 * 4)()
        self._check(self.lib.bcx_omp_stats(self.h, out))
        return {"steps": out[0], "columns_left": out[1], "resolves": out[2], "extra_entered": out[3]}

    def profile(self, on):
        self._check(self.lib.bcx_profile_scan(self.h, int(on)))

    def profile_read(self):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        self._check(self.lib.bcx_profile_read(self.h, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value


def device_row_sumsq(rows):
    """Sum of squares of every row of a device tensor (N x S fp64, unit column stride) as an ndarray: csrc/proj.hip
    row_sumsq_kernel (hilbert.py:19-22 drops the rows where it is zero)."""
    import torch
    lib = load()
    if rows.dtype != torch.float64 or rows.stride(1) != 1:
        rows = rows.to(torch.float64).contiguous()
    out = torch.empty(rows.shape[0], dtype=torch.float64, device=rows.device)
    with torch.cuda.device(rows.device):
        rc = lib.bcx_row_sumsq(int(torch.cuda.current_stream(rows.device).cuda_stream), rows.data_ptr(), rows.shape[0], rows.shape[1],
                               rows.stride(0), out.data_ptr())
    if rc != 0:
        raise EngineError(rc, lib.bcx_project_last_error().decode())
    return out.cpu().numpy()


def device_centred_copy(rows):
    """rows - rows.mean(axis=1)[:, None] for a device tensor (N x S fp64) as a NEW tensor: a device copy, then csrc/proj.hip
    center_kernel in place (projector.py:21)."""
    import torch
    lib = load()
    out = rows.to(torch.float64).contiguous().clone()
    with torch.cuda.device(out.device):
        rc = lib.bcx_center_rows(int(torch.cuda.current_stream(out.device).cuda_stream), out.data_ptr(), out.shape[0], out.shape[1], out.stride(0))
    if rc != 0:
        raise EngineError(rc, lib.bcx_project_last_error().decode())
    return out
