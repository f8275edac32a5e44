"""Device-backed sparse non-negative least squares solvers.

Same constructor and methods as the reference base class (bayesiancoresets/snnls/snnls.py:8-106):
``SparseNNLS(A, b)`` with ``A`` of shape d x N (columns = data points) and ``b`` of length d;
``build(itrs)``, ``weights()``, ``error()``, ``size()``, ``optimize()``, ``reset()``; attributes
``.A .b .w .reached_numeric_limit``.  The arithmetic runs in libbcx.so on the GPU; this class only
moves data across the C ABI and maps status codes back onto the reference's warnings, exceptions
and latch.  Extra keyword-only options (device, dtype, keep_exact_rows) default to the
reference-compatible configuration.
"""
import numpy as np

from .. import util
from ..util.errors import NumericalPrecisionError
from ..util.log import object_logger
from .. import _native as nat


def _as_row_matrix(A):
    """Return (kind, rows): the N x d row-major view of the d x N argument without copying when
    the caller passed ``vecs.T`` of a C-contiguous array (hilbert.py:24 does exactly that)."""
    try:
        import torch
        if isinstance(A, torch.Tensor):
            rows = A.t()
            if rows.dtype not in (torch.float32, torch.float64):
                rows = rows.to(torch.float64)      # the ingest kernels read fp32 or fp64 (half / integer tensors are converted)
            if not rows.is_contiguous():
                rows = rows.contiguous()
            return ("torch", rows)
    except ImportError:
        pass
    A = np.asarray(A)
    rows = A.T
    if rows.dtype not in (np.float32, np.float64):
        rows = rows.astype(np.float64)
    if not rows.flags.c_contiguous:
        rows = np.ascontiguousarray(rows)
    return ("numpy", rows)


def warn_failed_steps(log, status, latched=False):
    """The WARNING lines of the reference's except-branch (snnls.py:63-72), one group per failed loop iteration:
    the caught NumericalPrecisionError, then either the retry notice or -- for the failure that set the latch,
    which is the last one of a build() that ended latched -- the numeric-limit notice."""
    failed = np.flatnonzero(status != nat.IT_OK)
    for n, i in enumerate(failed):
        what = {nat.IT_FAIL_SELECT: "select failed (cdirnrm < TOL)",
                nat.IT_FAIL_REWEIGHT: "reweight step lost precision",
                nat.IT_FAIL_MONOTONE: "Error not monotone"}[int(status[i])]
        log.warning("numerical precision error: " + what + " at loop iteration " + str(int(i)))
        if latched and n == len(failed) - 1:
            log.warning("iterative step failed a second time. Assuming numeric limit reached.")
        else:
            log.warning("iterative step failed. Stabilizing and retrying...")


class SparseNNLS(object):
    """The reference's solver surface (snnls.py:8-106).  ``isinstance(x, bc.snnls.SparseNNLS)`` holds for every
    solver of this package.  The loop in ``build`` below drives *host-side* subclasses that supply ``_select`` /
    ``_reweight`` themselves (the sampling baselines, user subclasses); the greedy solvers (``DeviceSparseNNLS``)
    override every method and run the same state machine on the GPU."""

    def __init__(self, A, b, check_error_monotone=True):
        self.alg_name, self.log = object_logger(self)
        self.A, self.b = A, b
        self.check_error_monotone = check_error_monotone
        self.reached_numeric_limit = False
        self.w = np.zeros(A.shape[1])

    def reset(self):
        self.w = np.zeros(self.A.shape[1])
        self.reached_numeric_limit = False

    def size(self):
        return (self.w > 0).sum()

    def weights(self):
        return self.w.copy()

    def error(self):
        return np.sqrt(((self.A.dot(self.w) - self.b) ** 2).sum())

    def _state_suffix(self):
        return "size = " + str(self.size()) + ", error = " + str(self.error())

    def _guarded_step(self):
        """One _select + _reweight.  Returns True when the step passed the monotone check (the only outcome that
        clears the retry flag, snnls.py:56-62), False when it was accepted unchecked; raises
        NumericalPrecisionError after restoring the weights when the error grew."""
        checked = bool(self.check_error_monotone) and self.size() > 0   # sampled before _reweight changes it (:44)
        if checked:
            err_before, w_before = self.error(), self.w.copy()
        self._reweight(self._select())
        if checked:
            err_after = self.error()
            if err_after > err_before:
                self.w = w_before
                raise NumericalPrecisionError("Error not monotone: curr error = " + str(err_after)
                                              + " prev error = " + str(err_before))
        return checked

    def build(self, itrs):
        if self.reached_numeric_limit:                                    # snnls.py:32-34
            self.log.warning("the numeric limit was already reached; returning. " + self._state_suffix())
            return
        if self.A.size == 0:                                              # snnls.py:36-38
            self.log.warning("there are no data, returning.")
            return
        strikes = 0           # failed steps since the last checked success; the second one latches (:63-72)
        for _ in range(itrs):
            try:
                if self._guarded_step():
                    strikes = 0
            except NumericalPrecisionError as e:
                self.log.warning("numerical precision error: " + str(e))
                strikes += 1
                if strikes > 1:
                    self.log.warning("iterative step failed a second time. Assuming numeric limit reached.")
                    self.reached_numeric_limit = True
                    break
                self.log.warning("iterative step failed. Stabilizing and retrying...")
                self._stabilize()
        if self.reached_numeric_limit:                                    # snnls.py:77-78
            self.log.warning("the numeric limit has been reached. No more points will be added. " + self._state_suffix())

    def optimize(self):
        """Host re-solve of the weights on the current support (snnls.py:82-97) for host-side subclasses."""
        from scipy.optimize import nnls
        cost_before, w_before = self.error(), self.w.copy()
        support = self.w > 0
        self.w[support] = nnls(self.A[:, support], self.b, maxiter=100 * self.A.shape[1])[0]
        cost_after = self.error()
        if cost_after > cost_before * (1.0 + util.TOL):
            self.log.warning("self.optimize() returned a solution with increasing error. Numeric limit possibly "
                             "reached: preverr = " + str(cost_before) + " err = " + str(cost_after) + ".")
            self.w = w_before
            self.reached_numeric_limit = True

    def _stabilize(self):
        pass

    def _select(self):
        raise NotImplementedError

    def _reweight(self, f):
        raise NotImplementedError


class DeviceSparseNNLS(SparseNNLS):
    """Engine-backed solver: GIGA / FrankWolfe / OrthoPursuit derive from this.  Nothing of the host loop above is
    used -- build(), error(), optimize(), reset() and the weights all live in libbcx.so."""
    _ALG = None  # set by subclasses

    def __init__(self, A, b, check_error_monotone=True, *, device=0, dtype="float32", keep_exact_rows=True,
                 center_rows=False):
        """center_rows: the columns of ``A`` are raw log-likelihood vectors; the constructor pass subtracts each one's
        mean over the d samples (projector.py:21) while it forms the norms -- ``.A`` then reads as the centred matrix."""
        self.alg_name, self.log = object_logger(self)
        self._A_arg, self._A_centred, self._center_rows = A, None, bool(center_rows)
        self._b_arg = b
        self._w_cache = None
        self._eng = None
        self._check_monotone = bool(check_error_monotone)
        if self._ALG is None:
            raise NotImplementedError("DeviceSparseNNLS is abstract; use GIGA, FrankWolfe or OrthoPursuit")
        kind, rows = _as_row_matrix(A)
        self._N, self._d = int(rows.shape[0]), int(rows.shape[1])
        if self._d > nat.MAX_ROW_LENGTH:
            raise ValueError(self.alg_name + ".__init__(): vectors of length %d exceed the engine's row-length limit of %d "
                             "(BCX_MAX_ROW_LENGTH, include/bcx.h); reduce the projection dimension" % (self._d, nat.MAX_ROW_LENGTH))
        dt = str(dtype)
        store = nat.F64 if dt in ("float64", "f64", "double") else (nat.F16 if dt in ("float16", "f16", "half") else nat.F32)
        eng = nat.Engine(self._ALG, self._N, self._d, device=device, store_dtype=store,
                         keep_exact_rows=keep_exact_rows)
        self._eng = eng
        if not self._check_monotone:
            eng.set_check_monotone(False)      # snnls.py:9,45,56: no error comparison / revert in the device state machine
        if self._N:
            if kind == "torch":
                if rows.device.type != "cuda":
                    eng.load_host_rows(rows.numpy(), center=self._center_rows)
                else:
                    eng.load_device_rows(rows.data_ptr(), self._N, rows.stride(0),
                                         rows.element_size() == 8, center=self._center_rows)
            else:
                eng.load_host_rows(rows, center=self._center_rows)
        bb = None if b is None else np.asarray(b, dtype=np.float64)
        rc = eng.finalize(bb)
        if rc == nat.ERR_ZERO_ROW:
            raise ValueError(self.alg_name + ".__init__(): A must not have any 0 columns")   # giga.py:11-12
        if rc == nat.ERR_ZERO_B:
            raise NumericalPrecisionError("norm of b must be > 0")                            # giga.py:16-17
        if rc != nat.OK:
            raise nat.EngineError(rc, eng.lib.bcx_last_error(eng.h).decode())
        self.reached_numeric_limit = False
        self.last_trace = None

    @property
    def A(self):
        """The d x N matrix the solver works on.  With ``center_rows`` (HilbertCoreset behind a device projector: the engine
        centred the rows while it ingested them and keeps no d x N copy for the caller) the centred matrix is FORMED ON FIRST
        USE -- a second N x d fp64 array next to the engine's, on the argument's device; ``An`` goes through it too.  Nothing
        in the build path touches it: read ``Anorms`` / ``weights()`` / ``error()`` instead where that is enough."""
        if not self._center_rows:
            return self._A_arg
        if self._A_centred is None:
            A = self._A_arg
            if hasattr(A, "dim") and A.is_cuda:
                # the N x S rows on the device: a device copy, then the library's centring pass in place (csrc/proj.hip)
                self._A_centred = nat.device_centred_copy(A.t()).t()
            elif hasattr(A, "dim"):
                An = A.detach().numpy()
                self._A_centred = A.new_tensor(An - An.mean(axis=0, keepdims=True))       # (a host tensor: NumPy)
            else:
                self._A_centred = A - A.mean(axis=0, keepdims=True)
        return self._A_centred

    @property
    def check_error_monotone(self):
        return self._check_monotone

    @check_error_monotone.setter
    def check_error_monotone(self, on):
        """The reference's greedy constructors take no such argument; callers switch the check off by assigning the
        attribute (snnls.py:16) -- forwarded to the device state machine."""
        self._check_monotone = bool(on)
        if self._eng is not None:
            self._eng.set_check_monotone(self._check_monotone)

    @property
    def b(self):
        """The target vector (as passed, or the device column sums when constructed with b=None)."""
        if self._b_arg is None and self._eng is not None:
            self._b_arg = self._eng.vector(0)
        return self._b_arg

    # ---- state ------------------------------------------------------------
    @property
    def w(self):
        """Dense length-N weight vector (materialised lazily from the device's sparse list)."""
        if self._w_cache is None:
            idx, wv = self._eng.sparse_weights()
            w = np.zeros(self._N)
            w[idx] = wv
            self._w_cache = w
        return self._w_cache

    def reset(self):
        self._eng.reset()
        self._w_cache = None
        self.reached_numeric_limit = False

    def size(self):
        idx, wv = self._eng.sparse_weights()
        return int((wv > 0).sum())

    def support(self):
        """(indices ascending, weights) of the columns with a positive weight -- what ``weights()`` holds, without the dense
        length-N vector (HilbertCoreset reads the solver through this: k values instead of N after every build call)."""
        idx, wv = self._eng.sparse_weights()
        keep = wv > 0
        idx, wv = idx[keep], wv[keep]
        order = np.argsort(idx, kind="stable")
        return idx[order], wv[order]

    def weights(self):
        return self.w.copy()

    def error(self):
        return self._eng.error()

    # ---- the hot loop ------------------------------------------------------
    def build(self, itrs):
        if self.reached_numeric_limit:
            self.log.warning("the numeric limit was already reached; returning. size = " + str(self.size())
                             + ", error = " + str(self.error()))
            return
        if self._N == 0 or self._d == 0:
            self.log.warning("there are no data, returning.")
            return
        self._eng.use_current_stream()
        tr = self._eng.run_build(int(itrs), float(util.TOL))
        self._w_cache = None
        if tr is None:
            return
        self.last_trace = tr
        sel, err, status = tr
        latched = self._eng.reached_numeric_limit()
        warn_failed_steps(self.log, status, latched)
        if latched:
            self.reached_numeric_limit = True
            self.log.warning("the numeric limit has been reached. No more points will be added. size = "
                             + str(self.size()) + ", error = " + str(self.error()))

    def optimize(self):
        """Re-solve the weights on the current support (snnls.py:82-97)."""
        accepted = self._eng.optimize(float(util.TOL))
        self._w_cache = None
        if not accepted:
            self.log.warning("self.optimize() returned a solution with increasing error. "
                             "Numeric limit possibly reached.")
            self.reached_numeric_limit = True

    # reference subclasses expose these after construction
    @property
    def Anorms(self):
        return self._eng.norms()

    @property
    def An(self):
        kind, rows = _as_row_matrix(self.A)
        rows = rows.cpu().numpy() if kind == "torch" else rows
        return (rows / self.Anorms[:, None]).T
