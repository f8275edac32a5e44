"""CPU oracle for the greedy sparse-NNLS hot path (GIGA / Frank-Wolfe / OMP).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the shipped
engine: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this file, and only as the checker / the timed
CPU baseline.  The product path (``bayesian-coresets_amd/``) never imports it
and raises when the HIP library is missing.

This is a NumPy restatement of the reference algorithm, written from
SURVEY.md section 8a, organised as an explicit state machine (status codes per
greedy iteration) instead of the reference's exception-driven loop.  Each
function cites the reference file:line it follows (paths relative to the
upstream repository ``trevorcampbell/bayesian-coresets`` @ v0.9.1):

* driver / monotone check / retry / latch ...... bayesiancoresets/snnls/snnls.py:31-79
* error(), size(), weights(), reset() .......... bayesiancoresets/snnls/snnls.py:18-29
* optimize() ................................... bayesiancoresets/snnls/snnls.py:82-97
* GIGA  ctor / select / reweight ............... bayesiancoresets/snnls/giga.py:8-18 / 20-38 / 40-64
* FW    ctor / select / reweight ............... bayesiancoresets/snnls/frankwolfe.py:7-13 / 15-17 / 19-40
* OMP   ctor / select / reweight ............... bayesiancoresets/snnls/orthopursuit.py:9-15 / 17-35 / 37-42

Third-party arithmetic on this path that is NOT in the reference tree:
``scipy.optimize.nnls`` (SciPy, un-pinned in the reference's setup.py:11; this
image ships 1.15.3, Lawson-Hanson / Bro-de Jong active set).  The oracle calls
it exactly where the reference does (orthopursuit.py:40, snnls.py:87).

Pinning: the reference has no tests or golden vectors of its own, so the oracle
is pinned against outputs of the reference itself, imported in the build
container by ``tests/golden/make_golden.py`` (committed together with the
vectors it produced).  ``tests/test_oracle_golden.py`` checks every fixture.

Two execution modes share the same arithmetic for select/reweight:

``faithful``  recomputes ``A.dot(w)`` wherever the reference does (5 passes
              over the N x d matrix per greedy iteration).  Bit-for-bit the
              reference's operation sequence; this is the timed CPU baseline.
``onepass``   maintains ``xw = A w`` incrementally (1 pass per iteration), the
              algorithm the device engine implements.  Same selections; weights
              agree to ~1e-13 relative.
"""
import numpy as np
from scipy.optimize import nnls as _scipy_nnls

# per-iteration status codes (shared vocabulary with include/bcx.h)
ST_OK = 0            # step accepted
ST_FAIL_SELECT = 1   # numerical precision failure inside select  (giga.py:28-29)
ST_FAIL_REWEIGHT = 2 # numerical precision failure inside reweight (giga.py:50-51, frankwolfe.py:32-33)
ST_FAIL_MONOTONE = 3 # error increased; weights reverted (snnls.py:58-61)

ALGS = ("giga", "fw", "omp")


class _PrecisionFailure(Exception):
    def __init__(self, status):
        super().__init__(status)
        self.status = status


class SnnlsOracle:
    """State machine equivalent of reference ``SparseNNLS`` subclasses.

    ``A`` is d x N (columns are data points), ``b`` is length d, exactly as the
    reference constructors take them (snnls.py:9).  Pass ``vecs.T`` of an
    N x d C-contiguous array to get the reference's memory layout.
    """

    def __init__(self, A, b, alg="giga", tol=1e-12, mode="faithful", check_error_monotone=True):
        if alg not in ALGS:
            raise ValueError("alg must be one of %s" % (ALGS,))
        if mode not in ("faithful", "onepass"):
            raise ValueError("mode must be faithful|onepass")
        self.alg, self.mode, self.tol = alg, mode, tol
        self.check_error_monotone = check_error_monotone    # snnls.py:9,16
        self.A, self.b = A, b
        self.N = A.shape[1]
        self.w = np.zeros(self.N)
        self.reached_numeric_limit = False
        self.trace = []  # (f, error_after, status) per loop iteration
        # column norms + normalised copy: giga.py:10-13, frankwolfe.py:10-13, orthopursuit.py:12-15
        self.Anorms = np.sqrt((A ** 2).sum(axis=0))
        if np.any(self.Anorms == 0):
            raise ValueError("A must not have any 0 columns")
        self.An = A / self.Anorms
        if alg == "giga":
            # giga.py:15-18
            self.bnorm = np.sqrt((b ** 2).sum())
            if self.bnorm == 0.0:
                raise ArithmeticError("norm of b must be > 0")
            self.bn = b / self.bnorm
        self._xw = np.zeros(A.shape[0])  # onepass state

    # ---- snnls.py:18-29 -------------------------------------------------
    def reset(self):
        self.w = np.zeros(self.N)
        self._xw = np.zeros(self.A.shape[0])
        self.reached_numeric_limit = False

    def size(self):
        return int((self.w > 0).sum())

    def weights(self):
        return self.w.copy()

    def _Aw(self):
        if self.mode == "faithful":
            return self.A.dot(self.w)
        return self._xw.copy()

    def error(self):
        return float(np.sqrt(((self._Aw() - self.b) ** 2).sum()))

    # ---- select ---------------------------------------------------------
    def _select(self):
        if self.alg == "giga":
            return self._select_giga()
        residual = self.b - self._Aw()                      # frankwolfe.py:16 / orthopursuit.py:18
        dots = self.An.T.dot(residual)                      # frankwolfe.py:17 / orthopursuit.py:19
        if self.alg == "fw" or self.size() == 0:            # orthopursuit.py:22-23
            return int(dots.argmax())
        fpos = int(dots.argmax())                           # orthopursuit.py:26-35
        active = self.w > 0
        neg_dots = -dots[active]
        jneg = int(neg_dots.argmax())
        if dots[fpos] >= neg_dots[jneg]:
            return fpos
        return int(np.flatnonzero(active)[jneg])

    def _select_giga(self):
        xw = self._Aw()                                     # giga.py:21
        nw = np.sqrt((xw ** 2).sum())
        nw = 1.0 if nw == 0.0 else nw
        xw /= nw
        cdir = self.bn - self.bn.dot(xw) * xw               # giga.py:26
        cdirnrm = np.sqrt((cdir ** 2).sum())
        if cdirnrm < self.tol:                              # giga.py:28-29
            raise _PrecisionFailure(ST_FAIL_SELECT)
        cdir /= cdirnrm
        sc = self.An.T.dot(np.hstack((cdir[:, None], xw[:, None])))   # giga.py:31
        ok = np.logical_and(sc[:, 1] > -1.0 + 1e-14, 1.0 - sc[:, 1] ** 2 > 0.0)  # giga.py:33
        sc[ok, 1] = np.sqrt(1.0 - sc[ok, 1] ** 2)
        sc[np.logical_not(ok), 1] = np.inf
        return int((sc[:, 0] / sc[:, 1]).argmax())          # giga.py:38

    # ---- reweight -------------------------------------------------------
    def _apply_axpy(self, alpha, beta, f):
        """w <- alpha*w ; w[f] <- max(0, w[f]+beta)   (giga.py:63-64, frankwolfe.py:39-40)"""
        old_f = self.w[f]
        self.w = alpha * self.w
        self.w[f] = max(0.0, self.w[f] + beta)
        if self.mode == "onepass":
            # xw' = A w' = alpha*xw + (w'_f - alpha*w_f) * A[:,f]
            self._xw = alpha * self._xw + (self.w[f] - alpha * old_f) * self.A[:, f]

    def _reweight(self, f):
        if self.alg == "giga":
            xw = self._Aw()                                 # giga.py:42
            nw = np.sqrt((xw ** 2).sum())
            nw = 1.0 if nw == 0.0 else nw
            xf = self.A[:, f]
            nf = np.sqrt((xf ** 2).sum())
            gA = self.bn.dot(xf / nf) - self.bn.dot(xw / nw) * (xw / nw).dot(xf / nf)   # giga.py:48
            gB = self.bn.dot(xw / nw) - self.bn.dot(xf / nf) * (xw / nw).dot(xf / nf)   # giga.py:49
            if gA <= 0.0 or gB < 0:                         # giga.py:50-51
                raise _PrecisionFailure(ST_FAIL_REWEIGHT)
            a = gB / (gA + gB) / nw
            c = gA / (gA + gB) / nf
            x = a * xw + c * xf
            nx = np.sqrt((x ** 2).sum())
            scale = self.bnorm / nx * (x / nx).dot(self.bn)  # giga.py:58
            self._apply_axpy(a * scale, c * scale, f)
        elif self.alg == "fw":
            if self.size() == 0:                            # frankwolfe.py:20-23
                self._apply_axpy(0.0, self.Anorms.sum() / self.Anorms[f], f)
                return
            nsum = self.Anorms.sum()                        # frankwolfe.py:25-28
            nf = self.Anorms[f]
            xw = self._Aw()
            xf = self.A[:, f]
            gnum = (nsum / nf * xf - xw).dot(self.b - xw)   # frankwolfe.py:30
            gden = ((nsum / nf * xf - xw) ** 2).sum()       # frankwolfe.py:31
            if gnum < 0.0 or gden == 0.0 or gnum > gden:    # frankwolfe.py:33-34
                raise _PrecisionFailure(ST_FAIL_REWEIGHT)
            self._apply_axpy(1.0 - gnum / gden, nsum / nf * gnum / gden, f)
        else:
            self.w[f] = 1.0                                 # orthopursuit.py:38
            active = self.w > 0
            sol = _scipy_nnls(self.A[:, active], self.b, maxiter=100 * self.N)   # orthopursuit.py:40
            self.w[active] = sol[0]
            if self.mode == "onepass":
                self._xw = self.A[:, active].dot(self.w[active])

    # ---- driver: snnls.py:31-79 ----------------------------------------
    def build(self, itrs):
        """Run ``itrs`` loop iterations; returns the list of (f, err, status) appended."""
        start = len(self.trace)
        if self.reached_numeric_limit or self.A.size == 0:   # snnls.py:32-38
            return []
        retried = False
        for _ in range(itrs):
            f = -1
            checked = self.check_error_monotone and self.size() > 0   # snnls.py:44-45
            if checked:
                prev_err = self.error()                      # snnls.py:46-47
                prev_w, prev_xw = self.w.copy(), self._xw.copy()
            try:
                f = self._select()
                self._reweight(f)
                if checked:
                    err = self.error()                       # snnls.py:57
                    if err > prev_err:                       # snnls.py:58-61
                        self.w, self._xw = prev_w, prev_xw
                        raise _PrecisionFailure(ST_FAIL_MONOTONE)
                    retried = False                          # snnls.py:62
                self.trace.append((f, self.error(), ST_OK))
            except _PrecisionFailure as e:                   # snnls.py:63-72
                self.trace.append((f, self.error(), e.status))
                if retried:
                    self.reached_numeric_limit = True
                    break
                retried = True
        return self.trace[start:]

    # ---- snnls.py:82-97 -------------------------------------------------
    def optimize(self, tol=None):
        tol = self.tol if tol is None else tol
        prev_cost = self.error()
        prev_w, prev_xw = self.w.copy(), self._xw.copy()
        active = self.w > 0
        sol = _scipy_nnls(self.A[:, active], self.b, maxiter=100 * self.N)
        self.w[active] = sol[0]
        if self.mode == "onepass":
            self._xw = self.A[:, active].dot(self.w[active])
        if self.error() > prev_cost * (1.0 + tol):
            self.w, self._xw = prev_w, prev_xw
            self.reached_numeric_limit = True
            return False
        return True


def hilbert_readout(w, sub_idcs=None):
    """(wts, idcs) as HilbertCoreset._build produces them: hilbert.py:35-37 (index-sorted)."""
    keep = w > 0
    idcs = np.flatnonzero(keep) if sub_idcs is None else sub_idcs[keep]
    return w[keep], idcs


def synthetic_normal(trial, N, d):
    """Workload of examples/synthetic_vectors/main.py:44,63 (legacy RandomState stream)."""
    rs = np.random.RandomState(trial)
    return rs.randn(N, d)


def harness_sizes(size_max=1000, num=50):
    """Coreset-size schedule: examples/synthetic_vectors/main.py:51-52."""
    return np.unique(np.logspace(0.0, np.log10(size_max), num, dtype=np.int32))
