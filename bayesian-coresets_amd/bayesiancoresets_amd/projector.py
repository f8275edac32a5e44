"""Projector interface (reference: bayesiancoresets/projector.py:4-32), kept verbatim in
behaviour: ``project(pts, grad=False)`` returns an N x S array of log-likelihood vectors,
``update(wts, pts)`` refreshes the Monte-Carlo samples.  A projector may also return a
``torch.Tensor`` that already lives on the GPU; the solvers ingest it without a host copy."""
import numpy as np


class Projector(object):
    def project(self, pts, grad=False):
        raise NotImplementedError

    def update(self, wts, pts):
        raise NotImplementedError


class BlackBoxProjector(Projector):
    """Monte-Carlo projection: rows are ``loglikelihood(pts, samples)`` minus their row mean
    (projector.py:19-21; note: no 1/sqrt(S) scaling)."""

    def __init__(self, sampler, projection_dimension, loglikelihood, grad_loglikelihood=None):
        self.projection_dimension = projection_dimension
        self.sampler = sampler
        self.loglikelihood = loglikelihood
        self.grad_loglikelihood = grad_loglikelihood
        # samplers must accept empty wts/pts (projector.py:17)
        self.update(np.array([]), np.array([]))

    def update(self, wts, pts):
        self.samples = self.sampler(self.projection_dimension, wts, pts)

    def project(self, pts, grad=False):
        lls = self.loglikelihood(pts, self.samples)
        lls -= lls.mean(axis=1)[:, np.newaxis]
        if not grad:
            return lls
        if self.grad_loglikelihood is None:
            raise ValueError("grad_loglikelihood was requested but not initialized in BlackBoxProjector.project")
        glls = self.grad_loglikelihood(pts, self.samples)
        glls -= glls.mean(axis=2)[:, :, np.newaxis]
        return lls, glls


class DeviceProjector(Projector):
    """Monte-Carlo projection evaluated on the GPU for the reference's example likelihoods.

    Same contract as ``BlackBoxProjector`` (projector.py:11-32): ``update(wts, pts)`` draws
    ``samples = sampler(S, wts, pts)`` on the host (an S x D array), ``project(pts)`` returns the
    N x S matrix of log-likelihoods minus their row means -- here as a ``torch`` tensor resident on the
    GPU (fp64), produced by one fused kernel (Z Theta^T on the fp64 matrix cores + likelihood epilogue,
    csrc/proj.hip).  ``HilbertCoreset`` ingests it without a host copy; ``SparseVICoreset`` uses the
    two fused consumers ``project_colsum`` / ``project_select`` that never materialise N x S.

    family: "logistic"  rows z = y*x,            model_lr.py:25-32
            "poisson"   rows z = [x, y],         model_poiss.py:25-38
            "linreg"    rows z = [x, y], sigsq,  model_linreg.py:4-10
    """
    FAMILIES = {"logistic": 0, "poisson": 1, "linreg": 2}

    def __init__(self, family, sampler, projection_dimension, sigsq=1.0, device=0, group=None, row_offset=0, colsum="auto",
                 loglikelihood=None, grad_loglikelihood=None):
        """``group`` / ``row_offset``: row-sharded use (one process per GPU, every rank constructs the
        projector with the same sampler and seeds): ``pts`` passed to the fused consumers are this
        rank's rows starting at global row ``row_offset``; column sums are all-reduced and the arg-max
        is taken over all ranks (lowest global row wins ties).

        ``colsum`` (family "linreg" only; the others always project): how ``project_colsum`` of a large data set is formed --
        "mfma": one fused fp64-MFMA projection per call (2 N D S flops);
        "moments": in closed form from the (D+1) x (D+1) second moments of the data, formed once per data set
        (csrc/moments.hip; O(S D^2) per call -- SparseVI asks for 1 + opt_itrs column sums of the SAME data per step);
        "auto" (default): moments, after checking them ONCE per data set against the projection kernel (relative
        disagreement <= 1e-9 of the largest column sum; otherwise that data set stays on the projection kernel).

        ``loglikelihood`` / ``grad_loglikelihood`` (optional host callbacks with ``BlackBoxProjector``'s signatures):
        only for ``project(pts, grad=True)`` (projector.py:25-29), which is evaluated on the host at the current samples;
        everything N-sized stays on the device."""
        if colsum not in ("auto", "mfma", "moments"):
            raise ValueError("colsum must be 'auto', 'mfma' or 'moments'")
        self.colsum_mode = colsum
        self._mom, self._mom_ref, self._mom_key, self._mom_work, self._mom_ok = None, None, None, None, False
        self._mom_seen, self._mom_calls = None, 0
        self.moments_info = {}
        if family not in self.FAMILIES:
            raise ValueError("family must be one of %s" % sorted(self.FAMILIES))
        from . import _native
        import torch
        self._nat, self._torch = _native, torch
        self._lib = _native.load()
        if not torch.cuda.is_available():
            raise RuntimeError("DeviceProjector needs a GPU (there is no CPU fallback)")
        self.family, self._fam = family, self.FAMILIES[family]
        self.sampler = sampler
        self.projection_dimension = projection_dimension
        self.sigsq = float(sigsq)
        self.device = torch.device("cuda", device)
        self.group, self.row_offset = group, int(row_offset)
        self._world = 1
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self._world = torch.distributed.get_world_size(group)
        self._cache_val, self._cache_ref = None, None
        self._work = None
        self.loglikelihood, self.grad_loglikelihood = loglikelihood, grad_loglikelihood
        self.update(np.array([]), np.array([]))

    # -- plumbing -----------------------------------------------------------
    def _stream(self):
        return int(self._torch.cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc):
        if rc != 0:
            raise self._nat.EngineError(rc, self._lib.bcx_project_last_error().decode())

    # -- measurement: hipEvents around the projection kernel alone (csrc/proj.hip, bcx_project_profile) ------
    def profile(self, on):
        self._check(self._lib.bcx_project_profile(int(bool(on))))

    def profile_read(self):
        """(kernel milliseconds, launches, GEMM flops 2 N D S) since profile(True)."""
        import ctypes
        ms, n, fl = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double()
        self._check(self._lib.bcx_project_profile_read(ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)))
        return ms.value, n.value, fl.value

    def _launch(self, fn, args, Z):
        self._check(fn(*args))

    def _dev(self, pts):
        """Device copy of a host array.  The copy of the LAST large array is kept and reused only when the very same
        ndarray object comes back (SparseVI projects ``self.data`` 1 + opt_itrs times per step) -- the projector
        holds a reference to it, so the identity cannot be recycled.  A caller that edits that array in place must
        call ``invalidate_cache()``; device tensors pass through uncached."""
        torch = self._torch
        if isinstance(pts, torch.Tensor):
            t = pts.to(self.device, dtype=torch.float64)
            if t.dim() == 1:
                t = t[None, :]
            return t if t.stride(1) == 1 else t.contiguous()       # any row stride is fine, columns must be adjacent
        arr = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        big = arr.shape[0] >= 4096
        if big and self._cache_ref is pts and self._cache_val.shape == arr.shape:
            return self._cache_val
        if arr.shape[1] % 2:
            # odd row length: pad the device copy's leading dimension to even so rows stay 16-byte aligned (the kernel then
            # reads operands with 16-byte loads instead of 8-byte ones); the view has the caller's shape
            buf = torch.empty((arr.shape[0], arr.shape[1] + 1), dtype=torch.float64, device=self.device)
            buf[:, :arr.shape[1]] = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
            buf[:, arr.shape[1]] = 0.0
            t = buf[:, :arr.shape[1]]
        else:
            t = torch.from_numpy(np.ascontiguousarray(arr)).to(self.device)
        if big:
            self._cache_val, self._cache_ref = t, pts
        return t

    def invalidate_cache(self):
        self._cache_val, self._cache_ref = None, None
        self._mom, self._mom_ref, self._mom_key, self._mom_ok = None, None, None, False
        self._mom_seen, self._mom_calls = None, 0

    def release_scratch(self):
        """Give back the device scratch the fused consumers keep between calls (select: 32 bytes per row and 64-column
        group; column sums: 2048 x S doubles; the read-back buffer) -- e.g. between experiments on data sets of different
        sizes.  The next call allocates what it needs again."""
        self._work, self._sel_work, self._cc_buf, self._mom_work = None, None, None, None

    # -- second moments of a data set (linear-regression family): csrc/moments.hip -----------------------------------
    MOMENTS_RECHECK_EVERY = 64      # "auto": closed-form column sums are re-validated against the projection this often

    def _moments_for(self, pts, Z, persistent=True):
        """The (D+1) x (D+1) matrix Z^T Z of the data set ``pts`` (device copy ``Z``), kept while the SAME object keeps
        coming back (``invalidate_cache()`` after an in-place edit).  None when this call does not take the closed form.

        Every decision in here is the same on every rank of a row-sharded projector (the ranks call in lock step and
        each branch below has its own collectives): it depends on the family, the column count, ``persistent``, the
        GLOBAL row count and the call sequence -- never on the size of the local shard.
          * "auto": formed on the SECOND sight of the same object: a transient array (a fresh ``data[sub]`` per ADAM step)
            never pays for moments it cannot reuse;
          * row-sharded: only for the caller's persistent data (``persistent``: SparseVI passes ``pts is self.data``);
            per-step sub-samples always project;
          * "auto": checked against the projection kernel when formed and again every MOMENTS_RECHECK_EVERY-th use --
            theta moves with the weights, and the closed form (yy - 2 theta.X'y + theta'X'X theta) cancels as the
            residuals shrink; a disagreement > 1e-9 of the largest column sum retires it for this data set."""
        torch = self._torch
        if self.colsum_mode == "mfma" or self._fam != self.FAMILIES["linreg"] or Z.shape[1] > 1024:
            return None
        if self._world > 1 and not persistent:
            return None
        key = (Z.data_ptr(), tuple(Z.shape), Z.stride(0))
        if self._mom_ref is pts and self._mom_key == key:
            if not self._mom_ok:
                return None
            self._mom_calls += 1
            if self.colsum_mode == "auto" and self._mom_calls % self.MOMENTS_RECHECK_EVERY == 0:
                self._moments_check(Z, "rechecks")
            return self._mom if self._mom_ok else None
        if self.colsum_mode == "auto" and self._mom_seen is not pts:
            self._mom_seen = pts          # first sight: this call projects ("moments" was asked for by name: formed at once)
            return None
        import time
        t0 = time.perf_counter()
        n_rows = float(Z.shape[0])
        if self._world > 1:
            cnt = torch.tensor([n_rows], dtype=torch.float64, device=self.device)
            torch.distributed.all_reduce(cnt, op=torch.distributed.ReduceOp.SUM, group=self.group)
            n_rows = float(cnt.item())
        self._mom_ref, self._mom_key, self._mom_ok, self._mom_calls = pts, key, False, 0
        if n_rows < 4096:
            self._mom = None
            return None
        C = Z.shape[1]
        M = torch.zeros((C, C), dtype=torch.float64, device=self.device)
        if Z.shape[0]:
            need = int(self._lib.bcx_project_moments_scratch_bytes(int(Z.shape[0]), int(C)))
            work = torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device)
            self._check(self._lib.bcx_project_moments(self._stream(), Z.data_ptr(), Z.shape[0], Z.stride(0), C, M.data_ptr(), C,
                                                      work.data_ptr(), work.numel() * 8))
        if self._world > 1:
            torch.distributed.all_reduce(M, op=torch.distributed.ReduceOp.SUM, group=self.group)
        torch.cuda.synchronize(self.device)
        t1 = time.perf_counter()
        self._mom, self._mom_ok = M, True
        self.moments_info = {"rows": int(n_rows), "columns": int(C), "setup_ms": (t1 - t0) * 1e3, "checked": False}
        if self.colsum_mode == "auto":
            self._moments_check(Z, "checks")
        return self._mom if self._mom_ok else None

    def _moments_check(self, Z, counter):
        """One projection of this data set at the current samples: the closed form has to reproduce it."""
        import time
        t1 = time.perf_counter()
        a = self._colsum_projected(Z)
        b = self._colsum_from_moments(Z)
        scale = float(np.max(np.abs(a))) if a.size else 0.0
        dis = float(np.max(np.abs(a - b))) / scale if scale > 0 else 0.0
        self._mom_ok = bool(np.isfinite(dis) and dis <= 1e-9)
        self.moments_info.update({"checked": True, "disagreement": dis, "accepted": self._mom_ok,
                                  "check_ms": (time.perf_counter() - t1) * 1e3,
                                  counter: int(self.moments_info.get(counter, 0)) + 1})

    def _colsum_from_moments(self, Z, out=None, tbar=None):
        """``tbar`` (optional device vector, D doubles): the point the quadratic is expanded around -- the mean of the
        draws when their producer already has it (csrc/svi.hip); otherwise the library forms it."""
        torch = self._torch
        S, D = self.theta.shape[0], Z.shape[1] - 1
        if self.theta.shape[1] != D:
            raise ValueError("sampler returned %d-dimensional parameters for %d features" % (self.theta.shape[1], D))
        need = int(self._lib.bcx_project_colsum_moments_scratch_bytes(int(D), int(S))) // 8
        if self._mom_work is None or self._mom_work.numel() != need:
            self._mom_work = torch.zeros(need, dtype=torch.float64, device=self.device)     # (zero: the arrival counter)
        col = torch.empty(S, dtype=torch.float64, device=self.device) if out is None else out
        self._check(self._lib.bcx_project_colsum_moments_at(self._stream(), self._mom.data_ptr(), self._mom.stride(0), D, D,
                                                            self.theta.data_ptr(), S, self.theta.stride(0), self.sigsq,
                                                            col.data_ptr(), self._mom_work.data_ptr(),
                                                            None if tbar is None else tbar.data_ptr()))
        return col.cpu().numpy() if out is None else None

    def _dims(self, Z):
        cols = Z.shape[1]
        if self._fam == 0:
            return cols, -1
        return cols - 1, cols - 1

    def _common(self, Z):
        D, ycol = self._dims(Z)
        if self.theta.shape[1] != D:
            raise ValueError("sampler returned %d-dimensional parameters for %d features" % (self.theta.shape[1], D))
        return [self._stream(), self._fam, Z.data_ptr(), Z.shape[0], Z.stride(0), D, ycol, self.theta.data_ptr(),
                self.theta.shape[0], self.theta.stride(0), self.sigsq]

    # -- Projector interface ---------------------------------------------------
    def update(self, wts, pts):
        """``samples = sampler(S, wts, pts)`` (projector.py:23-24).  A sampler may return a torch tensor that is
        already on the GPU (S x D): it is then used in place -- no host round trip on the per-ADAM-step path of
        SparseVI, where a host sampler otherwise dominates (examples/common/model_linreg.py)."""
        self.use_draws(self.sampler(self.projection_dimension, wts, pts))

    def use_draws(self, drawn, mean=None):
        """Install ``drawn`` (S x D: ndarray or device tensor) as the current samples -- what ``update`` does with the
        sampler's return value.  A producer that enqueues its draws on the device (``SparseVICoreset``'s device-resident
        weight optimisation) calls this directly; ``mean`` is then the mean of the draws as a device vector, if it has it."""
        torch = self._torch
        self.theta_mean = mean
        if isinstance(drawn, torch.Tensor):
            t = drawn.to(self.device, dtype=torch.float64)
            if t.dim() == 1:
                t = t[None, :]
            self.samples = t
        else:
            self.samples = np.atleast_2d(np.asarray(drawn, dtype=np.float64))
            t = torch.from_numpy(np.ascontiguousarray(self.samples)).to(self.device)
        # the kernel reads 16-byte pieces of a parameter row: keep the rows 16-byte aligned (even leading dimension); a
        # sampler that already hands over such rows (a view of a padded buffer) is used in place
        if not (t.stride(1) == 1 and t.stride(0) % 2 == 0 and t.stride(0) >= t.shape[1] and t.data_ptr() % 16 == 0):
            buf = torch.zeros((t.shape[0], t.shape[1] + (t.shape[1] % 2)), dtype=torch.float64, device=self.device)
            buf[:, :t.shape[1]] = t
            t = buf[:, :t.shape[1]]
        self.theta = t

    def project(self, pts, grad=False):
        if grad:
            # (projector.py:19-29 on the host, at the samples the device holds)
            if self.loglikelihood is None or self.grad_loglikelihood is None:
                raise NotImplementedError("gradient projections are not on the device path: construct the projector with "
                                          "loglikelihood= and grad_loglikelihood= host callbacks")
            samples = self.theta.cpu().numpy()
            pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
            lls = self.loglikelihood(pts, samples)
            lls -= lls.mean(axis=1)[:, np.newaxis]
            glls = self.grad_loglikelihood(pts, samples)
            glls -= glls.mean(axis=2)[:, :, np.newaxis]
            return lls, glls
        torch = self._torch
        Z = self._dev(pts)
        N, S = Z.shape[0], self.theta.shape[0]
        out = torch.empty((N, S), dtype=torch.float64, device=self.device)
        if N:
            self._launch(self._lib.bcx_project_write, self._common(Z) + [out.data_ptr(), S, None], Z)
        return out

    def project_uncentred(self, pts):
        """The raw log-likelihoods ``loglikelihood(pts, samples)`` (N x S device tensor) WITHOUT the row-mean
        subtraction of projector.py:21 -- for a consumer that centres while it reads: ``HilbertCoreset`` hands them to
        the solver's constructor pass with ``center_rows=True`` (csrc/ingest.hip: the row is in registers between the
        norm and the stores), so the N x S matrix is written once and read once and the separate centring pass
        (another read + write of N x S) disappears.  ``project()`` is this followed by that pass."""
        torch = self._torch
        Z = self._dev(pts)
        N, S = Z.shape[0], self.theta.shape[0]
        out = torch.empty((N, S), dtype=torch.float64, device=self.device)
        if N:
            self._launch(self._lib.bcx_project_write_raw, self._common(Z) + [out.data_ptr(), S], Z)
        return out

    # -- fused consumers (SparseVI) -------------------------------------------------
    def _workspace(self, S):
        torch = self._torch
        if self._work is None or self._work.numel() < 2048 * max(S, 2):
            self._work = torch.empty(2048 * max(S, 2), dtype=torch.float64, device=self.device)
        return self._work

    def _select_scratch(self, N, S):
        """Caller-owned scratch of the select step (arg-max reduction + 32 bytes per row and 64-column group of partial row
        moments): kept between calls, so the step never goes to the stream-ordered allocator (whose pool would compete
        with torch's caching allocator: ~0.6 GB per call at N = 5M, S = 256)."""
        need = int(self._lib.bcx_project_select_scratch_bytes(self._fam, int(N), int(S)))
        if getattr(self, "_sel_work", None) is None or self._sel_work.numel() * 8 < need:
            self._sel_work = self._torch.empty((need + 7) // 8, dtype=self._torch.float64, device=self.device)
        return self._sel_work

    def project_colsum(self, pts, persistent=True):
        """sum_n vecs[n, :] as a length-S ndarray, without forming vecs.  ``persistent``: see ``_moments_for``."""
        Z = self._dev(pts)
        if self._moments_for(pts, Z, persistent) is not None:
            return self._colsum_from_moments(Z)        # (row-sharded: the moments are already the global ones)
        return self._colsum_projected(Z)

    def _colsum_projected(self, Z, out=None):
        torch = self._torch
        S = self.theta.shape[0]
        col = torch.empty(S, dtype=torch.float64, device=self.device) if out is None else out
        if Z.shape[0]:
            self._launch(self._lib.bcx_project_colsum, self._common(Z) + [col.data_ptr(), self._workspace(S).data_ptr()], Z)
        else:
            col.zero_()
        if self._world > 1:
            # every rank applied the centring correction with ITS raw sums; the correction is linear, so
            # the all-reduced vector is the centred global column sum
            torch.distributed.all_reduce(col, op=torch.distributed.ReduceOp.SUM, group=self.group)
        return col.cpu().numpy() if out is None else None

    def colsum_and_core(self, pts, core, persistent=True):
        """(project_colsum(pts), project(core) as an ndarray) with ONE device->host copy: what every ADAM step of SparseVI
        reads back (sparsevi.py:35-41, 70-74).  ``core`` is the k x (D+1) array of coreset points (ndarray or device tensor),
        k may be 0.  ``persistent``: ``pts`` is the caller's standing data set, not a per-call sub-sample (``_moments_for``)."""
        buf, k = self.colsum_and_core_enqueue(pts, core, persistent)
        S = self.theta.shape[0]
        h = buf.cpu().numpy()
        return h[:S], h[S:].reshape(k, S)

    def colsum_and_core_enqueue(self, pts, core, persistent=True):
        """The same two projections left ON THE DEVICE, nothing read back: (buf, k) with buf[:S] the column sums and
        buf[S:] the k x S projected coreset points (one buffer, reused by the next call; valid in stream order)."""
        torch = self._torch
        Z = self._dev(pts)
        S = self.theta.shape[0]
        C = None if core is None or core.shape[0] == 0 else self._dev(core)
        k = 0 if C is None else C.shape[0]
        if getattr(self, "_cc_buf", None) is None or self._cc_buf.numel() < S * (k + 1):
            self._cc_buf = torch.empty(S * (max(k, 7) + 1), dtype=torch.float64, device=self.device)
        buf = self._cc_buf[:S * (k + 1)]
        col = buf[:S]
        if self._moments_for(pts, Z, persistent) is not None:
            self._colsum_from_moments(Z, out=col, tbar=getattr(self, "theta_mean", None))
        else:
            self._colsum_projected(Z, out=col)      # (a shard without rows: zeros, and still the all-reduce its peers join)
        if k:
            # (the coreset points: every shard projects them alike -- the kernel may be chosen by their number)
            self._launch(self._lib.bcx_project_write_points, self._common(C) + [buf[S:].data_ptr(), S, 1], C)
        return buf, k

    def enqueue_step_plan(self, pts, core, persistent, draws, mean):
        """For a loop that repeats ``colsum_and_core_enqueue(pts, core)`` at draws that are rewritten IN PLACE between the
        repetitions (``SparseVICoreset``'s device-resident weight optimisation): installs ``draws`` / ``mean`` once and returns
        ``(run, buf, k)`` -- ``run()`` enqueues the column sums of ``pts`` into buf[:S] and the k x S projected coreset
        points into buf[S:], the latter as RAW log-likelihoods (uncentred: ``bcx_sparsevi_adam_step`` takes the row means),
        from argument lists built once.  The decisions of ``_moments_for`` are taken at every repetition as before."""
        self.use_draws(draws, mean=mean)
        if self.theta.data_ptr() != draws.data_ptr():
            # (use_draws copies rows that do not start on 16-byte boundaries: the copy would go stale at the next repetition)
            raise ValueError("enqueue_step_plan: the draws must be usable in place (device tensor, fp64, unit column stride, "
                             "even row stride, 16-byte aligned)")
        torch, lib = self._torch, self._lib
        Z, C = self._dev(pts), self._dev(core)
        S, k = self.theta.shape[0], C.shape[0]
        if getattr(self, "_cc_buf", None) is None or self._cc_buf.numel() < S * (k + 1):
            self._cc_buf = torch.empty(S * (max(k, 7) + 1), dtype=torch.float64, device=self.device)
        buf = self._cc_buf[:S * (k + 1)]
        col = buf[:S]
        core_args = self._common(C) + [buf[S:].data_ptr(), S, 0]
        state = {"mom": None, "both": None}

        def run():
            if self._moments_for(pts, Z, persistent) is not None:
                a = state["mom"]
                if a is None or a[1] != self._mom.data_ptr():
                    D = Z.shape[1] - 1
                    need = int(lib.bcx_project_colsum_moments_scratch_bytes(int(D), int(S))) // 8
                    if self._mom_work is None or self._mom_work.numel() != need:
                        self._mom_work = torch.zeros(need, dtype=torch.float64, device=self.device)
                    a = state["mom"] = [core_args[0], self._mom.data_ptr(), self._mom.stride(0), D, D, self.theta.data_ptr(), S,
                                        self.theta.stride(0), self.sigsq, col.data_ptr(), self._mom_work.data_ptr(),
                                        None if mean is None else mean.data_ptr()]
                    # both projections only read the draws: one launch (csrc/proj.hip proj_mid_quad_kernel) where they fit it
                    c = core_args          # stream, family, Z, N, ldz, D, ycol, theta, S, ldt, param, out, ldo, center
                    state["both"] = None if mean is None else [c[0], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11],
                                                               c[12], a[1], a[2], a[4], a[9], a[10], a[11]]
                if state["both"] is not None:
                    self._check(lib.bcx_project_points_colsum_moments(*state["both"]))
                    return
                self._check(lib.bcx_project_colsum_moments_at(*a))
            else:
                self._colsum_projected(Z, out=col)
            self._check(lib.bcx_project_write_points(*core_args))
        return run, buf, k

    def project_select(self, pts, resid, row_ids=None):
        """(max_n corr_n, arg-max row) with corr_n = vecs[n].resid / ||vecs[n]|| / S (first maximum).
        ``row_ids`` (ascending, one per local row): the identity under which a local row competes -- its global row
        number (default: ``row_offset`` + local row) or, for a random subsample, its position in the drawn index
        array -- so that the first-maximum rule of ``corrs.argmax()`` (sparsevi.py:55) holds across shards."""
        torch = self._torch
        Z = self._dev(pts)
        S = self.theta.shape[0]
        r = torch.from_numpy(np.ascontiguousarray(resid, dtype=np.float64)).to(self.device)
        res = torch.empty(2, dtype=torch.float64, device=self.device)
        if Z.shape[0]:
            work = self._select_scratch(Z.shape[0], S)
            self._launch(self._lib.bcx_project_select_ws, self._common(Z) + [r.data_ptr(), float(np.sum(resid)), res.data_ptr(),
                                                                     work.data_ptr(), work.numel() * 8], Z)
            h = res.cpu()
            best, row = float(h[0]), int(h[1:2].view(torch.int64)[0])
            row = int(row_ids[row]) if row_ids is not None else row + self.row_offset
        else:
            best, row = -np.inf, -1
        if self._world > 1:
            mine = torch.tensor([best, float(row)], dtype=torch.float64, device=self.device)
            allr = torch.empty(2 * self._world, dtype=torch.float64, device=self.device)
            try:
                torch.distributed.all_gather_into_tensor(allr, mine, group=self.group)
            except (RuntimeError, NotImplementedError):
                torch.distributed.all_gather(list(allr.view(self._world, 2).unbind(0)), mine, group=self.group)
            recs = allr.cpu().numpy().reshape(self._world, 2)
            best, row = -np.inf, -1
            for v, i in recs:
                # NaN (a zero vector's 0/0) is NumPy's maximum: the first NaN position wins outright
                if i >= 0 and (row < 0 or (np.isnan(v) and not np.isnan(best)) or (np.isnan(v) and i < row)
                               or (not np.isnan(best) and (v > best or (v == best and i < row)))):
                    best, row = float(v), int(i)
        return best, row
