"""Device-resident sampler of the weighted conjugate posterior of Gaussian linear regression -- the ``sampler`` argument
of ``DeviceProjector("linreg", ...)`` for the reference's linear-regression experiment (examples/linear_regression/
main.py:124-147: prior theta ~ N(mu0, Sig0), noise variance sigsq, rows z = [x, y]):

    Sigma_w^-1 = Sig0^-1 + X^T diag(w) X / sigsq,      mu_w = Sigma_w (Sig0^-1 mu0 + X^T diag(w) y / sigsq)

SparseVI calls its sampler once per ADAM step (sparsevi.py:25 -> projector.update), 1 + opt_itrs times per greedy step,
with the SAME points and new weights.  The draws are formed on the device -- for a few points by one kernel (csrc/svi.hip: a
rank-k correction of the prior's factor through a k x k Cholesky inside one workgroup), for more (the reference's experiment
grows the coreset to 300 points at D = 301) by the reference's own arithmetic, theta = mu_w + R L^-1 with L L^T the D x D
precision (csrc/lrpost.hip: one cooperative launch factors it and forms L^-1 and mu_w) -- in two ways:

* ``sampler(n, wts, pts)``: the reference's sampler signature -- uploads the k weights, returns the draws as a device
  tensor (``DeviceProjector`` uses them in place);
* ``sampler.enqueue_plan(n, pts, steps)``: for ``SparseVICoreset``'s device-resident weight optimisation -- the plan's
  ``draw(w_dev, i)`` takes the weights FROM the device and enqueues the kernel: no host synchronisation per ADAM step.

Everything that depends on the points alone (X U0, X Sig0, their Gram matrix, X mu0) is formed on the host when the
points change (once per greedy step) and uploaded in one piece.  There is no CPU fallback."""
import numpy as np


class _DeviceNormals(object):
    """The samplers' raw material: normal numbers and their column means by the library's own kernels (csrc/svi.hip).  Needs
    ``_torch``, ``_lib``, ``_nat``, ``device``, ``ld``, ``_seed``, ``_offset``."""

    # -- noise: separate hooks so that a test can feed both entry points the same numbers --------------------------------
    # Standard normal doubles from the library's own counter-based generator (csrc/svi.hip svi_normal_kernel: Philox-4x32-10 +
    # Box-Muller, key = the sampler's seed): one reproducible stream, the pair counter advances by what each call consumed.
    def _normal(self, *shape):
        torch = self._torch
        out = torch.empty(*shape, dtype=torch.float64, device=self.device)
        count = out.numel()
        rc = self._lib.bcx_standard_normal(int(torch.cuda.current_stream(self.device).cuda_stream), self._seed, self._offset, count,
                                           out.data_ptr())
        if rc != 0:
            raise self._nat.EngineError(rc, self._lib.bcx_project_last_error().decode())
        self._offset += (count + 1) // 2
        return out

    def _noise(self, n):
        return self._normal(n, self.ld)

    def _noise_block(self, steps, n):
        return self._normal(steps, n, self.ld)

    def _column_means(self, blocks):
        """Column means of every n x ld block of ``blocks`` (steps x n x ld, or n x ld): steps x ld (or ld), one launch."""
        torch = self._torch
        three = blocks.dim() == 3
        nb, n = (blocks.shape[0], blocks.shape[1]) if three else (1, blocks.shape[0])
        if blocks.stride(-1) != 1 or blocks.stride(-2) != self.ld or (three and nb > 1 and blocks.stride(0) < n * self.ld):
            blocks = blocks.contiguous()
        out = torch.empty((nb, self.ld) if three else (self.ld,), dtype=torch.float64, device=self.device)
        rc = self._lib.bcx_column_means(int(torch.cuda.current_stream(self.device).cuda_stream), blocks.data_ptr(), nb, n, self.ld,
                                        blocks.stride(0) if three else 0, out.data_ptr(), self.ld)
        if rc != 0:
            raise self._nat.EngineError(rc, self._lib.bcx_project_last_error().decode())
        return out


class LinregPosteriorSampler(_DeviceNormals):
    KMAX = 4096    # weighted points (csrc/lrpost.hip)
    KLOW = 64      # ... that the rank-k form of csrc/svi.hip takes (LRS_KMAX)
    DMAX = 1024    # features the D x D form takes (csrc/lrpost.hip LP_NB * LP_MAX_NT)
    SMAX = 4096    # draws per call (DeviceProjector's own limit on the projection dimension)
    NOISE_BUDGET = 2 << 30      # bytes of pre-drawn normal numbers (+ their images) an enqueue plan may hold
    ROWS_MAX = 1 << 22          # ... and rows of them one launch of the draw kernel takes (csrc/svi.hip LRS_SMAX)

    def __init__(self, mu0, Sig0, sigsq, device="cuda", seed=None):
        import torch
        from . import _native
        self._torch, self._nat = torch, _native
        self._lib = _native.load()
        if not torch.cuda.is_available():
            raise RuntimeError("LinregPosteriorSampler needs a GPU (there is no CPU fallback)")
        self.device = torch.device(device)
        self.mu0 = np.ascontiguousarray(mu0, dtype=np.float64)
        self.Sig0 = np.ascontiguousarray(Sig0, dtype=np.float64)
        self.sigsq = float(sigsq)
        D = self.D = self.mu0.shape[0]
        self.ld = D + (D % 2)                               # rows of the draws start on 16-byte boundaries
        self.U0 = np.linalg.cholesky(self.Sig0)             # Sig0 = U0 U0^T (any such factor serves)
        U0T = np.zeros((D, self.ld))
        U0T[:, :D] = self.U0.T
        self._U0T = torch.from_numpy(U0T).to(self.device)
        # an isotropic / diagonal prior (the reference's experiment: Sig0 = c I): R U0^T is a column scaling
        self._U0_diag = None
        if np.count_nonzero(self.U0 - np.diag(np.diag(self.U0))) == 0:
            dg = np.zeros(self.ld)
            dg[:D] = np.diag(self.U0)
            self._U0_diag = torch.from_numpy(dg).to(self.device)
        self._mu0 = torch.from_numpy(self.mu0).to(self.device)
        self._seed, self._offset = (0 if seed is None else int(seed)) & 0xFFFFFFFFFFFFFFFF, 0
        self._pts_key, self._pts_state = None, None
        self._theta, self._tbar = {}, torch.empty(D, dtype=torch.float64, device=self.device)
        self._none = torch.zeros(1, dtype=torch.float64, device=self.device)
        self._zero_mu = torch.zeros(D, dtype=torch.float64, device=self.device)
        self._scratch_mean = torch.empty(D, dtype=torch.float64, device=self.device)
        # the D x D form (csrc/lrpost.hip): prior precision and Sig0^-1 mu0 once; scratch, L^-1 and mu_w on first use
        self._S0inv_host = np.linalg.inv(self.Sig0)
        self._factor = None

    # -- per-point state ------------------------------------------------------------------------------------------------------
    def supports(self, n, k):
        return 0 <= k <= (self.KMAX if self.D <= self.DMAX else self.KLOW) and 1 <= n <= self.SMAX

    def _low_rank(self, k):
        """Which form serves k points: the rank-k correction inside one workgroup (csrc/svi.hip lrs_apply_kernel: every
        workgroup repeats the k x k Cholesky, a thread per column does the triangular solves) while that is the faster one,
        the D x D factorisation (csrc/lrpost.hip) beyond.  Measured per ADAM step of the enqueued loop at D = 301, S = 256
        (tools/c5_ksweep.py): rank-k 40 / 54 / 72 / 94 / 119 / 147 / 171 us at k = 4 / 8 / 12 / 16 / 20 / 24 / 27, the D x D form
        143 - 157 us whatever k (its chain of D pivots does not depend on k: ~9 us per block of 32 columns + ~60 us)."""
        if self.D > self.DMAX:
            return True
        nt = (self.D + 31) // 32
        return k == 0 or (k <= 4 + 2 * nt and bool(self._lib.bcx_linreg_posterior_apply_ok(k, self.ld)))

    def _factor_state(self):
        f = self._factor
        if f is None:
            torch, D = self._torch, self.D
            need = int(self._lib.bcx_linreg_posterior_factor_scratch_bytes(D))
            f = self._factor = {
                "S0inv": torch.from_numpy(np.ascontiguousarray(self._S0inv_host)).to(self.device),
                "rhs0": torch.from_numpy(self._S0inv_host.dot(self.mu0)).to(self.device),
                "work": torch.empty((need + 7) // 8, dtype=torch.float64, device=self.device),
                "U": torch.zeros(D, self.ld, dtype=torch.float64, device=self.device),        # U = L^-T (the lower triangle stays zero)
                "u": torch.zeros(D, dtype=torch.float64, device=self.device),                   # L^-1 rhs: mu_w = U u
            }
        return f

    def _factor_args(self, st, w_dev):
        """Argument list of bcx_linreg_posterior_factor at the points ``st`` and the device-resident weights ``w_dev``."""
        f = self._factor_state()
        stream = int(self._torch.cuda.current_stream(self.device).cuda_stream)
        return [stream, st["k"], self.D, st["ldk"], w_dev.data_ptr(), st["XT"].data_ptr(), st["y"].data_ptr(), f["S0inv"].data_ptr(), self.D,
                f["rhs0"].data_ptr(), self.sigsq, f["work"].data_ptr(), f["work"].numel() * 8, f["U"].data_ptr(), self.ld,
                f["u"].data_ptr(), None]                    # (no mean: the draws are U (u + r))

    def _draw_factored_args(self, theta, tbar=None):
        """Argument list of bcx_linreg_posterior_draw_factored (theta = mu_w + R U^T); slots 6 / 7 take the normal numbers and
        their column means."""
        f = self._factor
        stream = int(self._torch.cuda.current_stream(self.device).cuda_stream)
        return [stream, self.D, self.ld, f["U"].data_ptr(), self.ld, f["u"].data_ptr(), 0, 0, theta.shape[0], theta.data_ptr(),
                (self._tbar if tbar is None else tbar).data_ptr()]

    def factor_status(self):
        """Synchronises; raises if the last factorisation's workgroups lost each other or met a non-positive pivot."""
        if self._factor is not None:
            stream = int(self._torch.cuda.current_stream(self.device).cuda_stream)
            rc = self._lib.bcx_linreg_posterior_factor_status(stream, self.D, self._factor["work"].data_ptr())
            if rc != 0:
                raise self._nat.EngineError(rc, self._lib.bcx_project_last_error().decode())

    def _points(self, pts):
        """Device block [K0 (k x k) | X mu0 (k) | y (k) | X U0 (k x ld) | X Sig0 (k x ld)] of the points ``pts`` (k x (D+1))."""
        torch = self._torch
        pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        k = pts.shape[0]
        if pts.shape[1] != self.D + 1:
            raise ValueError("points have %d columns, the model has %d features + the response" % (pts.shape[1], self.D))
        if self._pts_key is not None and self._pts_key.shape == pts.shape and np.array_equal(self._pts_key, pts):
            return self._pts_state
        X, y = pts[:, :-1], pts[:, -1]
        Xp = np.zeros((k, self.ld))
        Xp[:, :self.D] = X
        if not self._low_rank(k):
            # the D x D form needs the responses and the features BY points (row a: feature a of the k points; the row stride
            # is k rounded up to 32, zero padded: csrc/lrpost.hip reads both operands of X^T diag(w) X along the points)
            ldk = (k + 31) // 32 * 32
            XT = np.zeros((self.D, ldk))
            XT[:, :k] = X.T
            d = torch.from_numpy(np.concatenate((y, np.zeros(k % 2), XT.ravel()))).to(self.device)
            st = {"k": k, "y": d[:k], "XT": d[k + k % 2:], "ldk": ldk, "blob": d, "low_rank": False}
            self._pts_key, self._pts_state = pts.copy(), st
            return st
        XU0, XS0 = np.zeros((k, self.ld)), np.zeros((k, self.ld))
        XU0[:, :self.D] = X.dot(self.U0)
        XS0[:, :self.D] = X.dot(self.Sig0)
        K0 = XU0.dot(XU0.T)
        pad = (k * k + 2 * k) % 2                           # the k x ld blocks start on 16-byte boundaries
        blob = np.concatenate((K0.ravel(), X.dot(self.mu0), y, np.zeros(pad), XU0.ravel(), XS0.ravel(), Xp.ravel()))
        d = torch.from_numpy(blob).to(self.device)
        o = k * k
        st = {"k": k, "K0": d[:o], "xmu0": d[o:o + k], "y": d[o + k:o + 2 * k], "low_rank": True}
        o += 2 * k + pad
        n = k * self.ld
        st["XU0"], st["XS0"], st["X"] = d[o:o + n], d[o + n:o + 2 * n], d[o + 2 * n:o + 3 * n]
        st["blob"] = d
        self._pts_key, self._pts_state = pts.copy(), st
        return st

    def _launch(self, st, w_dev, R, rbar, theta, mu0=None, tbar=None, U0T=None):
        """lrs_draw_kernel: theta = mu + [R; rbar] Uw^T.  ``U0T`` / ``mu0`` override the prior's factor and mean (with st = None:
        theta = mu0 + R U0T -- the D x D form passes L^-1 and mu_w)."""
        k = st["k"] if st is not None else 0
        lib, dp = self._lib, (lambda key: st[key].data_ptr() if k else None)
        stream = int(self._torch.cuda.current_stream(self.device).cuda_stream)
        rc = lib.bcx_linreg_posterior_draw(stream, k, self.D, self.ld, w_dev.data_ptr() if k else None, dp("K0"), dp("xmu0"), dp("y"),
                                           dp("XU0"), dp("XS0"), (self._U0T if U0T is None else U0T).data_ptr(), (self._mu0 if mu0 is None else mu0).data_ptr(),
                                           self.sigsq, R.data_ptr(), rbar.data_ptr(), R.shape[0], theta.data_ptr(),
                                           (self._tbar if tbar is None else tbar).data_ptr())
        if rc != 0:
            raise self._nat.EngineError(rc, lib.bcx_project_last_error().decode())

    def _theta_buf(self, n):
        t = self._theta.get(n)
        if t is None:
            t = self._theta[n] = self._torch.zeros(n, self.ld, dtype=self._torch.float64, device=self.device)
        return t

    # -- the reference's sampler signature --------------------------------------------------------------------------------------
    def __call__(self, n, wts, pts):
        torch = self._torch
        k = 0 if wts is None else len(wts)
        if not self.supports(n, k):
            raise ValueError("LinregPosteriorSampler: %d draws for %d weighted points (at most %d and %d)" % (n, k, self.SMAX, self.KMAX))
        st, w_dev = None, self._none
        if k:
            st = self._points(pts)
            w_dev = torch.from_numpy(np.ascontiguousarray(wts, dtype=np.float64)).to(self.device)
        theta = self._theta_buf(n)
        R = self._noise(n)
        if k and not st["low_rank"]:
            rc = self._lib.bcx_linreg_posterior_factor(*self._factor_args(st, w_dev))
            if rc != 0:
                raise self._nat.EngineError(rc, self._lib.bcx_project_last_error().decode())
            a = self._draw_factored_args(theta)
            rbar = self._column_means(R)
            a[6], a[7] = R.data_ptr(), rbar.data_ptr()
            rc = self._lib.bcx_linreg_posterior_draw_factored(*a)
            if rc != 0:
                raise self._nat.EngineError(rc, self._lib.bcx_project_last_error().decode())
            self.factor_status()
        else:
            self._launch(st, w_dev, R, self._column_means(R), theta)
        self.mean = self._tbar
        return theta[:, :self.D]

    # -- SparseVI's device-resident weight optimisation ------------------------------------------------------------------------
    def enqueue_plan(self, n, pts, steps):
        """None when this sampler cannot serve the loop from the device (too many points / draws)."""
        pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        if pts.shape[0] < 1 or not self.supports(n, pts.shape[0]):
            return None
        if 3 * steps * (n + 1) * self.ld * 8 > self.NOISE_BUDGET or steps * (n + 1) > self.ROWS_MAX:
            return None                                     # (the caller's host loop draws step by step)
        return _Plan(self, n, self._points(pts), self._noise_block(steps, n))


class _Plan(object):
    """The draws of ``steps`` consecutive sampler calls at the same points, from weights that live on the device.  The
    normal numbers of all steps are drawn up front; where the per-step kernel for a few points applies (k <= 32,
    csrc/svi.hip lrs_apply_kernel) their image under the prior's factor, G = R U0^T, is formed for all steps in one launch
    and a step is a rank-k correction of its rows."""

    def __init__(self, sampler, n, st, noise):
        s, torch = sampler, sampler._torch
        self.s, self.n, self.st = s, n, st
        self.theta = s._theta_buf(n)
        self._args, self._w_ptr, self._pre = None, None, None
        self.fast = bool(st["low_rank"] and s._lib.bcx_linreg_posterior_apply_ok(st["k"], s.ld))
        self.factored = not st["low_rank"]
        self.set_noise(noise)

    def set_noise(self, noise):
        s, torch, n = self.s, self.s._torch, self.n
        steps = noise.shape[0]
        self.noise, self._args = noise, None
        self.rbar = s._column_means(noise)                  # steps x ld: the column means of every step's normal draws
        self.gscale = None
        if self.fast:
            if s._U0_diag is not None:
                # an isotropic / diagonal prior: G = R U0^T is a column scaling, applied by the kernel as it reads the rows
                self.G, self.Gbar, self.gscale = noise, self.rbar, s._U0_diag
            else:
                # G = R U0^T for the rows of every step and for their means (lrs_draw_kernel with k = 0 and a zero prior mean)
                rows = noise if noise.is_contiguous() else noise.contiguous()
                G, Gbar = torch.empty_like(rows), torch.empty_like(self.rbar)
                s._launch(None, s._none, rows.view(steps * n, s.ld), self.rbar[0], G.view(steps * n, s.ld), mu0=s._zero_mu, tbar=s._scratch_mean)
                s._launch(None, s._none, self.rbar, self.rbar[0], Gbar, mu0=s._zero_mu, tbar=s._scratch_mean)
                self.G, self.Gbar = G, Gbar

    def buffers(self):
        """(draws S x D, their mean): the same two device buffers at every step, rewritten in stream order."""
        return self.theta[:, :self.s.D], self.s._tbar

    def draw(self, w_dev, i):
        """Enqueue the draws for ADAM step ``i`` at the device-resident weights ``w_dev``; returns ``buffers()``."""
        a = self._args
        if a is None or self._w_ptr != w_dev.data_ptr():
            # the argument lists, built once: per step only the two pointers into the noise (or its image) move
            s, st = self.s, self.st
            stream = int(s._torch.cuda.current_stream(s.device).cuda_stream)
            self._w_ptr, self._pre = w_dev.data_ptr(), None
            if self.fast:
                a = [stream, st["k"], s.D, s.ld, w_dev.data_ptr(), st["K0"].data_ptr(), st["xmu0"].data_ptr(), st["y"].data_ptr(),
                     st["X"].data_ptr(), st["XS0"].data_ptr(), s._mu0.data_ptr(), s.sigsq, 0, 0, self.n, self.theta.data_ptr(),
                     s._tbar.data_ptr(), None if self.gscale is None else self.gscale.data_ptr()]
                self._at, self._fn = (12, 13), s._lib.bcx_linreg_posterior_apply
                rows, means = self.G, self.Gbar
            elif self.factored:
                # the D x D form: factor at the current weights (csrc/lrpost.hip), then theta = mu_w + R L^-1
                self._pre = (s._lib.bcx_linreg_posterior_factor, s._factor_args(st, w_dev))
                a = s._draw_factored_args(self.theta)
                self._at, self._fn = (6, 7), s._lib.bcx_linreg_posterior_draw_factored
                rows, means = self.noise, self.rbar
            else:
                a = [stream, st["k"], s.D, s.ld, w_dev.data_ptr(), st["K0"].data_ptr(), st["xmu0"].data_ptr(), st["y"].data_ptr(),
                     st["XU0"].data_ptr(), st["XS0"].data_ptr(), s._U0T.data_ptr(), s._mu0.data_ptr(), s.sigsq, 0, 0, self.n,
                     self.theta.data_ptr(), s._tbar.data_ptr()]
                self._at, self._fn = (13, 14), s._lib.bcx_linreg_posterior_draw
                rows, means = self.noise, self.rbar
            self._r0, self._rstep = rows.data_ptr(), rows.stride(0) * 8
            self._b0, self._bstep = means.data_ptr(), means.stride(0) * 8
            self._args = a
        if self._pre is not None:
            rc = self._pre[0](*self._pre[1])
            if rc != 0:
                raise self.s._nat.EngineError(rc, self.s._lib.bcx_project_last_error().decode())
        a[self._at[0]], a[self._at[1]] = self._r0 + i * self._rstep, self._b0 + i * self._bstep
        rc = self._fn(*a)
        if rc != 0:
            raise self.s._nat.EngineError(rc, self.s._lib.bcx_project_last_error().decode())
        return self.buffers()

    def check(self):
        """After the loop's read-back: did every factorisation of the loop complete (csrc/lrpost.hip's status word)?"""
        if self.factored:
            self.s.factor_status()
