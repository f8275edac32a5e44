"""Device-resident sampler of the Laplace approximation of a WEIGHTED logistic / Poisson regression posterior -- the
``sampler`` argument of ``DeviceProjector("logistic" | "poisson", ...)`` for the reference's logistic / Poisson regression
experiment (examples/logistic_poisson_regression/main.py:15-41 ``get_laplace``, :155-162 ``sampler_w``: standard-normal
prior, mode of the weighted log joint, covariance = inverse negative Hessian there).

As ``bc.LinregPosteriorSampler`` it serves two callers:

* ``sampler(n, wts, pts)``: the reference's sampler signature -- uploads the k weights, returns the draws as a device tensor;
* ``sampler.enqueue_plan(n, pts, steps)``: for ``SparseVICoreset``'s device-resident weight optimisation -- ``draw(w_dev, i)``
  takes the weights FROM the device: no host synchronisation (no SciPy minimisation, no upload) per ADAM step.

One launch of one workgroup per call (csrc/laplace.hip: damped Newton on the points in LDS, the D x D Newton systems by an
in-register Cholesky, then the draws mu + R L^-1); the normal numbers come from the library's counter-based generator.  The
reference finds the mode with SciPy's BFGS to gtol 1e-5; this is Newton to |step| < 1e-10 on the same objective (the
iteration of examples/common/model_lr.py / model_poiss.py ``laplace_fit``, which the tests pin to the reference's outputs).
Limits: D <= 32 parameters, the points within 96 KiB of LDS (``supports``); beyond them ``enqueue_plan`` declines (the host
loop then runs) and the call form raises.  There is no CPU fallback."""
import numpy as np

from .linreg_sampler import _DeviceNormals

FAMILIES = {"logistic": 0, "poisson": 1}


class LaplacePosteriorSampler(_DeviceNormals):
    SMAX = 4096        # draws per call (DeviceProjector's own limit on the projection dimension)
    NOISE_BUDGET = 2 << 30

    def __init__(self, family, D, device="cuda", seed=None, tol=1e-10, max_iter=100):
        import torch
        from . import _native
        if family not in FAMILIES:
            raise ValueError("family must be 'logistic' or 'poisson'")
        self._torch, self._nat = torch, _native
        self._lib = _native.load()
        if not torch.cuda.is_available():
            raise RuntimeError("LaplacePosteriorSampler needs a GPU (there is no CPU fallback)")
        self.family, self._fam = family, FAMILIES[family]
        self.device = torch.device(device)
        self.D = int(D)
        self.ld = self.D + (self.D % 2)                     # rows of the draws start on 16-byte boundaries
        self.cols = self.D + (1 if family == "poisson" else 0)      # columns of a point: [x, y] for Poisson, y x for logistic
        self.tol, self.max_iter = float(tol), int(max_iter)
        self._seed, self._offset = (0 if seed is None else int(seed)) & 0xFFFFFFFFFFFFFFFF, 0
        self._mu = torch.zeros(self.D, dtype=torch.float64, device=self.device)       # the last mode (seeds a plan's next step)
        self._tbar = torch.zeros(self.D, dtype=torch.float64, device=self.device)
        self._status = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._none = torch.zeros(2, dtype=torch.float64, device=self.device)
        self._theta = {}
        self._pts_key, self._pts_dev = None, None

    def supports(self, n, k):
        return 1 <= n <= self.SMAX and bool(self._lib.bcx_laplace_sampler_ok(int(k), self.D))

    def _theta_buf(self, n):
        t = self._theta.get(n)
        if t is None:
            t = self._theta[n] = self._torch.zeros(n, self.ld, dtype=self._torch.float64, device=self.device)
        return t

    def _points(self, pts):
        pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        if pts.shape[1] != self.cols:
            raise ValueError("points have %d columns, the %s model takes %d" % (pts.shape[1], self.family, self.cols))
        if self._pts_key is None or self._pts_key.shape != pts.shape or not np.array_equal(self._pts_key, pts):
            self._pts_key = pts.copy()
            self._pts_dev = self._torch.from_numpy(np.ascontiguousarray(pts)).to(self.device)
        return self._pts_dev

    def _args(self, k, w_dev, pts_dev, warm, R, rbar, theta):
        stream = int(self._torch.cuda.current_stream(self.device).cuda_stream)
        return [stream, self._fam, k, self.D, w_dev.data_ptr() if k else None, pts_dev.data_ptr() if k else None, self.cols,
                self._mu.data_ptr(), int(warm), self.tol, self.max_iter, R.data_ptr(), rbar.data_ptr(), theta.shape[0], self.ld,
                theta.data_ptr(), self._tbar.data_ptr(), self._status.data_ptr()]

    def check(self):
        """Synchronises; raises if the last fit did not converge or met no positive definite Newton matrix."""
        st = self._status.cpu().numpy()
        if st[0] != 0:
            raise self._nat.EngineError(self._nat.ERR_STATE, "Laplace fit on the device: %s after %d Newton steps"
                                        % ("iteration limit" if st[0] == 1 else "no positive definite Newton matrix", int(st[1])))
        return int(st[1])

    # -- the reference's sampler signature --------------------------------------------------------------------------------------
    def __call__(self, n, wts, pts):
        torch = self._torch
        k = 0 if wts is None or pts is None else len(wts)
        if k and np.asarray(pts).size == 0:
            k = 0
        if not self.supports(n, k):
            raise ValueError("LaplacePosteriorSampler: %d draws for %d weighted points of %d parameters (D <= 32, the points within "
                             "96 KiB of LDS, at most %d draws)" % (n, k, self.D, self.SMAX))
        pts_dev, w_dev = None, self._none
        if k:
            pts_dev = self._points(pts)
            w_dev = torch.from_numpy(np.ascontiguousarray(wts, dtype=np.float64)).to(self.device)
        theta = self._theta_buf(n)
        R = self._noise(n)
        rc = self._lib.bcx_laplace_sampler(*self._args(k, w_dev, pts_dev, False, R, self._column_means(R), theta))
        if rc != 0:
            raise self._nat.EngineError(rc, self._lib.bcx_project_last_error().decode())
        self.newton_steps = self.check()
        self.mean = self._tbar
        return theta[:, :self.D]

    def posterior(self, wts, pts):
        """(mode, covariance factor W with Sigma = W^T W) as ndarrays: the rows of W come out as the draws of unit noise."""
        torch = self._torch
        keep = self._noise
        try:
            eye = torch.zeros(self.D + 1, self.ld, dtype=torch.float64, device=self.device)
            eye[:self.D, :self.D] = torch.eye(self.D, dtype=torch.float64, device=self.device)
            self._noise = lambda n: eye
            th = self(self.D + 1, wts, pts).cpu().numpy()
        finally:
            self._noise = keep
        mu = th[self.D]
        return mu, th[:self.D] - mu

    # -- SparseVI's device-resident weight optimisation ------------------------------------------------------------------------
    def enqueue_plan(self, n, pts, steps):
        """None when this sampler cannot serve the loop from the device (too many points / parameters / draws)."""
        pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        if pts.shape[0] < 1 or not self.supports(n, pts.shape[0]) or 2 * steps * (n + 1) * self.ld * 8 > self.NOISE_BUDGET:
            return None
        return _LaplacePlan(self, n, self._points(pts), self._noise_block(steps, n))


class _LaplacePlan(object):
    """The draws of ``steps`` consecutive sampler calls at the same points, from weights that live on the device; every step
    after the first starts its Newton iteration at the mode of the step before."""

    def __init__(self, sampler, n, pts_dev, noise):
        self.s, self.n, self.pts_dev = sampler, n, pts_dev
        self.theta = sampler._theta_buf(n)
        self.set_noise(noise)

    def set_noise(self, noise):
        self.noise, self._a = noise, None
        self.rbar = self.s._column_means(noise)

    def buffers(self):
        return self.theta[:, :self.s.D], self.s._tbar

    def draw(self, w_dev, i):
        s = self.s
        a = self._a
        if a is None or self._w_ptr != w_dev.data_ptr():
            a = self._a = s._args(self.pts_dev.shape[0], w_dev, self.pts_dev, False, self.noise, self.rbar, self.theta)
            self._w_ptr = w_dev.data_ptr()
            self._r0, self._rstep = self.noise.data_ptr(), self.noise.stride(0) * 8
            self._b0, self._bstep = self.rbar.data_ptr(), self.rbar.stride(0) * 8
        a[8] = 1 if i > 0 else 0                            # (warm: the mode of the previous ADAM step)
        a[11], a[12] = self._r0 + i * self._rstep, self._b0 + i * self._bstep
        rc = s._lib.bcx_laplace_sampler(*a)
        if rc != 0:
            raise s._nat.EngineError(rc, s._lib.bcx_project_last_error().decode())
        return self.buffers()

    def check(self):
        """After the loop's read-back: the LAST fit's status (a fit that failed leaves NaNs or stale draws behind and fails
        the ones after it)."""
        self.s.check()
