"""Device-resident sampler of the weighted conjugate posterior of Gaussian linear regression -- the ``sampler`` argument
of ``DeviceProjector("linreg", ...)`` for the reference's linear-regression experiment (examples/linear_regression/
main.py:124-147: prior theta ~ N(mu0, Sig0), noise variance sigsq, rows z = [x, y]):

    Sigma_w^-1 = Sig0^-1 + X^T diag(w) X / sigsq,      mu_w = Sigma_w (Sig0^-1 mu0 + X^T diag(w) y / sigsq)

SparseVI calls its sampler once per ADAM step (sparsevi.py:25 -> projector.update), 1 + opt_itrs times per greedy step,
with the SAME points and new weights.  The draws are formed by one kernel (csrc/svi.hip: a rank-k correction of the
prior's factor through a k x k Cholesky, k = number of weighted points <= 64), in two ways:

* ``sampler(n, wts, pts)``: the reference's sampler signature -- uploads the k weights, returns the draws as a device
  tensor (``DeviceProjector`` uses them in place);
* ``sampler.enqueue_plan(n, pts, steps)``: for ``SparseVICoreset``'s device-resident weight optimisation -- the plan's
  ``draw(w_dev, i)`` takes the weights FROM the device and enqueues the kernel: no host synchronisation per ADAM step.

Everything that depends on the points alone (X U0, X Sig0, their Gram matrix, X mu0) is formed on the host when the
points change (once per greedy step) and uploaded in one piece.  There is no CPU fallback."""
import numpy as np


class LinregPosteriorSampler(object):
    KMAX = 64      # weighted points the kernel takes (csrc/svi.hip LRS_KMAX)
    SMAX = 1024    # draws per call

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
        self._mu0 = torch.from_numpy(self.mu0).to(self.device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(0 if seed is None else int(seed))
        self._pts_key, self._pts_state = None, None
        self._theta, self._tbar = {}, torch.empty(D, dtype=torch.float64, device=self.device)
        self._none = torch.zeros(1, dtype=torch.float64, device=self.device)

    # -- noise: separate hooks so that a test can feed both entry points the same numbers --------------------------------
    def _noise(self, n):
        return self._torch.randn(n, self.ld, dtype=self._torch.float64, device=self.device, generator=self.gen)

    def _noise_block(self, steps, n):
        return self._torch.randn(steps, n, self.ld, dtype=self._torch.float64, device=self.device, generator=self.gen)

    # -- per-point state ------------------------------------------------------------------------------------------------------
    def supports(self, n, k):
        return 0 <= k <= self.KMAX and 1 <= n <= self.SMAX

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
        XU0, XS0 = np.zeros((k, self.ld)), np.zeros((k, self.ld))
        XU0[:, :self.D] = X.dot(self.U0)
        XS0[:, :self.D] = X.dot(self.Sig0)
        K0 = XU0.dot(XU0.T)
        pad = (k * k + 2 * k) % 2                           # the two k x ld blocks start on 16-byte boundaries
        blob = np.concatenate((K0.ravel(), X.dot(self.mu0), y, np.zeros(pad), XU0.ravel(), XS0.ravel()))
        d = torch.from_numpy(blob).to(self.device)
        o = k * k
        st = {"k": k, "K0": d[:o], "xmu0": d[o:o + k], "y": d[o + k:o + 2 * k]}
        o += 2 * k + pad
        st["XU0"], st["XS0"] = d[o:o + k * self.ld], d[o + k * self.ld:o + 2 * k * self.ld]
        st["blob"] = d
        self._pts_key, self._pts_state = pts.copy(), st
        return st

    def _launch(self, st, w_dev, R, theta):
        k = st["k"] if st is not None else 0
        lib, dp = self._lib, (lambda key: st[key].data_ptr() if k else None)
        stream = int(self._torch.cuda.current_stream(self.device).cuda_stream)
        rc = lib.bcx_linreg_posterior_draw(stream, k, self.D, self.ld, w_dev.data_ptr() if k else None, dp("K0"), dp("xmu0"), dp("y"),
                                           dp("XU0"), dp("XS0"), self._U0T.data_ptr(), self._mu0.data_ptr(), self.sigsq,
                                           R.data_ptr(), R.shape[0], theta.data_ptr(), self._tbar.data_ptr())
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
        self._launch(st, w_dev, self._noise(n), theta)
        self.mean = self._tbar
        return theta[:, :self.D]

    # -- SparseVI's device-resident weight optimisation ------------------------------------------------------------------------
    def enqueue_plan(self, n, pts, steps):
        """None when this sampler cannot serve the loop from the device (too many points / draws)."""
        pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        if pts.shape[0] < 1 or not self.supports(n, pts.shape[0]):
            return None
        return _Plan(self, n, self._points(pts), self._noise_block(steps, n))


class _Plan(object):
    def __init__(self, sampler, n, st, noise):
        self.s, self.n, self.st, self.noise = sampler, n, st, noise
        self.theta = sampler._theta_buf(n)

    def draw(self, w_dev, i):
        """Enqueue the draws for ADAM step ``i`` at the device-resident weights; (draws S x D, their mean)."""
        self.s._launch(self.st, w_dev, self.noise[i], self.theta)
        return self.theta[:, :self.s.D], self.s._tbar
