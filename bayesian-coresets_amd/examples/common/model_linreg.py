"""Gaussian linear regression: the conjugate posterior of a WEIGHTED data set and a sampler for it, the model
behind the reference's linear-regression experiment (examples/linear_regression/main.py:124-147 with
examples/common/model_linreg.py).  Rows are z = [x, y]; prior theta ~ N(mu0, Sig0); noise variance sigsq.

    Sigma_w^-1 = Sig0^-1 + X^T diag(w) X / sigsq,    mu_w = Sigma_w (Sig0^-1 mu0 + X^T diag(w) y / sigsq)

``posterior_sampler(..., device=None)`` is the host (NumPy) sampler; with ``device=`` a torch device the same
posterior is formed on the GPU (by ``bc.LinregPosteriorSampler``: a low-rank update of the prior's factor for a few weighted
points, the reference's own Cholesky of the D x D system beyond) and the draws come back as a device tensor,
which ``bc.DeviceProjector`` uses in place.
SparseVI calls the sampler once per ADAM step (sparsevi.py:25 via projector.update); at D = 301 the host
version costs ~29 ms per call on a 128-thread box (SciPy triangular solve + Cholesky of a 301 x 301 matrix)
against 2.9 ms for the whole N = 625k projection it feeds, so the sampler is what one moves next.
"""
import numpy as np


def weighted_posterior(mu0, Sig0inv, sigsq, pts, wts):
    """(mu, U) with Sigma = U U^T (NumPy)."""
    import scipy.linalg as sl
    D = mu0.shape[0]
    A, rhs = Sig0inv.copy(), Sig0inv.dot(mu0)
    if wts is not None and len(wts):
        pts = np.atleast_2d(pts)
        X, y = pts[:, :-1], pts[:, -1]
        A = A + (wts[:, None] * X).T.dot(X) / sigsq
        rhs = rhs + (wts * y).dot(X) / sigsq
    L = np.linalg.cholesky(A)
    U = sl.solve_triangular(L, np.eye(D), lower=True, check_finite=False).T      # A^-1 = U U^T
    return U.dot(U.T.dot(rhs)), U


def posterior_sampler(mu0, Sig0, sigsq, device=None, seed=None):
    """sampler(n, wts, pts) -> n x D draws from the weighted posterior (ndarray, or a tensor on ``device``)."""
    mu0 = np.asarray(mu0, dtype=np.float64)
    Sig0inv = np.linalg.inv(np.asarray(Sig0, dtype=np.float64))
    if device is None:
        def sampler(n, wts, pts):
            mu, U = weighted_posterior(mu0, Sig0inv, sigsq, pts, None if wts is None else np.asarray(wts))
            return mu + np.random.randn(n, mu.shape[0]).dot(U.T)
        return sampler

    import torch
    import scipy.linalg as sl
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0 if seed is None else int(seed))
    D = mu0.shape[0]
    Dp = D + (D % 2)      # rows of the returned draws start on 16-byte boundaries (bc.DeviceProjector then uses them in place)
    mu0_d = torch.from_numpy(mu0).to(dev)
    S0inv_d = torch.from_numpy(Sig0inv).to(dev)
    eye = torch.eye(D, dtype=torch.float64, device=dev)

    # The prior's factor, once: Sig0^-1 = L0 L0^T, U0 = L0^-T (so Sig0 = U0 U0^T).  A weighted coreset of k points is a
    # rank-k update  A = Sig0^-1 + B^T B,  B = diag(s) X,  s = sqrt(w / sigsq),  and with  C = B U0  (k x D)
    #     A^-1 = U0 (I + C^T C)^-1 U0^T,      (I + C^T C)^(-1/2) = I + W diag(d) W^T,      (I + C^T C)^-1 = I + W diag(e) W^T,
    # W = C^T Q, d_i = ((1 + l_i)^(-1/2) - 1) / l_i, e_i = -1 / (1 + l_i) from the k x k eigenproblem C C^T = Q diag(l) Q^T.
    # SparseVI calls the sampler once per ADAM step with the SAME points and new weights: everything that depends on the
    # points alone (X U0 on the device, its k x k Gram on the host) is kept while the points stay the same; a call then costs a
    # k x k eigenproblem on the host, ONE small upload (diag(s) Q, d, e, the right-hand side) and eleven small device
    # kernels -- against a 301 x 301 Cholesky + triangular solve (rocSOLVER small-matrix kernels, ~0.7 ms) per call.
    L0 = np.linalg.cholesky(Sig0inv)
    U0 = sl.solve_triangular(L0, np.eye(D), lower=True, check_finite=False).T
    U0p = np.zeros((Dp, Dp))
    U0p[:D, :D] = U0
    U0_d = torch.from_numpy(U0p).to(dev)
    U0T_d = U0_d.T.contiguous()
    rhs0 = Sig0inv.dot(mu0)
    LOWRANK_MAX = 32
    cache = {"pts": None}
    pinned = {}

    def staged(vec):
        """one host->device copy of a packed vector through a pinned buffer (event-guarded reuse)"""
        n = vec.shape[0]
        if dev.type != "cuda":
            return torch.from_numpy(np.ascontiguousarray(vec)).to(dev)
        slot = pinned.get(n)
        if slot is None:
            slot = pinned[n] = (torch.empty(n, dtype=torch.float64).pin_memory(), torch.empty(n, dtype=torch.float64, device=dev),
                                torch.cuda.Event())
        host, devbuf, ev = slot
        ev.synchronize()                       # the previous copy out of this buffer has completed
        host.numpy()[:] = vec
        devbuf.copy_(host, non_blocking=True)
        ev.record(torch.cuda.current_stream(dev))
        return devbuf

    def points_state(pts):
        c = cache
        if c["pts"] is None or c["pts"].shape != pts.shape or not np.array_equal(c["pts"], pts):
            X, y = pts[:, :-1], pts[:, -1]
            XU0 = X.dot(U0)                                         # k x D
            XU0Tp = np.zeros((Dp, pts.shape[0]))
            XU0Tp[:D] = XU0.T
            c.update(pts=pts.copy(), X=X.copy(), y=y.copy(), K0=XU0.dot(XU0.T), XU0T_d=torch.from_numpy(XU0Tp).to(dev))
        return c

    def lowrank_factors(wts, pts):
        """(W (Dp x k), d, e, t = U0^T rhs) on the device"""
        pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        wts = np.asarray(wts, dtype=np.float64)
        c = points_state(pts)
        k = wts.shape[0]
        sv = np.sqrt(wts / sigsq)
        lam, Q = np.linalg.eigh(c["K0"] * np.outer(sv, sv))
        lam = np.maximum(lam, 0.0)
        safe = np.where(lam > 1e-290, lam, 1.0)
        d = np.where(lam > 1e-290, (1.0 / np.sqrt(1.0 + lam) - 1.0) / safe, -0.5)
        e = -1.0 / (1.0 + lam)
        rhs = np.zeros(Dp)
        rhs[:D] = rhs0 + (wts * c["y"]).dot(c["X"]) / sigsq
        buf = staged(np.concatenate(((sv[:, None] * Q).ravel(), d, e, rhs)))
        sQ, d_d, e_d, rhs_d = buf[:k * k].view(k, k), buf[k * k:k * k + k], buf[k * k + k:k * k + 2 * k], buf[k * k + 2 * k:]
        W = c["XU0T_d"] @ sQ                                        # Dp x k
        return W, d_d, e_d, torch.mv(U0T_d, rhs_d)

    def lowrank_mean(W, e_d, t):
        a = torch.mv(W.T, t)
        a *= e_d
        return torch.mv(U0_d, torch.addmv(t, W, a))                 # mu = U0 (I + W diag(e) W^T) U0^T rhs

    def posterior_lowrank(wts, pts):
        W, d_d, e_d, t = lowrank_factors(wts, pts)
        U = U0_d + ((U0_d @ W) * d_d) @ W.T                         # U0 (I + W diag(d) W^T)
        return lowrank_mean(W, e_d, t)[:D], U[:D, :D]

    def use_lowrank(wts):
        return wts is not None and 0 < len(wts) <= LOWRANK_MAX and np.all(np.asarray(wts) >= 0)

    def posterior(wts, pts):
        if use_lowrank(wts):
            return posterior_lowrank(wts, pts)
        A, rhs = S0inv_d.clone(), S0inv_d @ mu0_d
        if wts is not None and len(wts):
            P = torch.as_tensor(np.atleast_2d(np.asarray(pts, dtype=np.float64)), device=dev)
            w = torch.as_tensor(np.asarray(wts, dtype=np.float64), device=dev)
            X, y = P[:, :-1], P[:, -1]
            A = A + (X * w[:, None]).T @ X / sigsq
            rhs = rhs + (w * y) @ X / sigsq
        L = torch.linalg.cholesky(A)
        U = torch.linalg.solve_triangular(L, eye, upper=False).T
        return U @ (U.T @ rhs), U

    # On a GPU the draws come from the library's own kernels (bc.LinregPosteriorSampler: a rank-k correction of the prior's factor
    # inside one workgroup for a few points, csrc/svi.hip; the reference's own D x D Cholesky form beyond, csrc/lrpost.hip -- up to
    # 4096 points and D = 1024, its normal numbers from the library's counter-based generator) and -- through `enqueue_plan` -- from
    # weights that never leave the device: SparseVI then enqueues its whole ADAM loop.  The torch forms below remain for
    # torch's CPU device and for sizes beyond those limits.
    fast = None
    if dev.type == "cuda":
        import bayesiancoresets_amd as bc
        fast = bc.LinregPosteriorSampler(mu0, Sig0, sigsq, device=dev, seed=seed)

    def sampler(n, wts, pts):
        k = 0 if wts is None else len(wts)
        if fast is not None and fast.supports(n, k) and (k == 0 or np.all(np.asarray(wts) >= 0)):
            return fast(n, wts, pts)
        if use_lowrank(wts):
            # draws = mu + R U^T with U = U0 (I + W diag(d) W^T), never forming U:  (R + ((R W) d) W^T) U0^T + mu
            W, d_d, e_d, t = lowrank_factors(wts, pts)
            mu = lowrank_mean(W, e_d, t)
            R = torch.randn(n, Dp, dtype=torch.float64, device=dev, generator=gen)
            B = R @ W
            B *= d_d
            return torch.addmm(mu, torch.addmm(R, B, W.T), U0T_d)[:, :D]
        mu, U = posterior(wts, pts)
        out = torch.zeros(n, Dp, dtype=torch.float64, device=dev)
        out[:, :D] = mu + torch.randn(n, D, dtype=torch.float64, device=dev, generator=gen) @ U.T
        return out[:, :D]
    sampler.posterior = posterior
    if fast is not None:
        sampler.enqueue_plan = fast.enqueue_plan
    return sampler


def tangent_space_projector(bc, bV, mu0, Sig0, sigsq):
    """The EXACT tangent-space projection of this model (examples/linear_regression/main.py:158-185, `LinRegProjector`), as a
    ``bc.Projector``.  With theta = mu_w + U_w eps the log-likelihood of a point is -(nu - beta.eps)^2 / (2 sigsq),
    beta = U_w^T x, nu = y - x.mu_w, and its centred version lives in the span of eps and eps eps^T: the vector

        [ nu beta ,  vec(beta beta^T) / sqrt(2) ] / sigsq            (D + D^2 numbers)

    has exactly the inner products Cov_eps(ll_n, ll_m) -- no Monte-Carlo samples.  As in the reference the quadratic block is
    formed from the projection of beta onto ``bV`` (D x p: leading eigenvectors of X^T X; main.py:110-111), p^2 numbers
    instead of D^2.  ``update(wts, pts)`` moves the tangent point to the weighted posterior (the prior when the coreset is
    empty, main.py:172-176).  Rows come out as a host array of D + p^2 columns -- e.g. 10301 at the defaults: a row length
    the device engine takes as any other."""
    mu0 = np.asarray(mu0, dtype=np.float64)
    Sig0 = np.asarray(Sig0, dtype=np.float64)
    Sig0inv = np.linalg.inv(Sig0)
    bV = np.asarray(bV, dtype=np.float64)

    class TangentSpaceProjector(bc.Projector):
        def __init__(self):
            self.bV = bV
            self.update(None, None)

        def update(self, wts, pts):
            if wts is None or pts is None or np.asarray(pts).shape[0] == 0:
                self.muw, self.USigw = mu0, np.linalg.cholesky(Sig0)           # any M with Sigma = M M^T serves
            else:
                self.muw, self.USigw = weighted_posterior(mu0, Sig0inv, sigsq, np.atleast_2d(pts), np.asarray(wts, dtype=np.float64))

        def project(self, pts, grad=False):
            if grad:
                raise NotImplementedError("the exact projector has no gradient form (nor has the reference's)")
            pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
            X, Y = pts[:, :-1], pts[:, -1]
            beta = X.dot(self.USigw)
            nu = Y - X.dot(self.muw)
            bp = beta.dot(self.bV)
            quad = (bp[:, :, None] * bp[:, None, :]).reshape(pts.shape[0], -1)
            return np.hstack((nu[:, None] * beta, quad / np.sqrt(2.0))) / sigsq

    return TangentSpaceProjector()
