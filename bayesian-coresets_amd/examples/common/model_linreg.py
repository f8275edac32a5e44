"""Gaussian linear regression: the conjugate posterior of a WEIGHTED data set and a sampler for it, the model
behind the reference's linear-regression experiment (examples/linear_regression/main.py:124-147 with
examples/common/model_linreg.py).  Rows are z = [x, y]; prior theta ~ N(mu0, Sig0); noise variance sigsq.

    Sigma_w^-1 = Sig0^-1 + X^T diag(w) X / sigsq,    mu_w = Sigma_w (Sig0^-1 mu0 + X^T diag(w) y / sigsq)

``posterior_sampler(..., device=None)`` is the host (NumPy) sampler; with ``device=`` a torch device the same
posterior is formed on the GPU (for up to 32 weighted points as a low-rank update of the prior's factor, otherwise by a
Cholesky of the D x D system) and the draws come back as a device tensor, which ``bc.DeviceProjector`` uses in place.
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
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(0 if seed is None else int(seed))
    mu0_d = torch.from_numpy(mu0).to(dev)
    S0inv_d = torch.from_numpy(Sig0inv).to(dev)
    eye = torch.eye(mu0.shape[0], dtype=torch.float64, device=dev)

    # The prior's factor, once: Sig0^-1 = L0 L0^T, U0 = L0^-T (so Sig0 = U0 U0^T).  A weighted coreset of k points is a
    # rank-k update  A = Sig0^-1 + B^T B,  B = diag(sqrt(w / sigsq)) X,  and with  C = B U0  (k x D)
    #     A^-1 = U0 (I + C^T C)^-1 U0^T,      (I + C^T C)^(-1/2) = I + W diag(d) W^T,
    # W = C^T Q, d_i = ((1 + l_i)^(-1/2) - 1) / l_i from the k x k eigenproblem C C^T = Q diag(l) Q^T (host, microseconds).
    # SparseVI calls the sampler once per ADAM step with a handful of points: this replaces a 301 x 301 Cholesky and a
    # triangular solve (rocSOLVER small-matrix kernels, ~0.7 ms of device time per call) by two D x D x k products.
    import scipy.linalg as sl
    L0 = np.linalg.cholesky(Sig0inv)
    U0 = sl.solve_triangular(L0, np.eye(mu0.shape[0]), lower=True, check_finite=False).T
    U0_d = torch.from_numpy(np.ascontiguousarray(U0)).to(dev)
    rhs0 = Sig0inv.dot(mu0)
    LOWRANK_MAX = 32

    def posterior_lowrank(wts, pts):
        pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
        wts = np.asarray(wts, dtype=np.float64)
        X, y = pts[:, :-1], pts[:, -1]
        C = (np.sqrt(wts / sigsq)[:, None] * X).dot(U0)
        lam, Q = np.linalg.eigh(C.dot(C.T))
        lam = np.maximum(lam, 0.0)
        d = np.where(lam > 1e-290, (1.0 / np.sqrt(1.0 + lam) - 1.0) / np.where(lam > 1e-290, lam, 1.0), -0.5)
        W = torch.from_numpy(np.ascontiguousarray(C.T.dot(Q))).to(dev)                 # D x k
        rhs = torch.from_numpy(rhs0 + (wts * y).dot(X) / sigsq).to(dev)
        U = U0_d + ((U0_d @ W) * torch.from_numpy(d).to(dev)) @ W.T                    # U0 (I + W diag(d) W^T)
        return U @ (U.T @ rhs), U

    def posterior(wts, pts):
        if wts is not None and 0 < len(wts) <= LOWRANK_MAX and np.all(np.asarray(wts) >= 0):
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

    def sampler(n, wts, pts):
        mu, U = posterior(wts, pts)
        return mu + torch.randn(n, mu.shape[0], dtype=torch.float64, device=dev, generator=gen) @ U.T
    sampler.posterior = posterior
    return sampler
