"""Gaussian linear regression: the conjugate posterior of a WEIGHTED data set and a sampler for it, the model
behind the reference's linear-regression experiment (examples/linear_regression/main.py:124-147 with
examples/common/model_linreg.py).  Rows are z = [x, y]; prior theta ~ N(mu0, Sig0); noise variance sigsq.

    Sigma_w^-1 = Sig0^-1 + X^T diag(w) X / sigsq,    mu_w = Sigma_w (Sig0^-1 mu0 + X^T diag(w) y / sigsq)

``posterior_sampler(..., device=None)`` is the host (NumPy) sampler; with ``device=`` a torch device the same
algebra runs on the GPU and the draws come back as a device tensor, which ``bc.DeviceProjector`` uses in place.
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

    def posterior(wts, pts):
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
