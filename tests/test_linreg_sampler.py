"""CPU: the device form of the weighted conjugate posterior of examples/common/model_linreg.py (low-rank update of the
prior's factor, run here on torch's CPU device) against the NumPy form that restates the reference's
examples/common/model_linreg.py:26-41 (pinned to reference outputs by tests/test_host_golden.py, F12)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))


@pytest.mark.parametrize("D,k", ((7, 3), (30, 1), (31, 12), (12, 32), (9, 40)))
def test_device_posterior_and_draws(D, k):
    import torch
    import model_linreg
    rs = np.random.RandomState(D * 100 + k)
    mu0 = rs.randn(D)
    A0 = rs.randn(D, D)
    Sig0 = A0.dot(A0.T) + D * np.eye(D)
    sigsq = 0.37
    pts = rs.randn(k, D + 1)
    wts = np.abs(rs.randn(k)) * 50.0
    wts[0] = 0.0                                                   # a zero weight is a legal state (new coreset point)
    smp = model_linreg.posterior_sampler(mu0, Sig0, sigsq, device="cpu", seed=5)
    mu_ref, U_ref = model_linreg.weighted_posterior(mu0, np.linalg.inv(Sig0), sigsq, pts, wts)
    cov_ref = U_ref.dot(U_ref.T)
    for _ in range(2):                                             # second call: the cached state of the points
        mu, U = smp.posterior(wts, pts)
        mu, U = mu.numpy(), U.numpy()
        np.testing.assert_allclose(mu, mu_ref, rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(U.dot(U.T), cov_ref, rtol=1e-7, atol=1e-10)
    # the draws are mu + R U^T for the R the sampler's generator produces
    n = 64
    draws = smp(n, wts, pts)
    assert tuple(draws.shape) == (n, D) and draws.stride(0) % 2 == 0 and draws.stride(1) == 1
    if k <= 32:
        g = torch.Generator(device="cpu")
        g.manual_seed(5)
        R = torch.randn(n, D + D % 2, dtype=torch.float64, generator=g).numpy()[:, :D]
        np.testing.assert_allclose(draws.numpy(), mu_ref + R.dot(U.T), rtol=1e-7, atol=1e-8)
    # other weights on the same points, then other points of the same shape: nothing stale
    wts2 = wts * 0.5 + 1.0
    mu2, _ = smp.posterior(wts2, pts)
    np.testing.assert_allclose(mu2.numpy(), model_linreg.weighted_posterior(mu0, np.linalg.inv(Sig0), sigsq, pts, wts2)[0], rtol=1e-7, atol=1e-9)
    pts3 = pts + 0.25
    mu3, _ = smp.posterior(wts2, pts3)
    np.testing.assert_allclose(mu3.numpy(), model_linreg.weighted_posterior(mu0, np.linalg.inv(Sig0), sigsq, pts3, wts2)[0], rtol=1e-7, atol=1e-9)
    # no points: the prior
    mu4, U4 = smp.posterior(np.array([]), np.array([]))
    np.testing.assert_allclose(mu4.numpy(), mu0, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(U4.numpy().dot(U4.numpy().T), Sig0, rtol=1e-9)
    assert tuple(smp(5, np.array([]), np.array([])).shape) == (5, D)


@pytest.mark.parametrize("D,k", ((9, 4), (30, 1), (12, 12), (7, 20)))
def test_rank_k_cholesky_form_of_the_posterior(D, k):
    """The algebra csrc/svi.hip evaluates on the device (lrs_draw_kernel / lrs_apply_kernel), restated in NumPy and checked
    against the direct form of the weighted conjugate posterior (tests/models.py, pinned to reference outputs by F12):
        Sigma_w = U0 (I + C^T C)^-1 U0^T,  C = diag(s) X U0,  (I + C^T C)^-1 = F F^T,  F = I - C^T T C,
        T = L^-T (I + L)^-1,  L L^T = I + C C^T;     Uw^T = U0^T - (X U0)^T [diag(s) T^T diag(s)] (X Sig0),
        mu_w = mu0 + (X Sig0)^T (c - s * a'),  c = w y / sigsq,  a' = (L L^T)^-1 [s * (X mu0 + K0 c)],  K0 = (X U0)(X U0)^T."""
    from models import linreg_weighted_post
    rs = np.random.RandomState(17 * D + k)
    mu0 = rs.randn(D)
    A0 = rs.randn(D, D)
    Sig0 = A0.dot(A0.T) / D + np.eye(D)
    sigsq = 0.37
    pts = rs.randn(k, D + 1)
    w = np.abs(rs.randn(k)) * 30.0
    w[0] = 0.0
    X, y = pts[:, :-1], pts[:, -1]
    U0 = np.linalg.cholesky(Sig0)
    XU0, XS0 = X.dot(U0), X.dot(Sig0)
    K0 = XU0.dot(XU0.T)
    s, c = np.sqrt(w / sigsq), w * y / sigsq
    L = np.linalg.cholesky(np.eye(k) + np.outer(s, s) * K0)
    a1 = np.linalg.solve(L.dot(L.T), s * (X.dot(mu0) + K0.dot(c)))
    mu = mu0 + (c - s * a1).dot(XS0)
    Tt = np.linalg.solve((np.eye(k) + L).T, np.linalg.inv(L))             # T^T = (I + L)^-T L^-1
    B2 = (s[:, None] * Tt * s[None, :]).dot(XS0)
    UwT = U0.T - XU0.T.dot(B2)
    mu_ref, U_ref = linreg_weighted_post(mu0, np.linalg.inv(Sig0), sigsq, pts, w)
    np.testing.assert_allclose(mu, mu_ref, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(UwT.T.dot(UwT), U_ref.dot(U_ref.T), rtol=1e-9, atol=1e-11)
    # the per-step form of the loop: theta = mu_w + G - (G X^T) B2 for G = R U0^T
    R = rs.randn(6, D)
    G = R.dot(U0.T)
    np.testing.assert_allclose(mu + G - G.dot(X.T).dot(B2), mu + R.dot(UwT), rtol=1e-11, atol=1e-12)
