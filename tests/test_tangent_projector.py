"""The exact tangent-space projector of the linear-regression experiment (examples/common/model_linreg.py
`tangent_space_projector`; reference: examples/linear_regression/main.py:158-185) against fixture F14
(tests/golden/tangent_golden.npz: outputs of the reference's own class, tests/golden/make_golden_tangent.py).
CPU: the projections themselves.  GPU: the greedy constructions the reference builds on them."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(1, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
import bayesiancoresets_amd as bc  # noqa: E402
import model_linreg  # noqa: E402


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "tangent_golden.npz"))


def _projector(g):
    return model_linreg.tangent_space_projector(bc, g["bV"], g["mu0"], g["Sig0"], float(g["datastd"]) ** 2)


def test_F14_projections_match_the_reference_class(g):
    prj = _projector(g)
    assert isinstance(prj, bc.Projector)
    Z = g["Z"]
    v = prj.project(Z)                                   # constructed at the prior (main.py:172-174)
    assert v.shape == (Z.shape[0], Z.shape[1] - 1 + g["bV"].shape[1] ** 2)
    np.testing.assert_allclose(v, g["v_prior"], rtol=1e-12, atol=1e-13 * np.abs(g["v_prior"]).max())
    prj.update(g["w"], Z[g["idx"]])
    np.testing.assert_allclose(prj.project(Z), g["v_core"], rtol=1e-11, atol=1e-12 * np.abs(g["v_core"]).max())
    prj.update(np.array([]), np.zeros((0, Z.shape[1])))   # an empty coreset is the prior again
    np.testing.assert_allclose(prj.project(Z), g["v_prior"], rtol=1e-12, atol=1e-13 * np.abs(g["v_prior"]).max())
    with pytest.raises(NotImplementedError):
        prj.project(Z, grad=True)


def test_inner_products_are_the_covariances_of_the_log_likelihoods():
    """With the full basis (proj_dim = D) v_n . v_m = Cov_theta(ll_n, ll_m) under the weighted posterior -- what makes the
    projection 'exact': checked against the closed form of that covariance and against 400k Monte-Carlo draws."""
    rs = np.random.RandomState(2)
    N, D, sigsq = 9, 4, 0.6
    Z = np.hstack((rs.randn(N, D), rs.randn(N, 1)))
    mu0, Sig0 = 0.2 * rs.randn(D), np.diag(rs.uniform(0.5, 2.0, D))
    prj = model_linreg.tangent_space_projector(bc, np.eye(D), mu0, Sig0, sigsq)
    pts, w = Z[:3], np.array([2.0, 0.5, 1.5])
    prj.update(w, pts)
    V = prj.project(Z)
    K = V.dot(V.T)
    mu, U = model_linreg.weighted_posterior(mu0, np.linalg.inv(Sig0), sigsq, pts, w)
    beta, nu = Z[:, :-1].dot(U), Z[:, -1] - Z[:, :-1].dot(mu)
    B = beta.dot(beta.T)
    np.testing.assert_allclose(K, (np.outer(nu, nu) * B + 0.5 * B ** 2) / sigsq ** 2, rtol=1e-12)
    th = mu + rs.randn(400000, D).dot(U.T)
    ll = -(Z[:, -1][:, None] - Z[:, :-1].dot(th.T)) ** 2 / (2 * sigsq)
    np.testing.assert_allclose(np.cov(ll), K, rtol=0.05, atol=0.02 * np.abs(K).max())


@pytest.mark.gpu
def test_F14_giga_on_the_exact_vectors(g):
    """GIGA-OPT-EXACT (main.py:189-197): HilbertCoreset on the tangent space at the full-data posterior."""
    Z = g["Z"]
    prj = _projector(g)
    prj.update(np.ones(Z.shape[0]), Z)
    h = bc.HilbertCoreset(Z, prj)
    h.build(12)
    wts, pts, idcs = h.get()
    assert np.array_equal(idcs, g["giga_idcs"])
    np.testing.assert_allclose(wts, g["giga_wts"], rtol=1e-5)
    np.testing.assert_allclose(h.error(), float(g["giga_err"]), rtol=1e-6)
    assert np.array_equal(pts, Z[idcs])


@pytest.mark.gpu
def test_F14_sparsevi_on_the_exact_projector(g):
    """SVI-EXACT (main.py:191): the projector follows the weighted coreset posterior, no samples anywhere."""
    Z = g["Z"]
    np.random.seed(3)
    s = bc.SparseVICoreset(Z, _projector(g), opt_itrs=15, step_sched=lambda i: 1.0 / (1.0 + i))
    s.build(6)
    wts, pts, idcs = s.get()
    assert np.array_equal(idcs, g["svi_idcs"])
    np.testing.assert_allclose(wts, g["svi_wts"], rtol=1e-6, atol=1e-9 * np.abs(g["svi_wts"]).max())
