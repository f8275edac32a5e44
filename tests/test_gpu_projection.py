"""GPU: device-native projection (csrc/proj.hip) against NumPy restatements of the reference's example
likelihoods, and SparseVICoreset (reference: coreset/sparsevi.py) against golden vectors produced by the
reference itself (tests/golden/make_golden_svi.py)."""
import os

import numpy as np
import pytest

from models import (logistic_log_likelihood, poisson_log_likelihood, linreg_log_likelihood, make_linreg_data,
                    make_poisson_data, linreg_sampler)
from lr_workload import make_data as make_lr_data

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bc():
    import bayesiancoresets_amd as bc
    return bc


def _cases():
    rs = np.random.RandomState(5)
    return {
        "logistic": (make_lr_data(3, 5000, 10), rs.randn(37, 10), lambda z, th: logistic_log_likelihood(z, th), 1.0),
        "poisson": (make_poisson_data(4, 4111, 6), 0.3 * rs.randn(64, 6), lambda z, th: poisson_log_likelihood(z, th), 1.0),
        "linreg": (make_linreg_data(5, 6007, 30), rs.randn(130, 30), lambda z, th: linreg_log_likelihood(z, th, 0.7), 0.7),
    }


@pytest.mark.parametrize("family", ("logistic", "poisson", "linreg"))
def test_project_matches_numpy(bc, family):
    Z, theta, ll, sigsq = _cases()[family]
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, theta.shape[0], sigsq=sigsq)
    ref = bc.BlackBoxProjector(lambda n, w, p: theta, theta.shape[0], ll)
    want = ref.project(Z)
    got = prj.project(Z).cpu().numpy()
    assert got.shape == want.shape
    scale = np.abs(want).max()
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12 * scale)
    # fused consumers: column sums and correlation arg-max without materialising N x S
    np.testing.assert_allclose(prj.project_colsum(Z), want.sum(axis=0), rtol=1e-9, atol=1e-9 * np.abs(want).sum() / want.shape[1])
    resid = np.random.RandomState(9).randn(theta.shape[0])
    corrs = want.dot(resid) / np.sqrt((want ** 2).sum(axis=1)) / want.shape[1]
    best, row = prj.project_select(Z, resid)
    assert row == int(np.argmax(corrs))
    np.testing.assert_allclose(best, corrs.max(), rtol=1e-7)
    # small inputs (the coreset points) take the same kernel
    np.testing.assert_allclose(prj.project(Z[:3]).cpu().numpy(), ref.project(Z[:3]), rtol=1e-11, atol=1e-12 * scale)


def test_hilbert_coreset_with_device_projector(bc):
    """Config-3 end to end on the device: logistic projection -> normalised rows -> OMP / GIGA, vs the
    same pipeline with the host BlackBoxProjector (identical selections, weights to 1e-5)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "lr_golden.npz"))
    Z = make_lr_data(1, int(g["N"]), int(g["D"]))
    samples = g["samples"]
    dev = bc.DeviceProjector("logistic", lambda n, w, p: samples, int(g["S"]))
    for alg, cls in (("giga", bc.snnls.GIGA), ("omp", bc.snnls.OrthoPursuit), ("fw", bc.snnls.FrankWolfe)):
        c = bc.HilbertCoreset(Z, dev, snnls=cls)
        c.build(int(g["itrs"]))
        wts, pts, idcs = c.get()
        assert np.array_equal(c.snnls.last_trace[0], g[alg + "_sel"])
        assert np.array_equal(idcs, g[alg + "_idcs"])
        np.testing.assert_allclose(wts, g[alg + "_wts"], rtol=1e-5)
        np.testing.assert_allclose(c.error(), float(g[alg + "_err"]), rtol=1e-7)


@pytest.mark.parametrize("kind", ("device", "blackbox"))
def test_sparsevi_matches_reference(bc, kind):
    """F6: 5 greedy steps x 20 ADAM steps; same points in the same order, weights to 1e-5."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "svi_golden.npz"))
    N, D, S, sigsq = int(g["N"]), int(g["D"]), int(g["S"]), float(g["sigsq"])
    Z = make_linreg_data(1, N, D)
    sampler = linreg_sampler(np.zeros(D), np.eye(D), sigsq)
    np.random.seed(2)
    if kind == "device":
        prj = bc.DeviceProjector("linreg", sampler, S, sigsq=sigsq)
    else:
        prj = bc.BlackBoxProjector(sampler, S, lambda z, th: linreg_log_likelihood(z, th, sigsq))
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=int(g["opt_itrs"]))
    for i in range(int(g["steps"])):
        alg.build(1)
        assert np.array_equal(alg.idcs, g["step%d_idcs" % i]), "step %d picked a different point" % i
        np.testing.assert_allclose(alg.wts, g["step%d_wts" % i], rtol=1e-5, atol=1e-8)
    wts, pts, idcs = alg.get()
    assert np.array_equal(idcs, g["get_idcs"])
    np.testing.assert_allclose(wts, g["get_wts"], rtol=1e-5)
    assert np.array_equal(pts, Z[idcs])


def test_device_sampler_feeds_projector_in_place(bc):
    """A sampler may hand back a GPU tensor (examples/common/model_linreg.py): same posterior as the NumPy form,
    and DeviceProjector uses the tensor without a host copy -- the fused consumers give the values they give for
    the same samples passed as an ndarray."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("model_linreg", os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd", "examples", "common", "model_linreg.py"))
    ml = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ml)
    rs = np.random.RandomState(3)
    D, S, N = 24, 64, 5000
    mu0, Sig0, sigsq = rs.randn(D), np.eye(D) * 2.0, 0.7
    pts = rs.randn(6, D + 1)
    wts = rs.rand(6) * 5
    dev_sampler = ml.posterior_sampler(mu0, Sig0, sigsq, device="cuda", seed=5)
    mu_d, U_d = dev_sampler.posterior(wts, pts)
    mu_h, U_h = ml.weighted_posterior(mu0, np.linalg.inv(Sig0), sigsq, pts, wts)
    np.testing.assert_allclose(mu_d.cpu().numpy(), mu_h, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose((U_d @ U_d.T).cpu().numpy(), U_h.dot(U_h.T), rtol=1e-10, atol=1e-12)
    prior = dev_sampler.posterior(None, None)[0].cpu().numpy()
    np.testing.assert_allclose(prior, mu0, rtol=1e-10, atol=1e-12)
    Z = rs.randn(N, D + 1)
    prj = bc.DeviceProjector("linreg", dev_sampler, S, sigsq=sigsq)
    prj.update(wts, pts)
    assert isinstance(prj.samples, torch.Tensor) and prj.samples.is_cuda and prj.samples.shape == (S, D)
    theta = prj.samples.cpu().numpy()
    host = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=sigsq)
    assert np.array_equal(prj.project_colsum(Z), host.project_colsum(Z))
    resid = rs.randn(S)
    assert prj.project_select(Z, resid) == host.project_select(Z, resid)
    # draws have the posterior's first two moments
    big = dev_sampler(20000, wts, pts).cpu().numpy()
    np.testing.assert_allclose(big.mean(axis=0), mu_h, atol=6 * np.sqrt(np.diag(U_h.dot(U_h.T)).max() / 20000))


@pytest.mark.parametrize("family", ("logistic", "poisson"))
def test_project_extreme_arguments(bc, family):
    """The piecewise branches of the example likelihoods (model_lr.py:29-31: linear tail for -m >= 100;
    model_poiss.py:25-30: softplus passthrough below -100): features scaled so that z.theta spans +-400."""
    rs = np.random.RandomState(17)
    D, S, N = 6, 48, 3001
    if family == "logistic":
        Z = rs.randn(N, D) * rs.choice([0.1, 5.0, 60.0], size=(N, 1))
        ll = logistic_log_likelihood
    else:
        X = rs.randn(N, D) * rs.choice([0.1, 5.0, 60.0], size=(N, 1))
        Z = np.hstack((X, rs.poisson(2.0, size=(N, 1)).astype(np.float64)))
        ll = poisson_log_likelihood
    theta = rs.randn(S, D)
    m = Z[:, :D].dot(theta.T)
    assert m.min() < -150 and m.max() > 150
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S)
    ref = bc.BlackBoxProjector(lambda n, w, p: theta, S, ll)
    want = ref.project(Z)
    got = prj.project(Z).cpu().numpy()
    assert np.isfinite(want).all() and np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 * np.abs(want).max())
