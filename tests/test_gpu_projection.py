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
