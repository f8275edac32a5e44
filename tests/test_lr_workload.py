"""Config-3 style workload: Laplace-projected logistic-regression vectors with row
norms from 1e-16 to 2, through BlackBoxProjector -> HilbertCoreset.  Golden outputs come from the
reference (tests/golden/make_golden_lr.py)."""
import os

import numpy as np
import pytest

from lr_workload import make_data, log_likelihood

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lr():
    g = np.load(os.path.join(ROOT, "tests", "golden", "lr_golden.npz"))
    Z = make_data(1, int(g["N"]), int(g["D"]))
    return g, Z


def _projector(bc, g):
    samples = g["samples"]
    return bc.BlackBoxProjector(lambda sz, w, p: samples, int(g["S"]), log_likelihood)


def test_projection_matches_reference(lr):
    import bayesiancoresets_amd as bc
    g, Z = lr
    vecs = _projector(bc, g).project(Z)
    np.testing.assert_allclose(vecs.sum(axis=0), g["vecs_sum"], rtol=1e-10)
    np.testing.assert_allclose(np.abs(vecs).sum(), float(g["vecs_abs_sum"]), rtol=1e-12)
    np.testing.assert_allclose(vecs[:4], g["vecs_head"], rtol=1e-12, atol=1e-300)
    norms = np.sqrt((vecs ** 2).sum(axis=1))
    assert norms.min() < 1e-15 and norms.max() > 1.0      # 16 decades of row norm


@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
def test_oracle_on_lr_vectors(lr, alg):
    import bayesiancoresets_amd as bc
    from oracle.snnls_oracle import SnnlsOracle, hilbert_readout
    g, Z = lr
    vecs = _projector(bc, g).project(Z)
    o = SnnlsOracle(vecs.T, vecs.sum(axis=0), alg=alg)
    o.build(int(g["itrs"]))
    assert np.array_equal(np.array([t[0] for t in o.trace]), g[alg + "_sel"])
    w, idx = hilbert_readout(o.weights())
    assert np.array_equal(idx, g[alg + "_idcs"])
    np.testing.assert_allclose(w, g[alg + "_wts"], rtol=1e-9)
    np.testing.assert_allclose(o.error(), float(g[alg + "_err"]), rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
@pytest.mark.parametrize("dtype", ("float32", "float64", "float16"))
def test_gpu_on_lr_vectors(lr, alg, dtype):
    """fp32 storage of the NORMALISED rows + fp64 norms keeps 16 decades of dynamic range exact enough
    for bit-exact selections and 1e-5 weights."""
    import bayesiancoresets_amd as bc
    g, Z = lr
    cls = {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe, "omp": bc.snnls.OrthoPursuit}[alg]

    class Solver(cls):
        def __init__(self, A, b):
            super().__init__(A, b, dtype=dtype)

    c = bc.HilbertCoreset(Z, _projector(bc, g), snnls=Solver)
    c.build(int(g["itrs"]))
    wts, pts, idcs = c.get()
    sel = c.snnls.last_trace[0]
    assert np.array_equal(sel, g[alg + "_sel"])
    assert np.array_equal(idcs, g[alg + "_idcs"])
    np.testing.assert_allclose(wts, g[alg + "_wts"], rtol=1e-5)
    np.testing.assert_allclose(c.error(), float(g[alg + "_err"]), rtol=1e-7)
    assert np.array_equal(pts, Z[idcs])
