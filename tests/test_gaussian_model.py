"""The Gaussian-mean experiment (examples/common/model_gaussian.py, examples/gaussian/main.py) against fixture F16
(tests/golden/gaussian_golden.npz: the reference's model_gaussian and its exact `GaussianProjector`, plus the reference's
HilbertCoreset / SparseVICoreset runs on it; tests/golden/make_golden_gaussian.py).  CPU: model and projector.  GPU: the greedy
constructions and the harness end to end."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(1, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
import bayesiancoresets_amd as bc  # noqa: E402
import model_gaussian  # noqa: E402


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "gaussian_golden.npz"))


def _projector(g):
    return model_gaussian.tangent_space_projector(bc, g["mu0"], g["Sig0inv"], np.linalg.inv(g["Sig"]))


def test_F16_model_matches_reference(g):
    Siginv = np.linalg.inv(g["Sig"])
    ll = model_gaussian.log_likelihood(g["x"], g["th"], Siginv, np.linalg.slogdet(g["Sig"])[1])
    np.testing.assert_allclose(ll, g["ll"], rtol=1e-12)
    mu, U = model_gaussian.weighted_posterior(g["mu0"], g["Sig0inv"], Siginv, g["x"][g["idx"]], g["w"])
    np.testing.assert_allclose(mu, g["post_mu"], rtol=1e-12)
    np.testing.assert_allclose(U.dot(U.T), g["post_cov"], rtol=1e-12)


def test_F16_exact_projector_matches_the_reference_class(g):
    prj = _projector(g)
    x = g["x"]
    np.testing.assert_allclose(prj.project(x), g["v_prior"], rtol=1e-11, atol=1e-12 * np.abs(g["v_prior"]).max())
    prj.update(g["w"], x[g["idx"]])
    np.testing.assert_allclose(prj.project(x), g["v_core"], rtol=1e-11, atol=1e-12 * np.abs(g["v_core"]).max())
    with pytest.raises(NotImplementedError):
        prj.project(x, grad=True)


@pytest.mark.gpu
def test_F16_giga_and_sparsevi_on_the_exact_vectors(g):
    x = g["x"]
    prj = _projector(g)
    prj.update(np.ones(x.shape[0]), x)
    h = bc.HilbertCoreset(x, prj)
    h.build(10)
    wts, pts, idcs = h.get()
    assert np.array_equal(idcs, g["giga_idcs"])
    np.testing.assert_allclose(wts, g["giga_wts"], rtol=1e-5)
    np.testing.assert_allclose(h.error(), float(g["giga_err"]), rtol=1e-6, atol=1e-9)
    np.random.seed(5)
    s = bc.SparseVICoreset(x, _projector(g), opt_itrs=12, step_sched=lambda i: 1.0 / (1.0 + i))
    s.build(5)
    wts, pts, idcs = s.get()
    assert np.array_equal(idcs, g["svi_idcs"])
    np.testing.assert_allclose(wts, g["svi_wts"], rtol=1e-6, atol=1e-9 * np.abs(g["svi_wts"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("alg", ("SVI", "SVI-EXACT", "GIGA-OPT", "GIGA-OPT-EXACT", "GIGA-REAL-EXACT", "US"))
def test_gaussian_example_cli(tmp_path, alg):
    """examples/gaussian/main.py (reference main.py:28-222): end to end, the reference's result columns."""
    import subprocess
    import pandas as pd
    script = os.path.join(ROOT, "bayesian-coresets_amd", "examples", "gaussian", "main.py")
    folder = str(tmp_path / "results") + "/"
    cmd = [sys.executable, script, "--alg", alg, "--trial", "1", "--data_num", "1000", "--data_dim", "20", "--proj_dim", "60",
           "--coreset_size_max", "30", "--coreset_num_sizes", "4", "--opt_itrs", "15", "--results_folder", folder, "run"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    files = [f for f in os.listdir(folder) if f != "manifest.csv"]
    assert len(files) == 1
    t = pd.read_csv(os.path.join(folder, files[0]))
    for col in ("csizes", "Ms", "cputs", "rklw", "fklw", "mu_errs", "Sig_errs"):
        assert col in t.columns, col
    assert t["Ms"].iloc[0] == 0 and t["csizes"].iloc[0] == 0
    assert np.isfinite(t["rklw"]).all() and np.isfinite(t["fklw"]).all()
    assert t["csizes"].iloc[-1] >= 1
    if alg != "US":
        assert t["fklw"].iloc[-1] < t["fklw"].iloc[0]
