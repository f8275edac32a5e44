"""The models of the logistic / Poisson regression experiment (examples/common/model_poiss.py, model_lr.py) against fixture
F15 (tests/golden/poiss_golden.npz: the reference's `get_laplace` with its own models, and its Poisson log-likelihood on
both branches of compute_s; tests/golden/make_golden_poiss.py).  CPU: models and Laplace fits.  GPU: the harness end to end."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(1, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
import model_lr  # noqa: E402
import model_poiss  # noqa: E402


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(ROOT, "tests", "golden", "poiss_golden.npz"))


def test_F15_poisson_log_likelihood_on_both_branches(g):
    with np.errstate(over="ignore"):
        got = model_poiss.log_likelihood(g["ll_z"], g["ll_th"])
    np.testing.assert_allclose(got, g["ll"], rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("tag", ("poiss_full", "poiss_wtd"))
def test_F15_poisson_laplace_fit(g, tag):
    """Newton's maximiser is BFGS's (model_poiss.py:44-45 log joint; main.py:15-41): mean to the optimiser's tolerance of the
    reference, covariance = inverse negative Hessian there."""
    w = None if tag == "poiss_full" else g["poiss_w"]
    mu, cov = model_poiss.laplace_fit(g["poiss_Z"], w)
    np.testing.assert_allclose(mu, g[tag + "_mu"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(cov, g[tag + "_cov"], rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("tag", ("lr_full", "lr_wtd"))
def test_F15_logistic_laplace_fit(g, tag):
    w = None if tag == "lr_full" else g["poiss_w"]
    mu, cov = model_lr.laplace_fit(g["lr_Z"], w)
    np.testing.assert_allclose(mu, g[tag + "_mu"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(cov, g[tag + "_cov"], rtol=2e-4, atol=1e-7)


def test_poisson_derivatives_against_finite_differences():
    rs = np.random.RandomState(4)
    s = np.hstack((rs.uniform(-30, 30, 40), [-150.0, -101.0, -99.0, 0.0, 90.0]))
    y = rs.poisson(2.0, size=s.shape).astype(float)
    f = lambda t: y * model_poiss._rate_and_log(t)[1] - model_poiss._rate_and_log(t)[0]
    gd, hd = model_poiss._derivs(s, y)
    h = 1e-5
    np.testing.assert_allclose(gd, (f(s + h) - f(s - h)) / (2 * h), rtol=1e-6, atol=1e-7)
    g1 = lambda t: model_poiss._derivs(t, y)[0]
    np.testing.assert_allclose(hd, (g1(s + h) - g1(s - h)) / (2 * h), rtol=1e-5, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("model,alg", (("lr", "GIGA-OPT"), ("lr", "SVI"), ("poiss", "GIGA-OPT"), ("poiss", "GIGA-REAL"), ("poiss", "SVI"), ("poiss", "US")))
def test_logistic_poisson_regression_example_cli(tmp_path, model, alg):
    """examples/logistic_poisson_regression/main.py (the harness BASELINE configs[2] names; reference main.py:58-230 without its
    MCMC evaluation): runs end to end on the device projectors ("logistic" / "poisson") and stores the reference's metric
    columns; a growing coreset brings the Laplace posterior of the coreset towards that of the full data."""
    import subprocess
    import pandas as pd
    script = os.path.join(ROOT, "bayesian-coresets_amd", "examples", "logistic_poisson_regression", "main.py")
    folder = str(tmp_path / "results") + "/"
    cmd = [sys.executable, script, "--model", model, "--dataset", "synth_" + model, "--alg", alg, "--trial", "1", "--data_num", "5000",
           "--data_dim", "4", "--proj_dim", "64", "--coreset_size_max", "40", "--coreset_num_sizes", "4", "--opt_itrs", "15",
           "--results_folder", folder, "run"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    files = [f for f in os.listdir(folder) if f != "manifest.csv"]
    assert len(files) == 1
    t = pd.read_csv(os.path.join(folder, files[0]))
    for col in ("csizes", "Ms", "cputs", "rklw", "fklw", "mu_errs", "Sig_errs"):
        assert col in t.columns, col
    assert np.isfinite(t["rklw"]).all() and np.isfinite(t["fklw"]).all()
    assert t["csizes"].iloc[-1] >= 1
    if alg in ("GIGA-OPT", "SVI"):
        assert t["fklw"].iloc[-1] < t["fklw"].iloc[0]
    again = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert again.returncode == 0 and "Results already exist" in again.stdout
