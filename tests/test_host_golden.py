"""CPU only: host-side pieces against fixtures produced by the reference itself (tests/golden/make_golden_host.py):
F10 monotone check switched off (oracle), F11 sampling baselines (product host code), F12 the test-side restatements
of the example likelihoods (tests/models.py, tests/lr_workload.py), F13 BlackBoxProjector centring."""
import logging
import os

import numpy as np
import pytest

import bayesiancoresets_amd as bc
from oracle.snnls_oracle import SnnlsOracle, hilbert_readout
from models import (logistic_log_likelihood, poisson_log_likelihood, linreg_log_likelihood, linreg_weighted_post)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hg():
    return np.load(os.path.join(ROOT, "tests", "golden", "host_golden.npz"))


@pytest.mark.parametrize("alg", ("giga", "fw"))
def test_F10_oracle_without_monotone_check(hg, normal_inputs, alg):
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="faithful", check_error_monotone=False)
    o.build(int(hg["F10_%s_itrs" % alg]))
    sel = np.array([t[0] for t in o.trace if t[0] >= 0])
    assert np.array_equal(sel, hg["F10_%s_sel" % alg])
    assert o.reached_numeric_limit == bool(hg["F10_%s_limit" % alg]) and o.size() == int(hg["F10_%s_size" % alg])
    w, idx = hilbert_readout(o.weights())
    assert np.array_equal(idx, hg["F10_%s_idx" % alg]) and np.array_equal(w, hg["F10_%s_w" % alg])
    assert o.error() == float(hg["F10_%s_final_err" % alg])


@pytest.mark.parametrize("name", ("unif", "imp"))
def test_F11_sampling_baselines_match_reference(hg, name):
    Xs = np.random.RandomState(0).randn(50, 4)
    cls = {"unif": bc.snnls.UniformSampling, "imp": bc.snnls.ImportanceSampling}[name]
    np.random.seed(3)
    s = cls(Xs.T, Xs.sum(axis=0))
    assert isinstance(s, bc.snnls.SparseNNLS) and s.check_error_monotone is False      # sampling.py:6,16
    k = "F11_%s_" % name
    np.testing.assert_array_equal(s.ps, hg[k + "ps"])
    s.build(30)
    np.testing.assert_array_equal(s.weights(), hg[k + "w30"])
    assert s.error() == float(hg[k + "err30"]) and int(s.size()) == int(hg[k + "size30"])
    s.build(25)
    np.testing.assert_array_equal(s.weights(), hg[k + "w55"])
    s.optimize()
    np.testing.assert_allclose(s.weights(), hg[k + "wopt"], rtol=1e-12, atol=1e-14)
    assert s.reached_numeric_limit == bool(hg[k + "limit_after_opt"])
    s.reset()
    assert s.weights().sum() == float(hg[k + "w_reset_sum"]) and not s.reached_numeric_limit and s.cts.sum() == 0


def test_F11_uniform_sampling_coreset(hg):
    Xs = np.random.RandomState(0).randn(50, 4)
    np.random.seed(4)
    c = bc.UniformSamplingCoreset(Xs)
    c.build(20)
    c.build(7)
    wts, pts, idcs = c.get()
    np.testing.assert_array_equal(wts, hg["F11_usc_wts"])
    np.testing.assert_array_equal(idcs, hg["F11_usc_idcs"])
    np.testing.assert_array_equal(pts, Xs[idcs])


def test_sampling_latch_and_no_data_early_outs(caplog):
    """snnls.py:32-38 for the host-side solvers: a latched solver and an empty matrix return with a WARNING."""
    Xs = np.random.RandomState(0).randn(20, 3)
    s = bc.snnls.UniformSampling(Xs.T, Xs.sum(axis=0))
    np.random.seed(0)
    s.build(5)
    w5 = s.weights()
    s.reached_numeric_limit = True
    with caplog.at_level(logging.WARNING):
        s.build(5)
    np.testing.assert_array_equal(s.weights(), w5)
    assert any("already reached" in r.getMessage() for r in caplog.records)
    caplog.clear()
    e = bc.snnls.UniformSampling(np.zeros((3, 0)), np.zeros(3))
    with caplog.at_level(logging.WARNING):
        e.build(5)
    assert any("no data" in r.getMessage() for r in caplog.records)


def test_host_state_machine_retry_then_latch(caplog):
    """A user subclass of bc.snnls.SparseNNLS whose steps raise NumericalPrecisionError: first failure -> 'Stabilizing
    and retrying', _stabilize() called; a checked success clears the strike; two failures in a row latch
    (snnls.py:56-74)."""
    from bayesiancoresets_amd.util.errors import NumericalPrecisionError
    A = np.eye(4)

    class Scripted(bc.snnls.SparseNNLS):
        script = []
        stabilized = 0

        def _select(self):
            return 0

        def _reweight(self, f):
            act = self.script.pop(0)
            if act == "raise":
                raise NumericalPrecisionError("scripted")
            self.w[f] += act

        def _stabilize(self):
            self.stabilized += 1

    s = Scripted(A, np.array([10.0, 0, 0, 0]))
    s.script = [1.0, "raise", 1.0, "raise", 20.0, 1.0]      # ok, fail, ok (checked: clears), fail, worse error (reverted), unused
    with caplog.at_level(logging.WARNING):
        s.build(6)
    assert s.reached_numeric_limit and s.stabilized == 2 and s.w[0] == 2.0 and s.script == [1.0]
    msgs = [r.getMessage() for r in caplog.records]
    assert sum("Stabilizing and retrying" in m for m in msgs) == 2 and sum("second time" in m for m in msgs) == 1
    assert any("Error not monotone" in m for m in msgs) and "No more points will be added" in msgs[-1]


def test_F12_example_likelihood_restatements(hg):
    np.testing.assert_allclose(logistic_log_likelihood(hg["F12_lr_Z"], hg["F12_lr_th"]), hg["F12_lr_ll"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(poisson_log_likelihood(hg["F12_poiss_Z"].copy(), hg["F12_poiss_th"]), hg["F12_poiss_ll"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(linreg_log_likelihood(hg["F12_linreg_Z"], hg["F12_linreg_th"], float(hg["F12_linreg_sigsq"])),
                               hg["F12_linreg_ll"], rtol=1e-14, atol=0)
    mu, U = linreg_weighted_post(hg["F12_post_mu0"], hg["F12_post_Sig0inv"], 0.37, hg["F12_linreg_Z"], hg["F12_post_w"])
    np.testing.assert_allclose(mu, hg["F12_post_mu"], rtol=1e-12)
    np.testing.assert_allclose(U.dot(U.T), hg["F12_post_Sigma"], rtol=1e-12, atol=1e-15)
    mu, U = linreg_weighted_post(hg["F12_post_mu0"], hg["F12_post_Sig0inv"], 0.37, np.zeros((0, 6)), np.zeros(0))
    np.testing.assert_allclose(mu, hg["F12_post_empty_mu"], rtol=1e-13)
    np.testing.assert_allclose(U.dot(U.T), hg["F12_post_empty_Sigma"], rtol=1e-12, atol=1e-15)
    # the product's own example model (examples/common/model_linreg.py) against the same fixture
    import sys
    sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
    import model_linreg as ours
    mu, U = ours.weighted_posterior(hg["F12_post_mu0"], hg["F12_post_Sig0inv"], 0.37, hg["F12_linreg_Z"], hg["F12_post_w"])
    np.testing.assert_allclose(mu, hg["F12_post_mu"], rtol=1e-12)
    np.testing.assert_allclose(U.dot(U.T), hg["F12_post_Sigma"], rtol=1e-12, atol=1e-15)


def test_F13_blackbox_projector_centring(hg):
    th = hg["F12_lr_th"]
    prj = bc.BlackBoxProjector(lambda n, w, p: th[:n], 9, logistic_log_likelihood)
    np.testing.assert_allclose(prj.project(hg["F12_lr_Z"]), hg["F13_vecs"], rtol=1e-13, atol=1e-13)


def test_laplace_fit_of_the_package_example_matches_reference_fit():
    """examples/common/model_lr.py (Newton) lands on the MAP / covariance the reference's BFGS + Hessian gave
    (tests/golden/make_golden_lr.py ran simple_lr/main.py:57-63 with the reference's model_lr)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
    import model_lr
    from lr_workload import make_data
    g = np.load(os.path.join(ROOT, "tests", "golden", "lr_golden.npz"))
    Z = make_data(1, int(g["N"]), int(g["D"]))
    mu, cov = model_lr.laplace_fit(Z)
    np.testing.assert_allclose(mu, g["mu"], atol=1e-6)          # BFGS stops at gtol 1e-5; Newton converges fully
    np.testing.assert_allclose(cov, g["cov"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(model_lr.log_likelihood(Z[:50], g["samples"][:7]), logistic_log_likelihood(Z[:50], g["samples"][:7]), rtol=1e-14)


def test_rbf_workload_generator_is_pinned():
    import hashlib
    import importlib.util
    spec = importlib.util.spec_from_file_location("rbf_workload", os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common", "rbf_workload.py"))
    rbf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rbf)
    g = np.load(os.path.join(ROOT, "tests", "golden", "rbf_golden.npz"))
    wl = rbf.make_rbf_regression(int(g["N"]), int(g["nb"]), seed=1)
    assert hashlib.sha256(wl["Z"].tobytes()).hexdigest() == str(g["Z_sha"])
    assert wl["Z"].shape == (int(g["N"]), 302) and abs(wl["sigsq"] - float(g["sigsq"])) == 0.0
    # collinear by construction: the wide bases are nearly constant on the unit square
    X = wl["Z"][:2000, :301]
    assert np.linalg.cond(X) > 1e8
