"""CPU-only tests of the host-side mirror of the reference interface (no device arithmetic):
Projector / BlackBoxProjector (projector.py:4-32), Coreset guards (coreset.py:22-64), sampling
baselines, util (TOL, logging, nn_opt)."""
import logging

import numpy as np
import pytest

import bayesiancoresets_amd as bc


def test_namespace_matches_reference_surface():
    for name in ("HilbertCoreset", "SparseVICoreset", "UniformSamplingCoreset", "BlackBoxProjector", "Projector", "Coreset"):
        assert hasattr(bc, name)
    for name in ("GIGA", "FrankWolfe", "OrthoPursuit", "ImportanceSampling", "UniformSampling", "SparseNNLS"):
        assert hasattr(bc.snnls, name)
    for name in ("nn_opt", "set_verbosity", "set_tolerance", "TOL"):
        assert hasattr(bc.util, name)
    assert issubclass(bc.util.errors.NumericalPrecisionError, Exception)
    # the one exported name that is not on the accelerated path says so when constructed (bayesiancoresets/__init__.py:1)
    with pytest.raises(NotImplementedError):
        bc.BatchPSVICoreset(np.zeros((3, 2)), None, 10)


def test_tolerance_global():
    old = bc.util.TOL
    bc.util.set_tolerance(1e-9)
    assert bc.util.TOL == 1e-9
    bc.util.set_tolerance(old)
    assert bc.util.TOL == old == 1e-12


def test_blackbox_projector_centres_rows_and_handles_empty_update():
    calls = []

    def sampler(n, wts, pts):
        calls.append((n, np.asarray(wts).shape, np.asarray(pts).shape))
        return np.random.RandomState(0).randn(n, 3)

    def loglik(pts, samples):
        return pts.dot(samples.T)

    p = bc.BlackBoxProjector(sampler, 7, loglik)
    assert calls[0] == (7, (0,), (0,))                # projector.py:17: update(np.array([]), np.array([]))
    pts = np.random.RandomState(1).randn(11, 3)
    v = p.project(pts)
    assert v.shape == (11, 7)
    np.testing.assert_allclose(v.mean(axis=1), 0.0, atol=1e-14)   # projector.py:21, no 1/sqrt(S) scaling
    np.testing.assert_allclose(v, loglik(pts, p.samples) - loglik(pts, p.samples).mean(axis=1)[:, None])
    with pytest.raises(ValueError):
        p.project(pts, grad=True)
    with pytest.raises(NotImplementedError):
        bc.Projector().project(pts)


def test_blackbox_projector_gradients():
    def sampler(n, wts, pts):
        return np.random.RandomState(0).randn(n, 2)

    def loglik(pts, samples):
        return pts.dot(samples.T)

    def grad(pts, samples):
        return np.repeat(samples.T[None, :, :], pts.shape[0], axis=0)     # N x D x S

    p = bc.BlackBoxProjector(sampler, 5, loglik, grad)
    lls, g = p.project(np.ones((4, 2)), grad=True)
    assert lls.shape == (4, 5) and g.shape == (4, 2, 5)
    np.testing.assert_allclose(g.mean(axis=2), 0.0, atol=1e-14)


def test_coreset_base_guards():
    class Fake(bc.Coreset):
        def __init__(self):
            super().__init__()
            self.calls = 0
            self._err = 1.0

        def _build(self, itrs):
            self.calls += 1
            self.wts = np.array([0.5, 0.0, 2.0])
            self.idcs = np.array([3, 5, 9])
            self.pts = np.arange(6.0).reshape(3, 2)

        def error(self):
            return self._err

        def _optimize(self):
            self._err = 2.0   # worse

    c = Fake()
    w, p, i = c.get()
    assert w.shape == (0,) and p.shape == (0,) and i.shape == (0,)      # coreset.py:26-27
    c.build(0)
    c.build(-3)
    assert c.calls == 0                                                  # coreset.py:37-38
    c.build(2)
    w, p, i = c.get()
    assert list(w) == [0.5, 2.0] and list(i) == [3, 9] and p.shape == (2, 2)   # only wts > 0
    assert c.size() == 2
    c.optimize()                                                         # error grew -> revert + latch
    assert c.reached_numeric_limit and list(c.wts) == [0.5, 0.0, 2.0]
    c.build(5)
    assert c.calls == 1                                                  # latched: coreset.py:34-35
    c.reset()
    assert c.size() == 0 and not c.reached_numeric_limit


def test_hilbert_kwargs_rejected_like_reference():
    class P(bc.Projector):
        def project(self, pts, grad=False):
            return pts
    # **kw is forwarded to Coreset.__init__ which accepts none (hilbert.py:27 / coreset.py:8); with no GPU the
    # solver constructor fails first, so only assert that SOME exception is raised either way
    with pytest.raises(Exception):
        bc.HilbertCoreset(np.ones((4, 2)), P(), bogus=1)


def test_sampling_baselines_host_numpy():
    rs = np.random.RandomState(0)
    X = rs.randn(50, 4)
    np.random.seed(3)
    s = bc.snnls.UniformSampling(X.T, X.sum(axis=0))
    s.build(30)
    w = s.weights()
    assert w.shape == (50,) and np.all(w >= 0) and s.size() > 0
    np.testing.assert_allclose(w.sum(), 50.0)              # counts/(n*p) with p = 1/N sums to N
    e0 = s.error()
    s.optimize()
    assert s.error() <= e0 * (1 + 1e-12)
    imp = bc.snnls.ImportanceSampling(X.T, X.sum(axis=0))
    np.testing.assert_allclose(imp.ps.sum(), 1.0)
    np.random.seed(4)
    c = bc.UniformSamplingCoreset(X)
    c.build(20)
    wts, pts, idcs = c.get()
    np.testing.assert_allclose(wts.sum(), 50.0)
    assert np.array_equal(pts, X[idcs])


def test_nn_opt_projects_and_converges():
    target = np.array([1.0, -2.0, 3.0])
    x = bc.util.nn_opt(np.zeros(3), lambda x: x - target, opt_itrs=3000, step_sched=lambda i: 0.05)
    np.testing.assert_allclose(x, [1.0, 0.0, 3.0], atol=2e-2)
    y = bc.util.nn_opt(np.zeros(3), lambda x: x - target, nn_idcs=np.array([0, 2]), opt_itrs=3000,
                       step_sched=lambda i: 0.05)
    np.testing.assert_allclose(y, [1.0, -2.0, 3.0], atol=2e-2)


def test_logging_format_and_verbosity(capsys):
    bc.util.set_verbosity("warning")
    assert logging.getLogger().level == logging.WARNING
    c = bc.Coreset()
    assert c.alg_name.startswith("Coreset-") and len(c.alg_name.split("-")[1]) == 6
    c.log.warning("hello")
    bc.util.set_verbosity("error")
    assert logging.getLogger().level == logging.ERROR
