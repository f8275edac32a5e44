"""CPU-only: the SparseVICoreset host loop (select / ADAM optimise / sampler call order, reference
coreset/sparsevi.py:16-76) against the reference's golden run F6, with the two fused device consumers
replaced by NumPy stand-ins (test infrastructure; the product's DeviceProjector needs a GPU)."""
import os

import numpy as np

import bayesiancoresets_amd as bc
from bayesiancoresets_amd.projector import DeviceProjector
from models import linreg_log_likelihood, linreg_sampler, make_linreg_data

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyFusedProjector(DeviceProjector):
    """Same interface as DeviceProjector (project / project_colsum / project_select), NumPy inside."""

    def __init__(self, sampler, S, sigsq):
        self.sampler, self.projection_dimension, self.sigsq = sampler, S, sigsq
        self.update(np.array([]), np.array([]))

    def update(self, wts, pts):
        self.samples = np.atleast_2d(self.sampler(self.projection_dimension, wts, pts))

    def _vecs(self, pts):
        ll = linreg_log_likelihood(pts, self.samples, self.sigsq)
        return ll - ll.mean(axis=1)[:, None]

    def project(self, pts, grad=False):
        return self._vecs(pts)

    def project_colsum(self, pts):
        return self._vecs(pts).sum(axis=0)

    def _dev(self, pts):
        return pts

    def colsum_and_core(self, pts, core, persistent=True):
        S = self.samples.shape[0]
        return self.project_colsum(pts), (self._vecs(core) if core is not None and len(core) else np.zeros((0, S)))

    def project_select(self, pts, resid, row_ids=None):
        v = self._vecs(pts)
        corrs = v.dot(resid) / np.sqrt((v ** 2).sum(axis=1)) / v.shape[1]
        i = int(np.argmax(corrs))
        return float(corrs[i]), i


def test_sparsevi_host_loop_matches_reference():
    g = np.load(os.path.join(ROOT, "tests", "golden", "svi_golden.npz"))
    N, D, S, sigsq = int(g["N"]), int(g["D"]), int(g["S"]), float(g["sigsq"])
    Z = make_linreg_data(1, N, D)
    np.random.seed(2)
    prj = NumpyFusedProjector(linreg_sampler(np.zeros(D), np.eye(D), sigsq), S, sigsq)
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=int(g["opt_itrs"]))
    for i in range(3):
        alg.build(1)
        assert np.array_equal(alg.idcs, g["step%d_idcs" % i])
        np.testing.assert_allclose(alg.wts, g["step%d_wts" % i], rtol=1e-7, atol=1e-10)
    wts, pts, idcs = alg.get()
    assert np.array_equal(pts, Z[idcs]) and alg.error() == 0.0
    alg.reset()
    assert alg.size() == 0 and alg.pts.shape == (0, D + 1)


def test_enqueue_plan_is_offered_only_where_the_loop_needs_no_host():
    """``SparseVICoreset._enqueue_plan`` (the switch between the device-resident ADAM loop and the host loop, csrc/svi.hip):
    asked of the sampler only with a device projector, the full data set at every step, a non-empty coreset of at most 4096
    points and opt_itrs > 0; a sampler without ``enqueue_plan`` or one that declines (None) keeps the host loop."""
    Z = make_linreg_data(3, 500, 4)
    calls = []

    class Sampler(object):
        def __init__(self, answer):
            self.answer = answer

        def __call__(self, n, wts, pts):
            return np.zeros((n, 4))

        def enqueue_plan(self, n, pts, steps):
            calls.append((n, np.asarray(pts).shape, steps))
            return self.answer

    def alg_for(sampler, k=2, **kw):
        a = bc.SparseVICoreset(Z, NumpyFusedProjector(sampler, 8, 1.0), **kw)
        a.wts, a.idcs, a.pts = np.ones(k), np.arange(k), Z[:k]
        return a

    assert alg_for(Sampler("plan"), opt_itrs=5)._enqueue_plan() == "plan" and calls == [(8, (2, 5), 5)]
    assert alg_for(Sampler(None), opt_itrs=5)._enqueue_plan() is None                       # the sampler declines
    assert alg_for(lambda n, w, p: np.zeros((n, 4)), opt_itrs=5)._enqueue_plan() is None    # no device form
    assert alg_for(Sampler("plan"), k=65, opt_itrs=5)._enqueue_plan() == "plan"             # (the reference grows coresets to 300 points)
    n = len(calls)
    assert alg_for(Sampler("plan"), opt_itrs=5, n_subsample_opt=100)._enqueue_plan() is None     # sub-sample drawn on the host
    assert alg_for(Sampler("plan"), opt_itrs=0)._enqueue_plan() is None
    assert alg_for(Sampler("plan"), k=0, opt_itrs=5)._enqueue_plan() is None
    assert alg_for(Sampler("plan"), k=4097, opt_itrs=5)._enqueue_plan() is None
    off = alg_for(Sampler("plan"), opt_itrs=5)
    off.ENQUEUE = False
    assert off._enqueue_plan() is None and len(calls) == n                                  # none of these asked the sampler
    other = bc.SparseVICoreset(Z, bc.BlackBoxProjector(Sampler("plan"), 8, lambda z, th: linreg_log_likelihood(z, th, 1.0)), opt_itrs=5)
    other.wts, other.idcs, other.pts = np.ones(2), np.arange(2), Z[:2]
    assert other._enqueue_plan() is None and len(calls) == n                                # host projector: host loop
