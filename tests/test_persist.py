"""GPU parity of the opt-in several-iterations-per-launch form of GIGA / Frank-Wolfe (BCX_PERSIST=1; csrc/persist.hip: the
tail's workgroup resident beside the scan's, hand-offs by stamps) against the one-launch-per-kernel form on the same solver inputs:
selection sequence, per-iteration error, status and weights must be IDENTICAL (same partials, same arithmetic), across
launch boundaries inside a build, across build() calls, through the exact-scan redo of tie-heavy rows and through GIGA's
latch.  The reference loop both implement: snnls.py:41-74."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bc():
    import bayesiancoresets_amd as bc
    return bc


class _env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _cls(bc, alg):
    return {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe}[alg]


def _run(bc, X, alg, calls, batched, chunk=None):
    env = {"BCX_PERSIST": "1" if batched else None, "BCX_PERSIST_REQUIRE": "1" if batched else None,
           "BCX_PERSIST_CHUNK": chunk}
    with _env(**env):
        s = _cls(bc, alg)(X.T, X.sum(axis=0))
        traces = []
        for m in calls:
            s.build(m)
            traces.append(tuple(np.array(t) for t in s.last_trace))
        return traces, s.weights(), s.error()


def _same(a, b):
    (ta, wa, ea), (tb, wb, eb) = a, b
    assert len(ta) == len(tb)
    for x, y in zip(ta, tb):
        for u, v in zip(x, y):
            assert np.array_equal(u, v)
    assert np.array_equal(wa, wb)
    assert ea == eb


@pytest.mark.parametrize("alg", ("giga", "fw"))
@pytest.mark.parametrize("shape", ((200000, 256), (60000, 100), (30000, 37), (40000, 512), (5000, 700)))
def test_batched_launches_equal_one_launch_per_kernel(bc, alg, shape):
    n, d = shape
    X = np.random.RandomState(11 + d).randn(n, d)
    one = _run(bc, X, alg, (40,), batched=False)
    _same(_run(bc, X, alg, (40,), batched=True), one)              # one launch holds the call
    _same(_run(bc, X, alg, (40,), batched=True, chunk=7), one)     # launch boundaries inside the call (7, 7, ..., 5)
    _same(_run(bc, X, alg, (40,), batched=True, chunk=2), one)


@pytest.mark.parametrize("alg", ("giga", "fw"))
def test_batched_launches_across_build_calls(bc, alg):
    X = np.random.RandomState(5).randn(80000, 128)
    calls = (5, 17, 1, 2, 30)
    _same(_run(bc, X, alg, calls, batched=True, chunk=4), _run(bc, X, alg, calls, batched=False))


@pytest.mark.parametrize("alg", ("giga", "fw"))
def test_batched_launches_against_the_oracle(bc, alg):
    from oracle.snnls_oracle import SnnlsOracle
    X = np.random.RandomState(8).randn(20000, 96)
    with _env(BCX_PERSIST="1", BCX_PERSIST_REQUIRE="1", BCX_PERSIST_CHUNK="9"):
        s = _cls(bc, alg)(X.T, X.sum(axis=0))
        s.build(60)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg)
    o.build(60)
    assert np.array_equal(s.last_trace[0], np.array([t[0] for t in o.trace]))
    ow = o.weights()
    w = s.weights()
    assert np.array_equal(np.flatnonzero(w > 0), np.flatnonzero(ow > 0))
    np.testing.assert_allclose(w[w > 0], ow[ow > 0], rtol=1e-5)      # north_star tolerance for weights
    np.testing.assert_allclose(s.error(), o.error(), rtol=1e-7)


@pytest.mark.parametrize("alg", ("giga", "fw"))
def test_batched_launches_through_the_exact_redo(bc, alg):
    """Rows that tie (duplicates up to sign and scale; axis vectors) overflow the candidate window: the batch stops, the
    iteration is redone with the exact scan, the rest is enqueued again."""
    rs = np.random.RandomState(3)
    base = rs.randn(300, 64)
    X = np.concatenate([base] * 40 + [np.eye(64)] * 3, axis=0)
    one = _run(bc, X, alg, (30,), batched=False)
    _same(_run(bc, X, alg, (30,), batched=True, chunk=8), one)


def test_batched_launches_through_the_latch(bc):
    """GIGA on few rows reaches its numeric limit inside a batch: the trace, the latch and the weights are the same."""
    X = np.random.RandomState(2).randn(3000, 48)
    one = _run(bc, X, "giga", (400, 10), batched=False)
    got = _run(bc, X, "giga", (400, 10), batched=True, chunk=16)
    _same(got, one)


def test_off_unless_asked_for(bc):
    """The batched form is opt-in (BCX_PERSIST=1): without it REQUIRE is never consulted and the build takes one launch per kernel."""
    X = np.random.RandomState(1).randn(20000, 64)
    for val in (None, "0"):
        with _env(BCX_PERSIST=val, BCX_PERSIST_REQUIRE="1"):
            s = bc.snnls.FrankWolfe(X.T, X.sum(axis=0))
            s.build(10)
        assert len(s.last_trace[0]) == 10
