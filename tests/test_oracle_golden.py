"""Pin the CPU oracle (oracle/snnls_oracle.py) against golden vectors produced by
the reference itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle.snnls_oracle import SnnlsOracle, hilbert_readout, harness_sizes, ST_OK

ALGS = ("giga", "fw", "omp")


def run_trace(X, alg, itrs, mode):
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode=mode)
    o.build(itrs)
    return o


def check_against(o, golden, key, mode):
    sel = np.array([t[0] for t in o.trace if t[0] >= 0], dtype=np.int64)
    gsel = golden[key + "sel"]
    assert np.array_equal(sel, gsel), "selection order differs"
    errs = np.array([t[1] for t in o.trace if t[2] == ST_OK])
    gerr = golden[key + "err"]
    n = min(len(errs), len(gerr))
    if mode == "faithful":
        assert np.array_equal(errs[:n], gerr[:n])
    else:
        np.testing.assert_allclose(errs[:n], gerr[:n], rtol=1e-9, atol=1e-9)
    w, idx = hilbert_readout(o.weights())
    assert np.array_equal(idx, golden[key + "idx"])
    if mode == "faithful":
        assert np.array_equal(w, golden[key + "w"])
    else:
        np.testing.assert_allclose(w, golden[key + "w"], rtol=1e-9)


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("itrs", (12, 100))
@pytest.mark.parametrize("mode", ("faithful", "onepass"))
def test_F1_axis(golden, alg, itrs, mode):
    X = np.eye(100)
    o = run_trace(X, alg, itrs, mode)
    check_against(o, golden, "F1_%s_%d_" % (alg, itrs), mode)
    # known answers quoted in SURVEY.md section 4
    assert [t[0] for t in o.trace[:12]] == list(range(12))


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("mode", ("faithful", "onepass"))
def test_F2_normal_10k(golden, normal_inputs, alg, mode):
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    o = run_trace(X, alg, 100, mode)
    check_against(o, golden, "F2_%s_" % alg, mode)
    if mode == "faithful":
        # F7: optimize() on that state
        o.optimize()
        w, idx = hilbert_readout(o.weights())
        assert np.array_equal(idx, golden["F7_%s_idx" % alg])
        assert np.array_equal(w, golden["F7_%s_w" % alg])
        assert o.error() == float(golden["F7_%s_final_err" % alg])
        assert o.reached_numeric_limit == bool(golden["F7_%s_limit" % alg])


@pytest.mark.parametrize("alg", ALGS)
@pytest.mark.parametrize("mode", ("faithful", "onepass"))
def test_F9_small(golden, normal_inputs, alg, mode):
    X = normal_inputs(7, 3000, 64, "F9_input_sha256")
    o = run_trace(X, alg, 60, mode)
    check_against(o, golden, "F9_%s_" % alg, mode)


@pytest.mark.parametrize("alg", ("fw", "omp", "giga"))
def test_F4_numeric_limit(golden, normal_inputs, alg):
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    itrs = int(golden["F4_%s_itrs" % alg])
    o = run_trace(X, alg, itrs, "faithful")
    k = "F4_%s_" % alg
    sel = np.array([t[0] for t in o.trace if t[0] >= 0], dtype=np.int64)
    assert np.array_equal(sel, golden[k + "sel"])
    assert o.reached_numeric_limit == bool(golden[k + "limit"])
    assert o.size() == int(golden[k + "size"])
    assert o.error() == float(golden[k + "final_err"])
    w, idx = hilbert_readout(o.weights())
    assert np.array_equal(idx, golden[k + "idx"])
    assert np.array_equal(w, golden[k + "w"])


@pytest.mark.parametrize("alg", ALGS)
def test_F3_harness_trial1(golden, normal_inputs, alg):
    """examples/synthetic_vectors harness: incremental build over the Ms schedule."""
    X = normal_inputs(1, 10000, 100, "F3_t1_input_sha256")
    Ms = harness_sizes()
    assert np.array_equal(Ms, golden["F3_Ms"])
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="faithful")
    csize, err = [], []
    for m in range(len(Ms)):
        o.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
        csize.append(o.size())
        err.append(o.error())
    k = "F3_t1_%s_" % alg
    assert np.array_equal(np.array(csize, dtype=float), golden[k + "csize"])
    assert np.array_equal(np.array(err), golden[k + "err"])
    assert o.reached_numeric_limit == bool(golden[k + "limit"])


def test_F8_error_paths():
    X = np.random.RandomState(0).randn(50, 8)
    X[7] = 0.0
    for alg in ALGS:
        with pytest.raises(ValueError):
            SnnlsOracle(X.T, X.sum(axis=0), alg=alg)
    Y = np.random.RandomState(0).randn(50, 8)
    with pytest.raises(ArithmeticError):
        SnnlsOracle(Y.T, np.zeros(8), alg="giga")
