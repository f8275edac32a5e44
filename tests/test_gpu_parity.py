"""GPU parity: the HIP engine (through the C ABI / Python mirror) against the golden vectors
produced by the reference and against the CPU oracle on the same seeded inputs.
Indices must be bit-exact; weights within 1e-5 relative (BASELINE.json north_star)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WEIGHT_RTOL = 1e-5   # north_star tolerance for weights
WEIGHT_ATOL_REL = 1e-10  # absolute floor, relative to the largest weight: weights ~1e-9 of the largest
                         # are only determined to the conditioning of the (near-square) active system
ERR_RTOL = 1e-7      # per-iteration error ||Aw-b||


def _solver(bc, alg):
    return {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe, "omp": bc.snnls.OrthoPursuit}[alg]


def _run(bc, X, alg, itrs, **kw):
    s = _solver(bc, alg)(X.T, X.sum(axis=0), **kw)
    s.build(itrs)
    return s


def _check(s, golden, key, weight_rtol=WEIGHT_RTOL):
    sel, err, status = s.last_trace
    gsel = golden[key + "sel"]
    ok = status == 0
    assert np.array_equal(sel[sel >= 0], gsel), "selection sequence differs from the reference"
    gerr = golden[key + "err"]
    n = min(int(ok.sum()), len(gerr))
    np.testing.assert_allclose(err[ok][:n], gerr[:n], rtol=ERR_RTOL, atol=1e-9)
    w = s.weights()
    idx = np.flatnonzero(w > 0)
    assert np.array_equal(idx, golden[key + "idx"])
    gw = golden[key + "w"]
    np.testing.assert_allclose(w[idx], gw, rtol=weight_rtol, atol=WEIGHT_ATOL_REL * gw.max())
    np.testing.assert_allclose(s.error(), float(golden[key + "final_err"]), rtol=ERR_RTOL, atol=1e-9)


@pytest.fixture(scope="module")
def bc():
    import bayesiancoresets_amd as bc
    return bc


@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
@pytest.mark.parametrize("itrs", (12, 100))
def test_F1_axis_ties(bc, golden, alg, itrs):
    """X = eye(N): every score ties on every iteration -> first-index tie-break, overflow fallback."""
    X = np.eye(100)
    s = _run(bc, X, alg, itrs)
    _check(s, golden, "F1_%s_%d_" % (alg, itrs))
    assert list(s.last_trace[0][:12]) == list(range(12))


@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
@pytest.mark.parametrize("dtype", ("float32", "float64", "float16"))
def test_F9_small(bc, golden, normal_inputs, alg, dtype):
    X = normal_inputs(7, 3000, 64, "F9_input_sha256")
    s = _run(bc, X, alg, 60, dtype=dtype)
    _check(s, golden, "F9_%s_" % alg)


@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
@pytest.mark.parametrize("dtype", ("float32", "float64", "float16"))
def test_F2_normal_10k(bc, golden, normal_inputs, alg, dtype):
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    s = _run(bc, X, alg, 100, dtype=dtype)
    _check(s, golden, "F2_%s_" % alg)


@pytest.mark.parametrize("alg", ("giga", "fw"))
def test_no_exact_rows_mode(bc, golden, normal_inputs, alg):
    """fp32-only storage (no resident fp64 rows): same selections, weights to fp32-row accuracy."""
    X = normal_inputs(7, 3000, 64, "F9_input_sha256")
    s = _run(bc, X, alg, 60, keep_exact_rows=False)
    sel = s.last_trace[0]
    assert np.array_equal(sel[sel >= 0], golden["F9_%s_sel" % alg])
    w = s.weights()
    idx = np.flatnonzero(w > 0)
    assert np.array_equal(idx, golden["F9_%s_idx" % alg])
    np.testing.assert_allclose(w[idx], golden["F9_%s_w" % alg], rtol=1e-4)


@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
def test_incremental_build_matches_single_call(bc, normal_inputs, alg):
    """build() keeps state on the device across calls (examples/synthetic_vectors/main.py:91-94)."""
    X = normal_inputs(7, 3000, 64, "F9_input_sha256")
    a = _run(bc, X, alg, 40)
    b = _solver(bc, alg)(X.T, X.sum(axis=0))
    for step in (1, 1, 3, 5, 10, 20):
        b.build(step)
    np.testing.assert_allclose(a.weights(), b.weights(), rtol=1e-12, atol=0)
    assert a.error() == pytest.approx(b.error(), rel=1e-12)


@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
def test_against_oracle_random_shape(bc, alg):
    """Fresh seeded input (not in the golden file): HIP engine vs the CPU oracle, d not a multiple of 4."""
    from oracle.snnls_oracle import SnnlsOracle
    X = np.random.RandomState(123).randn(5000, 50) * np.random.RandomState(5).uniform(0.1, 10, size=(5000, 1))
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg)
    o.build(40)
    s = _run(bc, X, alg, 40)
    sel = s.last_trace[0]
    assert np.array_equal(sel, np.array([t[0] for t in o.trace]))
    w, ow = s.weights(), o.weights()
    assert np.array_equal(np.flatnonzero(w > 0), np.flatnonzero(ow > 0))
    np.testing.assert_allclose(w[w > 0], ow[ow > 0], rtol=WEIGHT_RTOL)
    np.testing.assert_allclose(s.error(), o.error(), rtol=ERR_RTOL)


@pytest.mark.parametrize("dtype", ("float32", "float16"))
@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
@pytest.mark.parametrize("d", (256, 512, 1024, 260))
def test_wide_rows_against_oracle(bc, alg, d, dtype):
    """d >= 256 takes the 64-lanes-per-row path of the scan kernel (4-row transposed reduction);
    d = 260 exercises the masked tail piece.  HIP engine vs CPU oracle on a fresh seeded input."""
    from oracle.snnls_oracle import SnnlsOracle
    N, itrs = 6000, 25
    X = np.random.RandomState(1000 + d).randn(N, d)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg)
    o.build(itrs)
    s = _run(bc, X, alg, itrs, dtype=dtype)
    assert np.array_equal(s.last_trace[0], np.array([t[0] for t in o.trace]))
    w, ow = s.weights(), o.weights()
    assert np.array_equal(np.flatnonzero(w > 0), np.flatnonzero(ow > 0))
    np.testing.assert_allclose(w[w > 0], ow[ow > 0], rtol=WEIGHT_RTOL)
    np.testing.assert_allclose(s.error(), o.error(), rtol=ERR_RTOL)


@pytest.mark.parametrize("dtype", ("float32", "float64"))
@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
@pytest.mark.parametrize("d", (4096, 5000, 8192, 16384, 20000, 41001))
def test_long_rows_against_oracle(bc, alg, d, dtype):
    """Rows beyond the register form of the scan (more than 1024 16-byte pieces: d > 4096 floats / 2048 doubles take the
    one-wave-per-row kernel with the query in LDS) and beyond the LDS budget of the O(d) state kernels (d > 3584: their
    five d-vectors live in global scratch; OMP at d >= 8192: the multi-kernel step; d = 20000: the query's tail beyond the LDS
    budget of the long scan comes from global memory and the constructor pass adds its column sums in place; d = 41001: odd,
    beyond every LDS budget).  The reference accepts any projection dimension (snnls.py:9-16, projector.py:12)."""
    from oracle.snnls_oracle import SnnlsOracle
    N, itrs = 2500, 20
    X = np.random.RandomState(2000 + d).randn(N, d)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg)
    o.build(itrs)
    s = _run(bc, X, alg, itrs, dtype=dtype)
    assert np.array_equal(s.last_trace[0], np.array([t[0] for t in o.trace]))
    w, ow = s.weights(), o.weights()
    assert np.array_equal(np.flatnonzero(w > 0), np.flatnonzero(ow > 0))
    np.testing.assert_allclose(w[w > 0], ow[ow > 0], rtol=WEIGHT_RTOL)
    np.testing.assert_allclose(s.error(), o.error(), rtol=ERR_RTOL)
    s.optimize()
    o.optimize()
    np.testing.assert_allclose(s.error(), o.error(), rtol=1e-6)


def test_omp_large_active_set_crosses_workgroup_widths(bc):
    """OMP for 520 iterations on Gaussian rows (d = 640): the active set grows through the step kernel's three workgroup
    widths (256 threads up to k = 192, 512 up to 448, 1024 beyond: the rows of the double-double inverse change owners
    between launches) -- selections, weights and error against the CPU oracle throughout."""
    from oracle.snnls_oracle import SnnlsOracle
    N, d, itrs = 12000, 640, 520
    X = np.random.RandomState(77).randn(N, d)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg="omp", mode="onepass")
    o.build(itrs)
    s = _run(bc, X, "omp", itrs)
    assert np.array_equal(s.last_trace[0], np.array([t[0] for t in o.trace]))
    w, ow = s.weights(), o.weights()
    assert np.array_equal(np.flatnonzero(w > 0), np.flatnonzero(ow > 0)) and (w > 0).sum() > 450
    np.testing.assert_allclose(w[w > 0], ow[ow > 0], rtol=WEIGHT_RTOL)
    np.testing.assert_allclose(s.error(), o.error(), rtol=ERR_RTOL)
    st = s._eng.omp_stats()
    assert st["steps"] == itrs and st["resolves"] == 0


def test_row_length_limit_is_a_value_error(bc):
    """(the bound is on the argument -- 2^20 -- not a capacity of a kernel; see test_long_rows_against_oracle)"""
    from bayesiancoresets_amd import _native as nat
    d = nat.MAX_ROW_LENGTH + 1
    X = np.zeros((2, d))
    with pytest.raises(ValueError, match=str(nat.MAX_ROW_LENGTH)):
        bc.snnls.FrankWolfe(X.T, np.ones(d))


def test_monotone_error_property(bc, normal_inputs):
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    for alg in ("giga", "fw"):
        s = _run(bc, X, alg, 150)
        sel, err, status = s.last_trace
        e = err[status == 0]
        assert np.all(np.diff(e[1:]) <= 0.0), "accepted steps must not increase the error (snnls.py:58)"


def test_error_paths(bc):
    X = np.random.RandomState(0).randn(2048, 8)
    X[700] = 0.0
    for alg in ("giga", "fw", "omp"):
        with pytest.raises(ValueError):
            _solver(bc, alg)(X.T, X.sum(axis=0))
    Y = np.random.RandomState(0).randn(64, 8)
    with pytest.raises(bc.util.errors.NumericalPrecisionError):
        bc.snnls.GIGA(Y.T, np.zeros(8))


def test_reset(bc, normal_inputs):
    X = normal_inputs(7, 3000, 64, "F9_input_sha256")
    s = _run(bc, X, "giga", 20)
    w1 = s.weights()
    s.reset()
    assert s.size() == 0 and s.error() == pytest.approx(np.sqrt((X.sum(axis=0) ** 2).sum()), rel=1e-13)
    s.build(20)
    np.testing.assert_array_equal(w1, s.weights())


@pytest.mark.parametrize("kernel", ("default", "incremental"))
@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
def test_F7_optimize(bc, golden, normal_inputs, alg, kernel, monkeypatch):
    """optimize(): NNLS re-solve on the support (snnls.py:82-97); Gram on the fp64 matrix cores.  Both solve kernels: the
    bordering + refined solve of nnls_grid.hip (what a support of this size takes) and the incremental Lawson-Hanson on the
    double-double inverse (omp_lh.hip optimize_lh_kernel; larger supports and every k > d), forced here by BCX_OPT_LH."""
    if kernel == "incremental":
        monkeypatch.setenv("BCX_OPT_LH", "1")
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    s = _run(bc, X, alg, 100)
    s.optimize()
    assert s._eng.omp_stats()["resolves"] == 0             # (the incremental solve passed its closing Newton check)
    w = s.weights()
    idx = np.flatnonzero(w > 0)
    assert np.array_equal(idx, golden["F7_%s_idx" % alg])
    gw = golden["F7_%s_w" % alg]
    np.testing.assert_allclose(w[idx], gw, rtol=WEIGHT_RTOL, atol=WEIGHT_ATOL_REL * gw.max())
    np.testing.assert_allclose(s.error(), float(golden["F7_%s_final_err" % alg]), rtol=ERR_RTOL, atol=1e-9)
    if alg != "omp":   # OMP sits at k = d = 100 with error ~1e-12 * ||b||: accept/reject of optimize() is rounding noise
        assert s.reached_numeric_limit == bool(golden["F7_%s_limit" % alg])


def _optimize_stats(s):
    """(started warm, warm passive set, columns entered, columns left, closing check failed) of the last optimize_lh launch
    (DevState::dbg_t[20..], read through the unlisted debug entry point bcx_debug_stamps)."""
    import ctypes
    st = (ctypes.c_longlong * 32)()
    lib = s._eng.lib
    lib.bcx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    assert lib.bcx_debug_stamps(s._eng.h, st) == 0
    return int(st[20]), int(st[21]), int(st[23]), int(st[24]), int(st[26])


@pytest.mark.parametrize("alg,N,d,itrs", (("fw", 8000, 640, 700), ("fw", 8000, 256, 700), ("giga", 6000, 200, 500),
                                          ("fw", 20000, 1024, 600), ("giga", 20000, 768, 700)))
def test_optimize_large_supports_against_the_oracle(bc, alg, N, d, itrs):
    """optimize() on supports beyond 512 columns and beyond d columns: the incremental double-double Lawson-Hanson kernel
    started WARM from the largest independent part of the support (csrc/warm.hip: blocked Cholesky with pivot rejection,
    inverse on the fp64 matrix cores) -- error against the oracle's scipy.optimize.nnls re-solve; support and weights too
    where the minimiser is unique (k <= d).  The warm start has to have been taken and accepted (no second, cold run)."""
    from oracle.snnls_oracle import SnnlsOracle
    X = np.random.RandomState(1000 + d).randn(N, d)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="onepass")
    o.build(itrs)
    s = _run(bc, X, alg, itrs)
    k = int((s.weights() > 0).sum())
    assert k > 512 or k > d
    e0 = s.error()
    s.optimize()
    o.optimize()
    assert s._eng.omp_stats()["resolves"] == 0
    warm, p0, entered, left, failed = _optimize_stats(s)
    assert warm == 1 and failed == 0 and 0 < p0 <= min(k, d), (warm, p0, entered, left, failed)
    assert entered + left < k, (entered, left, k)            # pivots from the warm state, not a rebuild of the set
    assert s.error() <= e0 * (1 + 1e-12)
    np.testing.assert_allclose(s.error(), o.error(), rtol=1e-6, atol=1e-9 * np.sqrt((X.sum(axis=0) ** 2).sum()))
    if k <= d:
        w, ow = s.weights(), o.weights()
        assert np.array_equal(np.flatnonzero(w > 0), np.flatnonzero(ow > 0))
        np.testing.assert_allclose(w[w > 0], ow[ow > 0], rtol=WEIGHT_RTOL, atol=WEIGHT_ATOL_REL * ow.max())
    assert s.reached_numeric_limit == o.reached_numeric_limit


@pytest.mark.parametrize("alg,rank", (("fw", 40), ("fw", 90)))
def test_optimize_on_a_support_of_low_rank(bc, alg, rank):
    """optimize() where the support's Gram matrix is numerically singular long before k reaches d (rows of rank `rank` in
    d = 128 dimensions, as the projected log-likelihoods of a small model are): the warm start's Cholesky has to leave out
    every column that depends on the ones before it (pivot rule, not the d-column cap), the incremental kernel carries on
    from that independent set, and the error equals the oracle's scipy.optimize.nnls re-solve."""
    from oracle.snnls_oracle import SnnlsOracle
    rs = np.random.RandomState(77 + rank)
    X = rs.randn(6000, rank).dot(rs.randn(rank, 128))
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="onepass")
    o.build(260)
    s = _run(bc, X, alg, 260)
    k = int((s.weights() > 0).sum())
    assert k >= 128 and k > rank                           # (the warm start takes supports of 128 columns and more)
    e0 = s.error()
    s.optimize()
    o.optimize()
    warm, p0, entered, left, failed = _optimize_stats(s)
    assert 0 < p0 <= rank + 2, (p0, rank)                  # the independent part of the support: about `rank` columns
    assert s.error() <= e0 * (1 + 1e-12)
    np.testing.assert_allclose(s.error(), o.error(), rtol=1e-6, atol=1e-9 * np.sqrt((X.sum(axis=0) ** 2).sum()))
    assert s.reached_numeric_limit == o.reached_numeric_limit


def test_hilbert_coreset_api(bc, golden, normal_inputs):
    """Drop-in surface of examples/synthetic_vectors/main.py:82-99."""
    X = normal_inputs(7, 3000, 64, "F9_input_sha256")

    class IDProjector(bc.Projector):
        def update(self, wts, pts):
            pass

        def project(self, pts, grad=False):
            return pts

    alg = bc.HilbertCoreset(X, IDProjector(), snnls=bc.snnls.GIGA)
    assert alg.get()[0].shape == (0,)
    alg.build(0)
    assert alg.size() == 0
    for step in (10, 20, 30):
        alg.build(step)
    wts, pts, idcs = alg.get()
    assert np.array_equal(idcs, golden["F9_giga_idx"])
    np.testing.assert_allclose(wts, golden["F9_giga_w"], rtol=WEIGHT_RTOL)
    assert np.array_equal(pts, X[idcs])
    np.testing.assert_allclose(alg.error(), float(golden["F9_giga_final_err"]), rtol=ERR_RTOL)
    alg.optimize()
    assert alg.error() <= float(golden["F9_giga_final_err"]) * (1 + 1e-12)
    alg.reset()
    assert alg.size() == 0 and alg.snnls.size() == 0
    with pytest.raises(TypeError):
        bc.HilbertCoreset(X, IDProjector(), snnls=bc.snnls.GIGA, bogus=1)


# ---- numeric-limit regime and the config-1 harness (F3) -------------------------------
def _graded_error_trace(err, ref):
    """The whole error trace on the way into the numeric floor, at the accuracy the arithmetic allows: the error is a
    difference of vectors of size ||b||, so its absolute rounding noise is ~1e-16 ||b|| whatever its own size -- 1e-9
    relative plus 2e-14 of the first error (||b||-sized) absolute, i.e. 1e-9 relative while the error is above 1e-5 of its
    start and proportionally looser below (a flat rtol = 1e-4, which the floor regime needs, would let the first three
    hundred iterations drift by five digits)."""
    n = min(len(err), len(ref))
    assert n >= len(ref) - 2
    np.testing.assert_allclose(err[:n], ref[:n], rtol=1e-9, atol=2e-14 * ref[0])


def _graded_close(value, ref, scale):
    """One error value in the floor regime under the same graded bound: 1e-9 relative + 2e-14 of `scale` (the size of the
    vectors whose difference the error is: ||b||, or the first error of the run where the iterate starts far from b)."""
    np.testing.assert_allclose(value, ref, rtol=1e-9, atol=2e-14 * scale)


def test_F4_giga_latch(bc, golden, normal_inputs):
    """GIGA driven to the numeric limit: same 429 selections, then select fails twice in a row
    (cdirnrm < TOL, giga.py:28) and the solver latches (snnls.py:63-72) at the same place."""
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    s = _run(bc, X, "giga", int(golden["F4_giga_itrs"]))
    sel, err, status = s.last_trace
    assert np.array_equal(sel[sel >= 0], golden["F4_giga_sel"])
    assert len(sel) == len(golden["F4_giga_sel"]) + 2 and list(status[-2:]) == [1, 1] and list(sel[-2:]) == [-1, -1]
    assert s.reached_numeric_limit is True and bool(golden["F4_giga_limit"])
    assert s.size() == int(golden["F4_giga_size"])
    _graded_close(s.error(), float(golden["F4_giga_final_err"]), float(np.linalg.norm(X.sum(axis=0))))
    _graded_error_trace(err[status == 0], golden["F4_giga_err"])
    w = s.weights()
    assert np.array_equal(np.flatnonzero(w > 0), golden["F4_giga_idx"])
    gw = golden["F4_giga_w"]   # at the numeric limit weights ~1e-10 of the largest carry only rounding noise
    np.testing.assert_allclose(w[w > 0], gw, rtol=WEIGHT_RTOL, atol=WEIGHT_ATOL_REL * gw.max())
    # latched: further build() calls return immediately (snnls.py:32-34)
    s.build(10)
    assert s.size() == int(golden["F4_giga_size"])


def test_F4_fw_400(bc, golden, normal_inputs):
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    s = _run(bc, X, "fw", 400)
    sel, err, status = s.last_trace
    assert np.array_equal(sel, golden["F4_fw_sel"])
    assert s.size() == int(golden["F4_fw_size"]) and not s.reached_numeric_limit
    # (Frank-Wolfe's iterate starts at a polytope vertex of size sum(norms): the first error, 9.9e4, is the scale of the
    #  vectors its error is the difference of)
    _graded_close(s.error(), float(golden["F4_fw_final_err"]), float(golden["F4_fw_err"][0]))
    _graded_error_trace(err, golden["F4_fw_err"])


def test_F4_omp_past_k_equals_d(bc, golden, normal_inputs):
    """OMP for 140 iterations at d = 100: the active set saturates at d points, no latch, no exception;
    once k = d the scores are rounding noise, so selections are compared only while k < d."""
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    s = _run(bc, X, "omp", 140)
    sel = s.last_trace[0]
    assert np.array_equal(sel[:100], golden["F4_omp_sel"][:100])
    assert s.size() == 100 and not s.reached_numeric_limit
    # past k = d the weights are determined only up to the solver's tolerance: compare through A w.
    # Both solutions reproduce b; their images differ by no more than the two errors, each under the graded floor bound
    bn = float(np.linalg.norm(X.sum(axis=0)))
    gw = np.zeros(X.shape[0])
    gw[golden["F4_omp_idx"]] = golden["F4_omp_w"]
    w = s.weights()
    assert (w >= 0).all()
    gap = float(np.linalg.norm(X.T.dot(w) - X.T.dot(gw)))
    assert gap <= 2e-14 * bn + float(golden["F4_omp_final_err"]) * (1 + 1e-9), (gap, s.error(), bn)
    assert s.error() <= 2e-14 * bn, (s.error(), bn)


@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
def test_F3_harness_trial1(bc, golden, normal_inputs, alg):
    """examples/synthetic_vectors/main.py:82-99 with the Ms schedule: incremental build, get(), error()."""
    X = normal_inputs(1, 10000, 100, "F3_t1_input_sha256")

    class IDProjector(bc.Projector):
        def update(self, wts, pts):
            pass

        def project(self, pts, grad=False):
            return pts

    Ms = golden["F3_Ms"]
    a = bc.HilbertCoreset(X, IDProjector(), snnls=_solver(bc, alg))
    csize, err = [], []
    for m in range(len(Ms)):
        a.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
        wts, pts, idcs = a.get()
        csize.append((wts > 0).sum())
        err.append(a.error())
    csize, err = np.array(csize, dtype=float), np.array(err)
    k = "F3_t1_%s_" % alg
    gc, ge = golden[k + "csize"], golden[k + "err"]
    # away from the numeric limit everything matches; near it (error ~1e-9 of ||b||) FW's accept/reject
    # tests become rounding-dependent, so FW is compared up to M = 429 and OMP errors while k < d
    upto = {"giga": len(Ms), "fw": int(np.searchsorted(Ms, 429)) + 1, "omp": len(Ms)}[alg]
    assert np.array_equal(csize[:upto], gc[:upto])
    sane = ge[:upto] > 1e-6
    np.testing.assert_allclose(err[:upto][sane], ge[:upto][sane], rtol=1e-6)
    if alg == "giga":   # (FW / OMP end in the rounding-noise regime where accept/reject is luck in the reference too)
        assert a.snnls.reached_numeric_limit == bool(golden[k + "limit"])
        wts, pts, idcs = a.get()
        assert np.array_equal(idcs, golden[k + "idcs"])
        np.testing.assert_allclose(wts, golden[k + "wts"], rtol=WEIGHT_RTOL,
                                   atol=WEIGHT_ATOL_REL * golden[k + "wts"].max())


# ---- edge cases: ragged / tiny / maximal shapes, subsampling, empty input ------------------------------
@pytest.mark.parametrize("N,d", ((1, 1), (5, 3), (1023, 7), (1025, 33), (2049, 1), (3000, 2048), (4097, 129)))
@pytest.mark.parametrize("alg", ("giga", "fw", "omp"))
def test_ragged_and_extreme_shapes(bc, alg, N, d):
    from oracle.snnls_oracle import SnnlsOracle
    X = np.random.RandomState(N * 31 + d).randn(N, d)
    itrs = min(12, N + 2)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg)
    o.build(itrs)
    s = _run(bc, X, alg, itrs)
    sel, err, status = s.last_trace
    otrace = o.trace
    # compare while the oracle's error is above rounding noise (tiny problems converge exactly)
    scale = np.sqrt((X.sum(axis=0) ** 2).sum())
    n = 0
    for t in otrace:
        if t[2] != 0 or t[1] < 1e-9 * scale:
            break
        n += 1
    n = max(n, 1)
    assert np.array_equal(sel[:n], np.array([t[0] for t in otrace[:n]]))
    np.testing.assert_allclose(err[:n], np.array([t[1] for t in otrace[:n]]), rtol=1e-7, atol=1e-9 * scale)


def test_empty_input_warns_and_returns(bc):
    """snnls.py:36-38: no data -> build() returns at once."""
    X = np.zeros((0, 4))
    s = bc.snnls.FrankWolfe(X.T, np.zeros(4))
    s.build(5)
    assert s.size() == 0 and s.weights().shape == (0,)


def test_hilbert_subsample_branch(bc):
    """hilbert.py:13-22: unique(randint) subsample, zero vectors dropped, indices map back to the data."""
    from oracle.snnls_oracle import SnnlsOracle, hilbert_readout
    X = np.random.RandomState(77).randn(6000, 24)
    X[10] = 0.0
    X[4000] = 0.0

    class IDProjector(bc.Projector):
        def update(self, wts, pts):
            pass

        def project(self, pts, grad=False):
            return pts

    np.random.seed(5)
    c = bc.HilbertCoreset(X, IDProjector(), n_subsample=3000, snnls=bc.snnls.GIGA)
    np.random.seed(5)
    sub = np.unique(np.random.randint(6000, size=3000))
    sub = sub[np.sqrt((X[sub] ** 2).sum(axis=1)) > 0]
    assert np.array_equal(c.sub_idcs, sub)
    c.build(20)
    wts, pts, idcs = c.get()
    V = X[sub]
    o = SnnlsOracle(V.T, V.sum(axis=0), alg="giga")
    o.build(20)
    ow, oidx = hilbert_readout(o.weights(), sub)
    assert np.array_equal(idcs, oidx)
    np.testing.assert_allclose(wts, ow, rtol=WEIGHT_RTOL)
    assert np.array_equal(pts, X[idcs])


def test_float32_and_torch_inputs(bc):
    """Rows may arrive as float32 host arrays or as torch tensors already resident on the GPU."""
    import torch
    from oracle.snnls_oracle import SnnlsOracle
    X = np.random.RandomState(8).randn(5000, 48)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg="fw")
    o.build(15)
    want = np.array([t[0] for t in o.trace])
    Xd = torch.from_numpy(X).cuda()
    s = bc.snnls.FrankWolfe(Xd.t(), None)          # b from the device column sums
    s.build(15)
    assert np.array_equal(s.last_trace[0], want)
    np.testing.assert_allclose(s._eng.vector(0), X.sum(axis=0), rtol=1e-12, atol=1e-12)
    X32 = X.astype(np.float32)
    o32 = SnnlsOracle(X32.astype(np.float64).T, X32.astype(np.float64).sum(axis=0), alg="fw")
    o32.build(15)
    s32 = bc.snnls.FrankWolfe(X32.T, X32.astype(np.float64).sum(axis=0))
    s32.build(15)
    assert np.array_equal(s32.last_trace[0], np.array([t[0] for t in o32.trace]))


def test_candidate_storm_lowrank_rows(bc):
    """Effectively 2-dimensional data: hundreds of rows sit inside the fp32 candidate window, the iteration must
    take the exact fp64 scan (regression test: the overflowing candidate list used to be read out of bounds)."""
    from oracle.snnls_oracle import SnnlsOracle
    rs = np.random.RandomState(5)
    N, d = 120000, 17
    X = rs.randn(N, 2).dot(rs.randn(2, d)) + 1e-3 * rs.randn(N, d)
    for alg in ("giga", "fw"):
        o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="onepass")
        o.build(12)
        s = _run(bc, X, alg, 12)
        scale = np.sqrt((X.sum(axis=0) ** 2).sum())
        n = sum(1 for t in o.trace if t[2] == 0 and t[1] > 1e-7 * scale)
        assert n >= 3
        assert np.array_equal(s.last_trace[0][:n], np.array([t[0] for t in o.trace[:n]]))


@pytest.mark.parametrize("seed", (11, 12, 13))
def test_randomized_parity_sweep(bc, seed):
    """Random shapes / algorithms / storage types / data pathologies (duplicates, 11 decades of row scale,
    bundles of nearly parallel rows, low rank) against the CPU oracle (tests/stress_parity.py, reduced)."""
    from oracle.snnls_oracle import SnnlsOracle
    rs = np.random.RandomState(seed)
    for case in range(14):
        N = int(rs.choice([700, 5000, 30000]))
        d = int(rs.choice([3, 17, 64, 100, 256, 300, 512]))
        alg = str(rs.choice(["giga", "fw", "omp"]))
        dtype = str(rs.choice(["float32", "float16", "float64"]))
        kind = str(rs.choice(["plain", "dups", "scaled", "parallel", "lowrank"]))
        X = rs.randn(N, d)
        if kind == "dups":
            src, dst = rs.randint(0, N, size=N // 10), rs.randint(0, N, size=N // 10)
            X[dst] = X[src]
        elif kind == "scaled":
            X *= 10.0 ** rs.uniform(-8, 3, size=(N, 1))
        elif kind == "parallel":
            X[: N // 20] = rs.randn(d) * rs.uniform(0.5, 2.0, size=(N // 20, 1)) + 1e-7 * rs.randn(N // 20, d)
        elif kind == "lowrank":
            r = max(1, d // 8)
            X = rs.randn(N, r).dot(rs.randn(r, d)) + 1e-3 * rs.randn(N, d)
        itrs = int(min(30, d + 5))
        o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="onepass")
        o.build(itrs)
        s = _run(bc, X, alg, itrs, dtype=dtype)
        scale = np.sqrt((X.sum(axis=0) ** 2).sum())
        n = 0
        for t in o.trace:
            if t[2] != 0 or t[1] < 1e-7 * scale:
                break
            n += 1
        n = min(n, len(s.last_trace[0]))
        tag = "case %d: N=%d d=%d %s %s %s" % (case, N, d, alg, dtype, kind)
        assert np.array_equal(s.last_trace[0][:n], np.array([t[0] for t in o.trace[:n]])), tag
        np.testing.assert_allclose(s.last_trace[1][:n], np.array([t[1] for t in o.trace[:n]]), rtol=1e-6,
                                   atol=1e-9 * scale, err_msg=tag)


# ---- the experiment harness end to end -------------------------------------------------
@pytest.mark.parametrize("trial", (1, 2, 4, 5))
def test_harness_cli_writes_reference_results(golden, tmp_path, trial):
    """`main.py --alg GIGA --trial t run` (examples/synthetic_vectors/run_experiment.sh:7) stores Ms / csize /
    err columns equal to the reference's stored run for the same arguments; a second call is a no-op."""
    import subprocess
    import pandas as pd
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd", "examples",
                          "synthetic_vectors", "main.py")
    folder = str(tmp_path / "results") + "/"
    cmd = [sys.executable, script, "--alg", "GIGA", "--trial", str(trial), "--data_type", "normal",
           "--results_folder", folder, "run"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    files = [f for f in os.listdir(folder) if f != "manifest.csv"]
    assert len(files) == 1
    t = pd.read_csv(os.path.join(folder, files[0]))
    assert list(t.columns[:10]) == ["alg", "data_num", "data_dim", "data_type", "coreset_size_max",
                                    "coreset_num_sizes", "coreset_size_spacing", "trial", "results_folder",
                                    "verbosity"]
    if trial > 3:      # seeds 4 and 5: tests/golden/harness45_golden.npz (make_golden_harness45.py)
        golden = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "harness45_golden.npz"))
    k = "F3_t%d_giga_" % trial
    assert np.array_equal(t["Ms"].to_numpy(), golden["F3_Ms"])
    assert np.array_equal(t["csize"].to_numpy(), golden[k + "csize"])
    ge = golden[k + "err"]
    np.testing.assert_allclose(t["err"].to_numpy()[ge > 1e-6], ge[ge > 1e-6], rtol=1e-6)
    assert (np.diff(t["cput"].to_numpy()) >= 0).all() and (np.diff(t["wall"].to_numpy()) > 0).all()
    again = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert again.returncode == 0 and "Results already exist" in again.stdout
    assert len(open(os.path.join(folder, "manifest.csv")).read().splitlines()) == 1


# ---- constructor kernels: vectorised ingest == scalar ingest ---------------------------------------------
@pytest.mark.parametrize("d", (8, 64, 100, 512, 1030, 2048))
@pytest.mark.parametrize("src", ("float64", "float32"))
def test_vectorised_ingest_matches_scalar(bc, d, src, monkeypatch):
    """csrc/ingest.hip: the 16-byte-piece kernel and the scalar kernel (odd d / unaligned rows) must produce the
    same norms, the same column sums b (bit for bit: same summation tree) and the same stored rows (same trace)."""
    N = 3000
    X = np.random.RandomState(d).randn(N, d).astype(src)
    X[7] *= 1e-30          # tiny and huge rows
    X[11] *= 1e20
    out = []
    for scalar in (False, True):
        if scalar:
            monkeypatch.setenv("BCX_INGEST_SCALAR", "1")
        else:
            monkeypatch.delenv("BCX_INGEST_SCALAR", raising=False)
        s = bc.snnls.FrankWolfe(X.T, None)
        s.build(8)
        out.append((s.Anorms.copy(), np.array(s.b), s.last_trace[0].copy(), s.last_trace[1].copy()))
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    np.testing.assert_allclose(out[0][0], np.sqrt((X.astype(np.float64) ** 2).sum(axis=1)), rtol=1e-14)


# ---- multi-workgroup kernels: the same input must give the same bits every time -----------------------------
@pytest.mark.parametrize("alg,N,d,itrs,reps", (("giga", 30000, 100, 90, 200), ("omp", 20000, 300, 120, 60),
                                                ("fw", 5000, 256, 30, 100)))
def test_repeated_runs_are_bit_identical(bc, alg, N, d, itrs, reps):
    """optimize() and the OMP step are single launches of 16 workgroups on different XCDs (grid barriers, per-XCD
    L2s that are not coherent with each other).  A hand-off that is not published / acquired properly shows up as a
    rare deviation between runs (the first version of optimize_grid_kernel: 0.3 % of the runs of the first case;
    tests/race_hunt.py is the long-running form of this test)."""
    import hashlib
    X = np.random.RandomState(N + d).randn(N, d)
    b = X.sum(axis=0)
    seen = set()
    for _ in range(reps):
        s = _solver(bc, alg)(X.T, b)
        s.build(itrs)
        h1 = hashlib.md5(s.weights().tobytes() + np.float64(s.error()).tobytes()).hexdigest()
        s.optimize()
        h2 = hashlib.md5(s.weights().tobytes() + np.float64(s.error()).tobytes()).hexdigest()
        seen.add((h1, h2, bool(s.reached_numeric_limit)))
    assert len(seen) == 1, "%d distinct outcomes in %d identical runs" % (len(seen), reps)


# ---- check_error_monotone = False on the device state machine (reference fixture F10, tests/golden/make_golden_host.py) ----
@pytest.fixture(scope="module")
def host_golden():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return np.load(os.path.join(root, "tests", "golden", "host_golden.npz"))


@pytest.mark.parametrize("how", ("ctor", "attribute"))
def test_F10_giga_without_monotone_check(bc, host_golden, normal_inputs, how):
    """snnls.py:45,56-62 switched off: same 429 selections and the same latch as the reference run with
    ``solver.check_error_monotone = False``."""
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    if how == "ctor":
        s = bc.snnls.GIGA(X.T, X.sum(axis=0), check_error_monotone=False)
    else:
        s = bc.snnls.GIGA(X.T, X.sum(axis=0))
        s.check_error_monotone = False
    assert s.check_error_monotone is False
    s.build(int(host_golden["F10_giga_itrs"]))
    sel, err, status = s.last_trace
    assert np.array_equal(sel[sel >= 0], host_golden["F10_giga_sel"])
    assert not (status == 3).any()                                   # no monotone failures can be reported
    assert s.reached_numeric_limit == bool(host_golden["F10_giga_limit"]) and s.size() == int(host_golden["F10_giga_size"])
    _graded_close(s.error(), float(host_golden["F10_giga_final_err"]), float(np.linalg.norm(X.sum(axis=0))))
    _graded_error_trace(err[status == 0], host_golden["F10_giga_err"])


def test_F10_fw_without_monotone_check(bc, host_golden, normal_inputs):
    X = normal_inputs(1, 10000, 100, "F2_input_sha256")
    s = bc.snnls.FrankWolfe(X.T, X.sum(axis=0), check_error_monotone=False)
    s.build(int(host_golden["F10_fw_itrs"]))
    sel, err, status = s.last_trace
    n_ok = int((status == 0).sum())
    assert np.array_equal(sel[sel >= 0][:400], host_golden["F10_fw_sel"][:400])
    assert not (status == 3).any() and s.reached_numeric_limit == bool(host_golden["F10_fw_limit"])
    _graded_error_trace(err[status == 0][:400], host_golden["F10_fw_err"][:400])
    assert n_ok >= 400


def test_monotone_flag_changes_the_state_machine(bc):
    """A case where the two settings differ on the device as they do in the oracle: rows of rank 3 make Frank-Wolfe
    reach its floor within a few steps; with the check every further step that raises the error is reverted and the
    second such failure latches, without it steps keep being accepted."""
    from oracle.snnls_oracle import SnnlsOracle
    rs = np.random.RandomState(8)
    X = rs.randn(600, 3).dot(rs.randn(3, 24))
    out = {}
    for flag in (True, False):
        o = SnnlsOracle(X.T, X.sum(axis=0), alg="fw", mode="onepass", check_error_monotone=flag)
        o.build(80)
        s = bc.snnls.FrankWolfe(X.T, X.sum(axis=0), check_error_monotone=flag, dtype="float64")
        s.build(80)
        sel, err, status = s.last_trace
        ost = np.array([t[2] for t in o.trace])
        n = min(len(ost), len(status))
        # identical control flow for as long as the error is far above rounding noise
        scale = np.sqrt((X.sum(axis=0) ** 2).sum())
        far = np.array([t[1] for t in o.trace])[:n] > 1e-9 * scale
        k = int(np.argmin(far)) if not far.all() else n
        assert k >= 3 and np.array_equal(status[:k], ost[:k]) and np.array_equal(sel[:k], np.array([t[0] for t in o.trace])[:k])
        out[flag] = (status, s.reached_numeric_limit)
    assert not (out[False][0] == 3).any()


def test_torch_inputs_of_other_dtypes_are_converted(bc):
    """Round-1 ADVICE: a half-precision or integer tensor must not be reinterpreted as fp32 / fp64 bytes."""
    import torch
    rs = np.random.RandomState(2)
    Xi = rs.randint(-5, 6, size=(3000, 24))
    Xi[np.abs(Xi).sum(axis=1) == 0, 0] = 1
    want = _run(bc, Xi.astype(np.float64), "fw", 15)
    for t in (torch.from_numpy(Xi), torch.from_numpy(Xi).cuda(), torch.from_numpy(Xi.astype(np.float16)).cuda()):
        s = bc.snnls.FrankWolfe(t.t(), None)
        s.build(15)
        assert np.array_equal(s.last_trace[0], want.last_trace[0])
        np.testing.assert_allclose(s.weights(), want.weights(), rtol=1e-9)
