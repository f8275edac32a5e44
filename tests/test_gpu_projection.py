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


@pytest.mark.parametrize("family", ("logistic", "poisson", "linreg"))
def test_a_rows_projection_does_not_depend_on_how_many_rows_its_shard_holds(bc, family):
    """Data rows go through ONE kernel whatever their number (beyond the 32-row form): a row's values are the same to the last
    bit in every slice of the data -- what lets row shards of any size reproduce the single-shard build.  The
    32 x 32 kernel for a few hundred rows (csrc/proj.hip proj_mid_kernel) sums the inner index in another order and serves
    the coreset points only (bcx_project_write_points): checked here against NumPy and against the data-row kernel to rounding."""
    import torch
    Z, theta, ll, sigsq = _cases()[family]
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, theta.shape[0], sigsq=sigsq)
    full_raw = prj.project_uncentred(Z).cpu().numpy()
    full = prj.project(Z).cpu().numpy()
    for lo, hi in ((0, 33), (100, 400), (1000, 5096), (17, 4000), (4000, Z.shape[0])):
        assert np.array_equal(prj.project_uncentred(Z[lo:hi]).cpu().numpy(), full_raw[lo:hi]), (lo, hi)
        assert np.array_equal(prj.project(Z[lo:hi]).cpu().numpy(), full[lo:hi]), (lo, hi)
    S = theta.shape[0]
    scale = np.abs(full_raw).max()
    for lo, hi in ((0, 33), (100, 400), (7, 2008)):
        C = prj._dev(Z[lo:hi])
        for center, want in ((0, full_raw), (1, full)):
            out = torch.full((hi - lo, S + 1), -7.0, dtype=torch.float64, device="cuda")
            prj._launch(prj._lib.bcx_project_write_points, prj._common(C) + [out.data_ptr(), S + 1, center], C)
            got = out.cpu().numpy()
            np.testing.assert_allclose(got[:, :S], want[lo:hi], rtol=1e-12, atol=1e-13 * scale)
            assert np.all(got[:, S] == -7.0)


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


@pytest.mark.parametrize("family,S", (("logistic", 128), ("linreg", 37), ("poisson", 600)))
def test_centring_folded_into_ingest_equals_the_centring_pass(bc, family, S):
    """projector.py:21 inside the solver's constructor pass (csrc/ingest.hip, BCX_LOAD_CENTER_ROWS): HilbertCoreset
    behind a DeviceProjector takes the RAW log-likelihoods and centres them while it reads them.  The solver must end up
    with the state it gets from the centred projection -- norms, b, trace, weights: bit for bit for S <= 512 on 16-byte
    rows (same lane -> column ownership and association as the former centring pass), to rounding otherwise (odd S: the
    scalar ingest kernel; S > 512) -- and project_uncentred() minus its row means is project()."""
    import torch
    Z, theta0, ll, sigsq = _cases()[family]
    theta = np.random.RandomState(S).randn(S, theta0.shape[1]) * (0.3 if family == "poisson" else 1.0)
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S, sigsq=sigsq)
    centred = prj.project(Z)
    raw = prj.project_uncentred(Z)
    torch.testing.assert_close(raw - raw.mean(dim=1, keepdim=True), centred, rtol=1e-12, atol=1e-12 * float(centred.abs().max()))
    assert float((raw.mean(dim=1).abs()).max()) > 1e-3          # (the raw rows really are uncentred)
    for cls in (bc.snnls.GIGA, bc.snnls.OrthoPursuit):
        two_pass = cls(centred.t(), None)
        folded = bc.HilbertCoreset(Z, prj, snnls=cls)
        assert folded.snnls._center_rows
        exact = S <= 512 and S % 2 == 0
        cmp = np.testing.assert_array_equal if exact else (lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-300))
        cmp(folded.snnls.Anorms, two_pass.Anorms)
        np.testing.assert_allclose(folded.snnls.b, two_pass.b, rtol=1e-10, atol=1e-10 * np.abs(two_pass.b).max())
        if exact:
            np.testing.assert_array_equal(folded.snnls.b, two_pass.b)
        two_pass.build(25)
        folded.build(25)
        assert np.array_equal(folded.snnls.last_trace[0], two_pass.last_trace[0])
        cmp(folded.snnls.weights(), two_pass.weights())
        # .A of the folded solver reads as the centred matrix
        torch.testing.assert_close(folded.snnls.A, centred.t(), rtol=1e-12, atol=1e-12 * float(centred.abs().max()))


def test_center_rows_flag_on_host_and_fp32_sources(bc):
    """BCX_LOAD_CENTER_ROWS through every upload path: host fp64 (in place in the resident raw copy), host fp32 (staged),
    fp64 storage without raw rows; all against the explicitly centred matrix."""
    rs = np.random.RandomState(12)
    V = rs.randn(5000, 24) * np.exp(rs.randn(5000, 1)) + 3.0 * rs.randn(5000, 1)      # large row means
    C = V - V.mean(axis=1)[:, None]
    for src, kw in ((V, {}), (V.astype(np.float32), {}), (V, {"dtype": "float64"}), (V, {"keep_exact_rows": False})):
        want = bc.snnls.FrankWolfe((src.astype(np.float64) - src.astype(np.float64).mean(axis=1)[:, None]).T, None, **kw)
        got = bc.snnls.FrankWolfe(src.T, None, center_rows=True, **kw)
        np.testing.assert_allclose(got.Anorms, want.Anorms, rtol=1e-6 if src.dtype == np.float32 else 1e-13)
        np.testing.assert_allclose(got.b, want.b, rtol=1e-6 if src.dtype == np.float32 else 1e-11,
                                   atol=(1e-5 if src.dtype == np.float32 else 1e-11) * np.abs(want.b).max())
        got.build(20)
        want.build(20)
        assert np.array_equal(got.last_trace[0], want.last_trace[0])
    np.testing.assert_allclose(np.asarray(bc.snnls.FrankWolfe(V.T, None, center_rows=True).A), C.T, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("kind", ("device", "device-mfma", "device-moments", "blackbox"))
def test_sparsevi_matches_reference(bc, kind):
    """F6: 5 greedy steps x 20 ADAM steps; same points in the same order, weights to 1e-5 -- with the column sums from the
    fused projection kernel ("device-mfma"), in closed form from the data's moments ("device-moments"), and in the default
    mode (moments after a one-time check against the projection)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "svi_golden.npz"))
    N, D, S, sigsq = int(g["N"]), int(g["D"]), int(g["S"]), float(g["sigsq"])
    Z = make_linreg_data(1, N, D)
    sampler = linreg_sampler(np.zeros(D), np.eye(D), sigsq)
    np.random.seed(2)
    if kind.startswith("device"):
        prj = bc.DeviceProjector("linreg", sampler, S, sigsq=sigsq, colsum={"device": "auto"}.get(kind, kind[7:]))
    else:
        prj = bc.BlackBoxProjector(sampler, S, lambda z, th: linreg_log_likelihood(z, th, sigsq))
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=int(g["opt_itrs"]))
    for i in range(int(g["steps"])):
        alg.build(1)
        assert np.array_equal(alg.idcs, g["step%d_idcs" % i]), "step %d picked a different point" % i
        np.testing.assert_allclose(alg.wts, g["step%d_wts" % i], rtol=1e-5, atol=1e-8)
    if kind == "device":
        assert prj.moments_info["checked"] and prj.moments_info["accepted"] and prj.moments_info["disagreement"] < 1e-10
    if kind == "device-mfma":
        assert not prj.moments_info
    wts, pts, idcs = alg.get()
    assert np.array_equal(idcs, g["get_idcs"])
    np.testing.assert_allclose(wts, g["get_wts"], rtol=1e-5)
    assert np.array_equal(pts, Z[idcs])


def test_device_sampler_feeds_projector_in_place(bc):
    """A sampler may hand back a GPU tensor (examples/common/model_linreg.py): same posterior as the NumPy form,
    and DeviceProjector uses the tensor without a host copy -- the fused consumers give the values they give for
    the same samples passed as an ndarray."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("model_linreg", os.path.join(
        os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd", "examples", "common", "model_linreg.py"))
    ml = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ml)
    rs = np.random.RandomState(3)
    D, S, N = 24, 64, 5000
    mu0, Sig0, sigsq = rs.randn(D), np.eye(D) * 2.0, 0.7
    pts = rs.randn(6, D + 1)
    wts = rs.rand(6) * 5
    dev_sampler = ml.posterior_sampler(mu0, Sig0, sigsq, device="cuda", seed=5)
    mu_d, U_d = dev_sampler.posterior(wts, pts)
    mu_h, U_h = ml.weighted_posterior(mu0, np.linalg.inv(Sig0), sigsq, pts, wts)
    np.testing.assert_allclose(mu_d.cpu().numpy(), mu_h, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose((U_d @ U_d.T).cpu().numpy(), U_h.dot(U_h.T), rtol=1e-10, atol=1e-12)
    prior = dev_sampler.posterior(None, None)[0].cpu().numpy()
    np.testing.assert_allclose(prior, mu0, rtol=1e-10, atol=1e-12)
    Z = rs.randn(N, D + 1)
    prj = bc.DeviceProjector("linreg", dev_sampler, S, sigsq=sigsq)
    prj.update(wts, pts)
    assert isinstance(prj.samples, torch.Tensor) and prj.samples.is_cuda and prj.samples.shape == (S, D)
    theta = prj.samples.cpu().numpy()
    host = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=sigsq)
    assert np.array_equal(prj.project_colsum(Z), host.project_colsum(Z))
    resid = rs.randn(S)
    assert prj.project_select(Z, resid) == host.project_select(Z, resid)
    # draws have the posterior's first two moments
    big = dev_sampler(20000, wts, pts).cpu().numpy()
    np.testing.assert_allclose(big.mean(axis=0), mu_h, atol=6 * np.sqrt(np.diag(U_h.dot(U_h.T)).max() / 20000))


@pytest.mark.parametrize("family", ("logistic", "poisson"))
def test_project_extreme_arguments(bc, family):
    """The piecewise branches of the example likelihoods (model_lr.py:29-31: linear tail for -m >= 100;
    model_poiss.py:25-30: softplus passthrough below -100): features scaled so that z.theta spans +-400."""
    rs = np.random.RandomState(17)
    D, S, N = 6, 48, 3001
    if family == "logistic":
        Z = rs.randn(N, D) * rs.choice([0.1, 5.0, 60.0], size=(N, 1))
        ll = logistic_log_likelihood
    else:
        X = rs.randn(N, D) * rs.choice([0.1, 5.0, 60.0], size=(N, 1))
        Z = np.hstack((X, rs.poisson(2.0, size=(N, 1)).astype(np.float64)))
        ll = poisson_log_likelihood
    theta = rs.randn(S, D)
    m = Z[:, :D].dot(theta.T)
    assert m.min() < -150 and m.max() > 150
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S)
    ref = bc.BlackBoxProjector(lambda n, w, p: theta, S, ll)
    want = ref.project(Z)
    got = prj.project(Z).cpu().numpy()
    assert np.isfinite(want).all() and np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 * np.abs(want).max())


def test_poisson_responses_outside_the_count_table(bc):
    """gammaln(y + 1) of model_poiss.py:37: counts 0 .. 255 come from the kernel's log-factorial table, every other
    response (large counts, non-integers) from the library routine.  Uncentred values, so the constant is visible; the
    fused consumers (column sums, select) drop it analytically and have to agree with the centred reference."""
    rs = np.random.RandomState(23)
    D, S, N = 9, 130, 2600
    X = 0.4 * rs.randn(N, D)
    y = rs.poisson(3.0, size=N).astype(np.float64)
    y[::7] = rs.choice([255.0, 256.0, 300.0, 1000.0, 1e5, 2.5, 0.25, 17.75, 254.0], size=y[::7].shape)
    Z = np.hstack((X, y[:, None]))
    theta = 0.5 * rs.randn(S, D)
    prj = bc.DeviceProjector("poisson", lambda n, w, p: theta, S)
    want = poisson_log_likelihood(Z, theta)
    got = prj.project_uncentred(Z).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12)
    centred = want - want.mean(axis=1)[:, None]
    np.testing.assert_allclose(prj.project(Z).cpu().numpy(), centred, rtol=1e-9, atol=1e-11 * np.abs(want).max())
    resid = rs.randn(S)
    corrs, best = _reference_select(centred, resid)
    val, idx = prj.project_select(Z, resid)
    assert int(idx) == best
    np.testing.assert_allclose(float(val), corrs[best], rtol=1e-9)
    np.testing.assert_allclose(prj.project_colsum(Z), centred.sum(axis=0), rtol=1e-9, atol=1e-10 * np.abs(centred).sum(axis=0).max())


@pytest.mark.parametrize("family", ("logistic", "poisson", "linreg"))
def test_nan_feature_gives_a_nan_row_as_in_the_reference(bc, family):
    """A NaN among a point's features makes its whole log-likelihood row NaN in the reference (np.log1p / np.exp
    propagate it); the table-driven epilogues pick table entries from bit patterns, so this pins that a NaN neither
    disappears (max(NaN, 0) = 0 would) nor disturbs the other rows."""
    rs = np.random.RandomState(5)
    D, S, N = 7, 70, 700
    X = rs.randn(N, D)
    if family == "logistic":
        Z, ll = X.copy(), logistic_log_likelihood
    elif family == "poisson":
        Z, ll = np.hstack((X, rs.poisson(2.0, size=(N, 1)).astype(np.float64))), poisson_log_likelihood
    else:
        Z, ll = np.hstack((X, rs.randn(N, 1))), (lambda z, th: linreg_log_likelihood(z, th, 1.3))
    Z[41, 2] = np.nan
    Z[300, 0] = -np.nan
    theta = rs.randn(S, D)
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S, **({"sigsq": 1.3} if family == "linreg" else {}))
    with np.errstate(invalid="ignore"):
        want = ll(Z, theta)
    got = prj.project_uncentred(Z).cpu().numpy()
    assert np.isnan(want[41]).all() and np.isnan(want[300]).all()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-11, atol=1e-12)


# ---- numerically hard rows for the fused SELECT / COLSUM consumers ---------------------------------------------------
def _reference_select(vecs, resid):
    """sparsevi.py:49-55 on centred vectors: corrs, first arg-max (NaN counts as the maximum, as in NumPy)."""
    with np.errstate(invalid="ignore", divide="ignore"):
        corrs = vecs.dot(resid) / np.sqrt((vecs ** 2).sum(axis=1)) / vecs.shape[1]
    return corrs, int(np.argmax(corrs))


@pytest.mark.parametrize("N", (20000, 70000))
def test_select_rows_with_huge_mean_and_tiny_spread(bc, N):
    """|mean| >> spread: a concentrated posterior (samples theta_0 + 1e-7 noise) and large responses give log-likelihoods
    ~ -1e9 whose variation across samples is ~1e-3.  The one-pass moments s2 - S mean^2 of the raw values would lose
    every digit (round 1); the shifted accumulation must name the reference's row and value.  N = 70000 fills the grid:
    the two column groups of a row are then handled by different workgroups and merged from moments about two shifts."""
    rs = np.random.RandomState(31)
    D, S, sigsq = 12, 96, 1e-4
    X = rs.randn(N, D)
    th0 = rs.randn(D)
    y = X.dot(th0) + 400.0 + rs.randn(N)          # residual ~400 => ll ~ -400^2 / 2e-4 = -8e8
    Z = np.hstack((X, y[:, None]))
    theta = th0 + 1e-7 * rs.randn(S, D)
    ll = linreg_log_likelihood(Z, theta, sigsq)
    assert np.abs(ll).min() > 1e8 and (ll.max(axis=1) - ll.min(axis=1)).max() < 100.0    # |mean| / spread > 1e6
    vecs = ll - ll.mean(axis=1)[:, None]
    resid = rs.randn(S)
    corrs, want = _reference_select(vecs, resid)
    prj = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=sigsq)
    best, row = prj.project_select(Z, resid)
    assert row == want
    # the centred values carry |ll| * eps ~ 1e-7 absolute noise on a spread of ~1e-1 in the reference itself
    np.testing.assert_allclose(best, corrs[want], rtol=1e-4)
    got_col = prj.project_colsum(Z)
    np.testing.assert_allclose(got_col, vecs.sum(axis=0), rtol=1e-6, atol=1e-6 * np.abs(vecs).sum() / S)


@pytest.mark.parametrize("N", (30000, 70000))
def test_select_logistic_rows_over_sixteen_decades(bc, N):
    """Saturated logistic rows: projected-vector norms from ~1e-16 (log-likelihood ~ -exp(-m), m ~ 37) to O(1), the
    range real Laplace-projected vectors show.  Tiny-norm rows must neither win through a
    cancelled norm nor turn into NaN.  (N = 70000: on XCD teams, see above.)"""
    rs = np.random.RandomState(33)
    D, S = 8, 128
    scale = 10.0 ** rs.uniform(-1.0, 1.62, size=N)               # |z.theta| from ~0.1 to ~42
    Zdir = rs.randn(N, D)
    th0 = rs.randn(D)
    Zdir *= np.sign(Zdir.dot(th0))[:, None]                       # correctly classified: m > 0, saturating as it grows
    Z = Zdir * (scale / np.abs(Zdir.dot(th0)))[:, None]
    theta = th0 + 0.02 * rs.randn(S, D)
    ll = logistic_log_likelihood(Z, theta)
    vecs = ll - ll.mean(axis=1)[:, None]
    norms = np.sqrt((vecs ** 2).sum(axis=1))
    assert norms.min() < 1e-14 and norms.max() > 1e-2 and norms.min() > 0
    resid = rs.randn(S)
    corrs, want = _reference_select(vecs, resid)
    prj = bc.DeviceProjector("logistic", lambda n, w, p: theta, S)
    best, row = prj.project_select(Z, resid)
    top2 = np.sort(corrs)[-2:]
    assert top2[1] - top2[0] > 1e-9, "test input has a near-tie"
    assert row == want
    np.testing.assert_allclose(best, corrs[want], rtol=1e-6)
    # every tiny-norm row on its own: the device's value equals the reference's to the accuracy the reference has
    tiny = np.argsort(norms)[:64]
    b2, r2 = prj.project_select(Z[tiny], resid)
    c2, w2 = _reference_select(vecs[tiny], resid)
    assert np.isfinite(b2) and abs(b2) <= 1.0 / np.sqrt(S) * np.abs(resid).sum()


@pytest.mark.parametrize("N,S", ((5000, 64), (70000, 128)))
def test_select_zero_vector_is_numpys_nan_pick(bc, N, S):
    """A data row whose log-likelihood is the same for every sample projects to the zero vector: corrs is 0/0 = NaN
    there and ``corrs.argmax()`` returns the first NaN (sparsevi.py:51-55).  Same pick on the device -- also when the row's
    two column groups are merged from two workgroups' partial moments (N = 70000, S = 128: a power of two keeps the
    reference's own row mean exact)."""
    rs = np.random.RandomState(35)
    D = 6
    Z = np.hstack((rs.randn(N, D), rs.randn(N, 1)))
    Z[1234, :D] = 0.0                                             # x = 0: the likelihood does not depend on theta
    Z[4000, :D] = 0.0
    theta = rs.randn(S, D)
    vecs = linreg_log_likelihood(Z, theta, 1.0)
    vecs = vecs - vecs.mean(axis=1)[:, None]
    resid = rs.randn(S)
    if N == 5000:
        assert np.all(vecs[1234] == 0.0)
        corrs, want = _reference_select(vecs, resid)
        assert want == 1234 and np.isnan(corrs[want])
    else:
        # (NumPy's own row mean of 128 equal values may be one ulp off the value, which turns its 0/0 into noise; the
        #  device forms the moments about a value of the row itself, so a constant row is exactly zero there)
        assert np.abs(vecs[1234]).max() < 1e-15
    prj = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=1.0)
    best, row = prj.project_select(Z, resid)
    assert row == 1234 and np.isnan(best)


# ---- config-5 workload as specified: RBF-basis regression, reference fixture F6b --------------
def _rbf():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rbf_workload", os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common", "rbf_workload.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_rbf_workload_one_state_against_reference(bc):
    """F6b, one state: prior samples on the collinear 301-column design (|ll| ~ 1e7): device projection, column sums
    and the correlation arg-max against the reference's own numbers."""
    import hashlib
    g = np.load(os.path.join(ROOT, "tests", "golden", "rbf_golden.npz"))
    wl = _rbf().make_rbf_regression(int(g["N"]), int(g["nb"]), seed=1)
    Z = wl["Z"]
    assert hashlib.sha256(Z.tobytes()).hexdigest() == str(g["Z_sha"]), "the workload generator drifted from the fixture"
    theta0 = g["theta0"]
    prj = bc.DeviceProjector("linreg", lambda n, w, p: theta0, int(g["S"]), sigsq=float(g["sigsq"]))
    col = prj.project_colsum(Z)
    np.testing.assert_allclose(col, g["colsum0"], rtol=1e-9, atol=1e-9 * np.abs(g["colsum0"]).max())
    best, row = prj.project_select(Z, g["colsum0"])
    assert row == int(g["corr_argmax0"])
    np.testing.assert_allclose(best, float(g["corr_max0"]), rtol=1e-9)
    vecs = prj.project(Z[:4096]).cpu().numpy()
    corr_head = vecs.dot(g["colsum0"]) / np.sqrt((vecs ** 2).sum(axis=1)) / vecs.shape[1]
    np.testing.assert_allclose(corr_head, g["corr_head0"], rtol=1e-9)


@pytest.mark.parametrize("kind", ("device", "device-mfma", "device-moments", "blackbox"))
def test_sparsevi_rbf_matches_reference(bc, kind):
    """F6b: the reference's SparseVI run on the RBF regression (N = 50k, D = 301, S = 64, 20 ADAM steps per greedy step):
    same points in the same order, weights to 1e-5."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "rbf_golden.npz"))
    wl = _rbf().make_rbf_regression(int(g["N"]), int(g["nb"]), seed=1)
    Z, sigsq, S = wl["Z"], wl["sigsq"], int(g["S"])
    sampler = linreg_sampler(wl["mu0"], wl["Sig0"], sigsq)
    np.random.seed(2)
    if kind.startswith("device"):
        prj = bc.DeviceProjector("linreg", sampler, S, sigsq=sigsq, colsum={"device": "auto"}.get(kind, kind[7:]))
    else:
        prj = bc.BlackBoxProjector(sampler, S, lambda z, th: linreg_log_likelihood(z, th, sigsq))
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=int(g["opt_itrs"]))
    for i in range(int(g["steps"])):
        alg.build(1)
        assert np.array_equal(alg.idcs, g["step%d_idcs" % i]), "step %d picked a different point" % i
        np.testing.assert_allclose(alg.wts, g["step%d_wts" % i], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("N,C,pad", ((4096, 2, 0), (5000, 22, 1), (33333, 65, 3), (20011, 302, 0), (4100, 130, 2), (1, 5, 0), (31, 64, 0), (0, 3, 1)))
def test_moments_kernel_matches_numpy(bc, N, C, pad):
    """csrc/moments.hip: M = Z^T Z (fp64 MFMA, block pairs of the upper triangle x row slices) against NumPy, for column
    counts around the 64-wide block edges, padded leading dimensions and row counts that leave ragged slices."""
    import torch
    from bayesiancoresets_amd import _native as nat
    lib = nat.load()
    rs = np.random.RandomState(N + C)
    Z = rs.randn(N, C) * np.exp(rs.randn(C))[None, :]
    buf = torch.zeros(max(N, 1), C + pad, dtype=torch.float64, device="cuda")
    buf[:N, :C] = torch.from_numpy(Z).cuda()
    need = lib.bcx_project_moments_scratch_bytes(N, C)
    assert need > 0 and need % 8 == 0
    work = torch.empty(need // 8, dtype=torch.float64, device="cuda")
    M = torch.full((C, C + 1), float("nan"), dtype=torch.float64, device="cuda")
    rc = lib.bcx_project_moments(int(torch.cuda.current_stream().cuda_stream), buf.data_ptr(), N, C + pad, C, M.data_ptr(), C + 1,
                                 work.data_ptr(), need)
    assert rc == 0, lib.bcx_project_last_error()
    got = M[:, :C].cpu().numpy()
    want = Z.T.dot(Z)
    assert np.array_equal(got, got.T)
    bound = 1e-12 * np.sqrt(np.outer(np.diag(want), np.diag(want)))       # |M_ij| <= sqrt(M_ii M_jj)
    assert (np.abs(got - want) <= bound).all(), float((np.abs(got - want) / bound).max())
    assert torch.isnan(M[:, C]).all()                         # nothing written outside the C x C block
    # same bits on a second run (fixed summation order)
    M2 = torch.empty_like(M)
    assert lib.bcx_project_moments(int(torch.cuda.current_stream().cuda_stream), buf.data_ptr(), N, C + pad, C, M2.data_ptr(), C + 1,
                                   work.data_ptr(), need) == 0
    assert torch.equal(M[:, :C], M2[:, :C])
    # argument errors come back as codes, not crashes
    assert lib.bcx_project_moments(0, buf.data_ptr(), N, C + pad, C, M.data_ptr(), C + 1, work.data_ptr(), need - 8) != 0
    assert lib.bcx_project_moments_scratch_bytes(N, 2000) == -1


@pytest.mark.parametrize("D,S", ((30, 130), (301, 256), (7, 1), (64, 37)))
def test_colsum_from_moments_equals_projected_colsum(bc, D, S):
    """project_colsum of the linear-regression family: the closed form on the data's moments against the fused projection
    kernel (rtol 1e-10 of the largest column sum) and against NumPy."""
    rs = np.random.RandomState(D * 1000 + S)
    N = 50_000
    Z = make_linreg_data(5, N, D)
    theta = 0.7 + 1.3 * rs.randn(S, D)
    a = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=0.7, colsum="mfma")
    b = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=0.7, colsum="moments")
    ca, cb = a.project_colsum(Z), b.project_colsum(Z)
    assert b.moments_info["rows"] == N and not a.moments_info
    scale = np.abs(ca).max() if S > 1 else 1.0
    np.testing.assert_allclose(cb, ca, rtol=1e-10, atol=1e-10 * scale)
    want = linreg_log_likelihood(Z, theta, 0.7)
    want -= want.mean(axis=1)[:, None]
    np.testing.assert_allclose(cb, want.sum(axis=0), rtol=1e-9, atol=1e-10 * scale)
    # new samples, same data object: the moments are reused
    theta2 = theta + 0.1 * rs.randn(S, D)
    b.sampler = a.sampler = lambda n, w, p: theta2
    a.update(np.array([]), np.array([])); b.update(np.array([]), np.array([]))
    info = dict(b.moments_info)
    np.testing.assert_allclose(b.project_colsum(Z), a.project_colsum(Z), rtol=1e-10, atol=1e-10 * scale)
    assert b.moments_info == info
    # another data object: new moments; a small one: the projection kernel
    Z2 = Z[:9000].copy()
    np.testing.assert_allclose(b.project_colsum(Z2), a.project_colsum(Z2), rtol=1e-10, atol=1e-10 * scale)
    assert b.moments_info["rows"] == 9000
    np.testing.assert_allclose(b.project_colsum(Z[:100].copy()), a.project_colsum(Z[:100].copy()), rtol=1e-12, atol=1e-12 * scale)
    assert b.moments_info["rows"] == 9000


@pytest.mark.parametrize("D,S,k", ((301, 256, 300), (30, 130, 65), (24, 64, 33), (301, 256, 20), (64, 37, 700)))
def test_points_and_closed_form_column_sums_in_one_launch(bc, D, S, k):
    """bcx_project_points_colsum_moments (csrc/proj.hip proj_mid_quad_kernel: the two projections of a SparseVI ADAM step, which
    both only read the draws, as ONE launch) against the two calls it stands for, bit for bit -- for point counts on both
    sides of the 32 x 32-block kernel's range (k = 20: the two calls inside) -- and against NumPy."""
    import torch
    from bayesiancoresets_amd import _native
    lib = _native.load()
    rs = np.random.RandomState(D * 7 + S + k)
    N = 20_000
    Z = make_linreg_data(3, N, D)
    theta = 0.4 + 0.9 * rs.randn(S, D)
    sigsq = 0.8
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    ld = D + D % 2
    th = torch.zeros(S, ld, dtype=torch.float64, device="cuda")
    th[:, :D] = dev(theta)
    Zd, Cd = dev(Z), dev(Z[rs.choice(N, k, replace=False)])
    tbar = th[:, :D].mean(dim=0).contiguous()
    st = int(torch.cuda.current_stream().cuda_stream)
    C = D + 1
    M = torch.empty(C, C, dtype=torch.float64, device="cuda")
    need = int(lib.bcx_project_moments_scratch_bytes(N, C))
    w0 = torch.empty(max(need // 8, 1), dtype=torch.float64, device="cuda")
    assert lib.bcx_project_moments(st, Zd.data_ptr(), N, C, C, M.data_ptr(), C, w0.data_ptr(), need) == 0
    qn = int(lib.bcx_project_colsum_moments_scratch_bytes(D, S)) // 8
    outs = []
    for merged in (False, True):
        work = torch.zeros(qn, dtype=torch.float64, device="cuda")
        col = torch.full((S,), np.nan, dtype=torch.float64, device="cuda")
        out = torch.full((k, S), np.nan, dtype=torch.float64, device="cuda")
        for _ in range(2):                          # (twice: the closing workgroup's counter must be left at zero)
            if merged:
                assert lib.bcx_project_points_colsum_moments(st, Cd.data_ptr(), k, C, D, D, th.data_ptr(), S, ld, sigsq, out.data_ptr(), S,
                                                             M.data_ptr(), C, D, col.data_ptr(), work.data_ptr(), tbar.data_ptr()) == 0
            else:
                assert lib.bcx_project_colsum_moments_at(st, M.data_ptr(), C, D, D, th.data_ptr(), S, ld, sigsq, col.data_ptr(),
                                                         work.data_ptr(), tbar.data_ptr()) == 0
                assert lib.bcx_project_write_points(st, 2, Cd.data_ptr(), k, C, D, D, th.data_ptr(), S, ld, sigsq, out.data_ptr(), S, 0) == 0
        outs.append((col.cpu().numpy(), out.cpu().numpy()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    want = linreg_log_likelihood(Z, theta, sigsq)
    want -= want.mean(axis=1)[:, None]
    scale = np.abs(want.sum(axis=0)).max()
    np.testing.assert_allclose(outs[1][0], want.sum(axis=0), rtol=1e-9, atol=1e-10 * scale)
    np.testing.assert_allclose(outs[1][1], linreg_log_likelihood(Cd.cpu().numpy(), theta, sigsq), rtol=1e-10, atol=1e-9)
    # bad arguments come back as codes
    assert lib.bcx_project_points_colsum_moments(st, Cd.data_ptr(), 300, C, D, D, th.data_ptr(), S, ld, sigsq, None, S,
                                                 M.data_ptr(), C, D, col.data_ptr(), work.data_ptr(), tbar.data_ptr()) != 0


def test_rbf_shard_size_select_and_colsum(bc):
    """BASELINE configs[4] per-GPU shard (N = 625k, D = 301, S = 256) on the RBF design: project_select / project_colsum
    against a torch fp64 centre-then-norm restatement of sparsevi.py:47-56 over all rows (chunked), at a prior state
    (|ll| ~ 1e7, spread ~ 1e7) and at a concentrated state (samples within 1e-3 of the least-squares fit: |mean| >>
    spread on most rows)."""
    import torch
    rbf = _rbf()
    N, nb, S = 625_000, 50, 256
    rs = np.random.RandomState(1)
    obs = rbf.synthetic_observations(N, rs)
    scales, centres = rbf.basis_layout(obs, nb, rs)
    Zd = rbf.design_rows_device(torch, torch.from_numpy(obs).cuda(), scales, centres)
    assert Zd.shape == (N, 302)
    std, mean = obs[:, 2].std(), obs[:, 2].mean()
    sigsq = float(std ** 2)
    D = 301
    # ridge fit on a subsample: the centre of the "concentrated" state
    sub = Zd[::50]
    A = sub[:, :D].T @ sub[:, :D] + 1e-6 * torch.eye(D, dtype=torch.float64, device="cuda")
    fit = torch.linalg.solve(A, sub[:, :D].T @ sub[:, D]).cpu().numpy()
    states = {"prior": mean + np.sqrt(std ** 2 + mean ** 2) * rs.randn(S, D), "concentrated": fit + 1e-3 * rs.randn(S, D)}
    for name, theta in states.items():
        th = torch.from_numpy(theta).cuda()
        resid_h = rs.randn(S)
        resid = torch.from_numpy(resid_h).cuda()
        clin = -0.5 * np.log(2.0 * np.pi * sigsq)
        colsum = torch.zeros(S, dtype=torch.float64, device="cuda")
        bestv, besti = -np.inf, -1
        second = -np.inf
        for r in range(0, N, 1 << 16):
            z = Zd[r:r + (1 << 16)]
            m = z[:, :D] @ th.T
            y = z[:, D:D + 1]
            ll = clin - (y * y - 2.0 * m * y + m * m) / (2.0 * sigsq)
            v = ll - ll.mean(dim=1, keepdim=True)
            colsum += v.sum(dim=0)
            corr = (v @ resid) / torch.sqrt((v * v).sum(dim=1)) / S
            top = torch.topk(corr, 2)
            for val, idx in zip(top.values.tolist(), top.indices.tolist()):
                if val > bestv:
                    second, bestv, besti = bestv, val, r + idx
                elif val > second:
                    second = val
        prj = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=sigsq, colsum="mfma")
        got = prj.project_colsum(Zd)
        np.testing.assert_allclose(got, colsum.cpu().numpy(), rtol=1e-7, atol=1e-9 * float(colsum.abs().max()), err_msg=name)
        # the closed form on the moments: at the prior state to 1e-10 of the projection kernel's sums; at the concentrated one
        # (v = G thetabar - g cancels to ~1e-9 of its terms next to the least-squares fit) to the reference's own accuracy
        mom = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=sigsq, colsum="moments").project_colsum(Zd)
        tol = 1e-10 if name == "prior" else 1e-6
        np.testing.assert_allclose(mom, got, rtol=tol, atol=tol * float(np.abs(got).max()), err_msg=name)
        best, row = prj.project_select(Zd, resid_h)
        assert bestv - second > 1e-7 * abs(bestv), "near-tie in the test input (%s)" % name
        assert row == besti, name
        np.testing.assert_allclose(best, bestv, rtol=1e-6, err_msg=name)


@pytest.mark.parametrize("alg", ("SVI", "GIGA-OPT", "US", "SVI-EXACT", "GIGA-OPT-EXACT", "GIGA-REAL-EXACT"))
def test_linear_regression_example_cli(tmp_path, alg):
    """examples/linear_regression/main.py (the harness of BASELINE configs[4], linear_regression/main.py:29-259 on the
    synthetic observations): runs end to end on the device projector and stores the reference's result columns; a
    coreset of growing size brings the weighted posterior towards the full-data posterior."""
    import subprocess
    import sys
    import pandas as pd
    script = os.path.join(ROOT, "bayesian-coresets_amd", "examples", "linear_regression", "main.py")
    folder = str(tmp_path / "results") + "/"
    # (the exact tangent-space projectors: rows of D + proj_dim^2 = 19 + 81 numbers)
    cmd = [sys.executable, script, "--alg", alg, "--trial", "1", "--data_num", "4000", "--n_bases_per_scale", "3",
           "--proj_dim", "9" if alg.endswith("EXACT") else "48",
           "--coreset_size_max", "12", "--coreset_num_sizes", "4", "--opt_itrs", "10", "--results_folder", folder, "run"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    files = [f for f in os.listdir(folder) if f != "manifest.csv"]
    assert len(files) == 1
    t = pd.read_csv(os.path.join(folder, files[0]))
    for col in ("csizes", "Ms", "cputs", "rklw", "fklw", "mu_errs", "Sig_errs"):
        assert col in t.columns, col
    assert t["Ms"].iloc[0] == 0 and t["csizes"].iloc[0] == 0          # the first recorded size is the empty coreset
    assert np.isfinite(t["rklw"]).all() and np.isfinite(t["fklw"]).all()
    assert t["csizes"].iloc[-1] >= 1
    if alg not in ("US", "GIGA-REAL-EXACT"):      # (the poorly tuned tangent space at a sqrt(N)-point posterior promises nothing at 12 points)
        assert t["fklw"].iloc[-1] < t["fklw"].iloc[0]
    again = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert again.returncode == 0 and "Results already exist" in again.stdout


@pytest.mark.parametrize("family", ("logistic", "poisson", "linreg"))
@pytest.mark.parametrize("N,D,S", ((1, 1, 1), (2, 3, 5), (127, 2, 16), (129, 16, 64), (257, 17, 65), (300, 33, 130), (1031, 48, 200),
                                   (140, 20, 100), (515, 40, 256), (700, 24, 1100),
                                   (257, 64, 65), (300, 81, 200), (515, 72, 256), (260, 64, 1100), (131, 67, 128)))
def test_projection_tile_edges(bc, family, N, D, S):
    """Shapes around the kernel's tile sizes (128 rows x 64 or -- linreg column sums when S pads to a multiple of 128: S =
    65, 100, 200, 256, 1100 here -- 128 columns x 16 features per stage; S = 1100 also takes the column-sum accumulators past
    the LDS budget of two workgroups per CU; project() takes the 128-column tile for rows of 64 features and more), odd and even leading
    dimensions (16-byte and 8-byte operand loads), single row / column / feature: values, column sums and the
    correlation arg-max against NumPy for every consumer."""
    rs = np.random.RandomState(N * 1000 + D * 10 + S)
    X = rs.randn(N, D) * 0.7
    theta = rs.randn(S, D) * 0.5
    if family == "logistic":
        Z, ll = X, logistic_log_likelihood
    elif family == "poisson":
        Z, ll = np.hstack((X, rs.poisson(2.0, size=(N, 1)).astype(np.float64))), poisson_log_likelihood
    else:
        Z, ll = np.hstack((X, rs.randn(N, 1))), (lambda z, th: linreg_log_likelihood(z, th, 0.9))
    want = ll(Z.copy(), theta)
    want = want - want.mean(axis=1)[:, None]
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S, sigsq=0.9)
    got = prj.project(Z).cpu().numpy()
    scale = max(np.abs(want).max(), 1e-300)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-11 * scale)
    np.testing.assert_allclose(prj.project_colsum(Z), want.sum(axis=0), rtol=1e-8, atol=1e-10 * max(np.abs(want).sum(), 1e-300))
    if S > 1:
        resid = rs.randn(S)
        corrs, arg = _reference_select(want, resid)
        best, row = prj.project_select(Z, resid)
        order = np.sort(corrs[np.isfinite(corrs)])
        if len(order) >= 2 and order[-1] - order[-2] > 1e-9:
            assert row == arg
            np.testing.assert_allclose(best, corrs[arg], rtol=1e-7)


def test_simple_lr_example_cli():
    """examples/simple_lr/main.py: data, Laplace fit on the device, device projection, greedy coreset -- end to end."""
    import subprocess
    import sys
    script = os.path.join(ROOT, "bayesian-coresets_amd", "examples", "simple_lr", "main.py")
    out = subprocess.run([sys.executable, script, "--rows", "20000", "--samples", "64", "--alg", "GIGA", "--size", "20"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "coreset size" in out.stdout and "error" in out.stdout


def test_blackbox_sparsevi_survives_a_zero_projected_vector(bc):
    """Round-1 ADVICE: SparseVI with a user-callback projector reuses one engine across steps; a zero projected vector
    (here: a data row with x = 0, whose likelihood does not depend on theta) must only affect the arithmetic the way it
    does in the reference -- corrs is NaN there, `corrs.argmax()` picks it, and it enters an EMPTY coreset only
    (sparsevi.py:49-60) -- and must not leave the engine unusable for the next step."""
    from oracle.sparsevi_oracle import SparseVIOracle, linreg_loglik
    N, D, S, sigsq = 6000, 8, 32, 1.0
    Z = make_linreg_data(7, N, D)
    Z[4321, :D] = 0.0
    sampler = linreg_sampler(np.zeros(D), np.eye(D), sigsq)
    np.random.seed(3)
    prj = bc.BlackBoxProjector(sampler, S, lambda z, th: linreg_log_likelihood(z, th, sigsq))
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=4)
    np.random.seed(3)
    o = SparseVIOracle(Z, sampler, lambda z, th: linreg_loglik(z, th, sigsq), S, opt_itrs=4)
    import warnings
    for step in range(4):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")              # the oracle divides 0 by 0 on purpose (as the reference does)
            with np.errstate(all="ignore"):
                o.step()
        alg.build(1)
        assert np.array_equal(alg.idcs, o.idcs), (step, alg.idcs, o.idcs)
        np.testing.assert_allclose(alg.wts, o.wts, rtol=1e-6, atol=1e-9)
    assert alg.idcs[0] == 4321 and len(alg.idcs) == 1      # the NaN pick entered the empty coreset; later NaN > x is False


def test_host_solver_behind_device_projector(bc):
    """hilbert.py:24-25 with a host-side solver class (sampling baseline) and a projector that returns a GPU tensor: the
    vectors are brought to the host for that solver (round-1 ADVICE: np.sqrt / A.dot on a CUDA tensor used to crash)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "lr_golden.npz"))
    Z = make_lr_data(1, 2000, int(g["D"]))
    samples = g["samples"][:16]
    dev = bc.DeviceProjector("logistic", lambda n, w, p: samples, 16)
    np.random.seed(5)
    c = bc.HilbertCoreset(Z, dev, snnls=bc.snnls.UniformSampling)
    c.build(30)
    wts, pts, idcs = c.get()
    assert isinstance(c.snnls, bc.snnls.SparseNNLS) and len(wts) > 0 and np.array_equal(pts, Z[idcs])
    host = bc.BlackBoxProjector(lambda n, w, p: samples, 16, logistic_log_likelihood)
    np.random.seed(5)
    c2 = bc.HilbertCoreset(Z, host, snnls=bc.snnls.UniformSampling)
    c2.build(30)
    assert np.array_equal(c2.get()[2], idcs)
    np.testing.assert_allclose(c2.error(), c.error(), rtol=1e-9)


def test_colsum_tile_and_team_variants_agree():
    """The column sums (and the correlation arg-max) come out of three block -> tile arrangements (128-column tile on XCD teams: the default for this
    shape; 64-column tile on teams; 64-column tile with every workgroup walking its own column groups: the fallback for
    grids that do not cover the XCDs evenly).  They differ only in summation order: same result to rounding, and each
    against NumPy.  One subprocess per arrangement (the knobs are read once per process)."""
    import json
    import subprocess
    import sys
    code = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%r, "bayesian-coresets_amd"))
sys.path.insert(0, %r)
import bayesiancoresets_amd as bc
rs = np.random.RandomState(5)
N, D, S = 70000, 40, 256                      # 547 row blocks: the full 512-workgroup grid
Z = np.hstack((rs.randn(N, D) * 0.6, rs.randn(N, 1)))
theta = rs.randn(S, D) * 0.4
prj = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=0.8)
best, row = prj.project_select(Z, np.random.RandomState(6).randn(S))
print(json.dumps({"colsum": [float(v) for v in prj.project_colsum(Z)], "select": [float(best), int(row)]}))
''' % (ROOT, ROOT)
    outs, sels = [], []
    for env in ({}, {"BCX_PROJ_NCT": "4"}, {"BCX_PROJ_NCT": "4", "BCX_PROJ_NO_TEAM": "1"}):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode == 0, r.stderr[-2000:]
        rec = json.loads(r.stdout.strip().splitlines()[-1])
        outs.append(np.array(rec["colsum"]))
        sels.append(rec["select"])
    rs = np.random.RandomState(5)
    N, D, S = 70000, 40, 256
    Z = np.hstack((rs.randn(N, D) * 0.6, rs.randn(N, 1)))
    theta = rs.randn(S, D) * 0.4
    ll = linreg_log_likelihood(Z, theta, 0.8)
    want = (ll - ll.mean(axis=1)[:, None]).sum(axis=0)
    scale = np.abs(want).max()
    for got in outs:
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11 * scale)
    np.testing.assert_allclose(outs[1], outs[0], rtol=1e-10, atol=1e-12 * scale)
    np.testing.assert_allclose(outs[2], outs[0], rtol=1e-10, atol=1e-12 * scale)
    # the correlation arg-max: on teams (partial row moments about per-group shifts, merged by select_combine_kernel) and
    # with the one-workgroup walk, against NumPy
    vecs = ll - ll.mean(axis=1)[:, None]
    corrs, arg = _reference_select(vecs, np.random.RandomState(6).randn(S))
    for best, row in sels:
        assert row == arg
        np.testing.assert_allclose(best, corrs[arg], rtol=1e-9)
    assert sels[2][1] == sels[0][1] and abs(sels[2][0] - sels[0][0]) <= 1e-12 * abs(sels[0][0])


@pytest.mark.parametrize("family", ("logistic", "poisson", "linreg"))
@pytest.mark.parametrize("N,D,S", ((1, 301, 256), (7, 30, 37), (32, 16, 1000), (33, 16, 1000), (5, 301, 1500)))
def test_few_rows_take_the_row_per_workgroup_kernel(bc, family, N, D, S):
    """project() / project_uncentred() of a handful of rows (csrc/proj.hip proj_small_kernel: N <= 32, S <= 1024; one row and
    one sample beyond, and S beyond, take the tiled kernel) against NumPy, with even and odd parameter row lengths."""
    rs = np.random.RandomState(N * 1000 + D + S)
    theta = 0.3 * rs.randn(S, D)
    if family == "logistic":
        Z, ll = 1.5 * rs.randn(N, D), logistic_log_likelihood
    elif family == "poisson":
        Z = np.hstack((rs.randn(N, D - 1), np.ones((N, 1)), rs.poisson(2.0, size=(N, 1)).astype(np.float64)))
        ll = poisson_log_likelihood
    else:
        Z = np.hstack((rs.randn(N, D), rs.randn(N, 1)))
        ll = lambda z, th: linreg_log_likelihood(z, th, 0.9)
    want = ll(Z, theta)
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S, sigsq=0.9)
    scale = np.abs(want).max()
    np.testing.assert_allclose(prj.project_uncentred(Z).cpu().numpy(), want, rtol=1e-11, atol=1e-12 * scale)
    want -= want.mean(axis=1)[:, None]
    np.testing.assert_allclose(prj.project(Z).cpu().numpy(), want, rtol=1e-11, atol=1e-12 * scale)


@pytest.mark.parametrize("family,S", (("linreg", 256), ("logistic", 640)))
def test_write_on_teams_matches_numpy(bc, family, S):
    """project() on a full 512-workgroup grid: the column groups of a row block are written by different workgroups (XCD
    teams; S = 640 leaves four workgroups per XCD idle) and the centring pass forms the row means from the stored values."""
    rs = np.random.RandomState(11)
    N, D = 70000, 24
    X = rs.randn(N, D) * 0.6
    theta = rs.randn(S, D) * 0.4
    if family == "linreg":
        Z, ll = np.hstack((X, rs.randn(N, 1))), (lambda z, th: linreg_log_likelihood(z, th, 0.8))
    else:
        Z, ll = X, logistic_log_likelihood
    want = ll(Z.copy(), theta)
    want -= want.mean(axis=1)[:, None]
    got = bc.DeviceProjector(family, lambda n, w, p: theta, S, sigsq=0.8).project(Z).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-11 * np.abs(want).max())


@pytest.mark.parametrize("family", ("logistic", "poisson", "linreg"))
@pytest.mark.parametrize("pad,shift", ((1, 0), (0, 1), (3, 1), (2, 0)))
def test_write_into_strided_and_unaligned_outputs(bc, family, pad, shift):
    """bcx_project_write_raw through the C ABI into an output whose leading dimension is odd and / or whose base is only
    8-byte aligned (S = 256: the 128-column tile, whose linear-regression epilogue stores 16-byte column pairs when the
    output allows it and single values otherwise), with rows and columns beyond the matrix left untouched."""
    import torch
    rs = np.random.RandomState(77 + pad + 10 * shift)
    N, D, S = 700, 70, 256
    X = rs.randn(N, D) * 0.4
    theta = rs.randn(S, D) * 0.3
    if family == "logistic":
        Z, ll = X, logistic_log_likelihood
    elif family == "poisson":
        Z, ll = np.hstack((X, rs.poisson(2.0, size=(N, 1)).astype(np.float64))), poisson_log_likelihood
    else:
        Z, ll = np.hstack((X, rs.randn(N, 1))), (lambda z, th: linreg_log_likelihood(z, th, 0.8))
    want = ll(Z.copy(), theta)
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S, sigsq=0.8)
    Zd = prj._dev(Z)
    ldo = S + pad
    buf = torch.full((N * ldo + 8,), -7.0, dtype=torch.float64, device=Zd.device)
    out = buf[shift:shift + N * ldo].view(N, ldo)
    assert out.data_ptr() % 16 == 8 * (shift % 2)
    prj._launch(prj._lib.bcx_project_write_raw, prj._common(Zd) + [out.data_ptr(), ldo], Zd)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:, :S], want, rtol=1e-11, atol=1e-12 * np.abs(want).max())
    assert np.all(got[:, S:] == -7.0)
    h = buf.cpu().numpy()
    assert np.all(h[:shift] == -7.0) and np.all(h[shift + N * ldo:] == -7.0)


def test_write_tile_variants_agree():
    """project() from the 128-column tile (transposed product, the default where S pads to a multiple of 128) and from the
    64-column tile (BCX_PROJ_WRITE_NCT=4): every value accumulates its D products in the same order in both, so the raw
    log-likelihoods agree to the last bits; each against NumPy.  One subprocess per tile (the knob is read once)."""
    import json
    import subprocess
    import sys
    code = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%r, "bayesian-coresets_amd"))
sys.path.insert(0, %r)
import bayesiancoresets_amd as bc
rs = np.random.RandomState(15)
N, D = 66000, 65
X = rs.randn(N, D) * 0.4
out = {}
for family, S in (("linreg", 200), ("poisson", 256), ("logistic", 128)):
    theta = rs.randn(S, D) * 0.3
    y = rs.randn(N, 1) if family == "linreg" else rs.poisson(2.0, size=(N, 1)).astype(np.float64)
    Z = X if family == "logistic" else np.hstack((X, y))
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S, sigsq=0.8)
    v = prj.project_uncentred(Z).cpu().numpy()
    np.save(os.path.join(sys.argv[1], family + ".npy"), v)
''' % (ROOT, ROOT)
    import tempfile
    dirs = []
    for env in ({}, {"BCX_PROJ_WRITE_NCT": "4"}):
        e = dict(os.environ)
        e.update(env)
        d = tempfile.mkdtemp()
        r = subprocess.run([sys.executable, "-c", code, d], capture_output=True, text=True, timeout=600, env=e)
        assert r.returncode == 0, r.stderr[-2000:]
        dirs.append(d)
    rs = np.random.RandomState(15)
    N, D = 66000, 65
    X = rs.randn(N, D) * 0.4
    for family, S in (("linreg", 200), ("poisson", 256), ("logistic", 128)):
        theta = rs.randn(S, D) * 0.3
        y = rs.randn(N, 1) if family == "linreg" else rs.poisson(2.0, size=(N, 1)).astype(np.float64)
        Z = X if family == "logistic" else np.hstack((X, y))
        ll = {"linreg": (lambda z, th: linreg_log_likelihood(z, th, 0.8)), "poisson": poisson_log_likelihood,
              "logistic": logistic_log_likelihood}[family]
        want = ll(Z.copy(), theta)
        a, b = (np.load(os.path.join(d, family + ".npy")) for d in dirs)
        scale = np.abs(want).max()
        np.testing.assert_allclose(a, want, rtol=1e-11, atol=1e-12 * scale)
        np.testing.assert_allclose(b, want, rtol=1e-11, atol=1e-12 * scale)
        np.testing.assert_allclose(a, b, rtol=1e-13, atol=1e-14 * scale)


@pytest.mark.parametrize("k,d,pad", ((1, 1, 0), (5, 3, 1), (64, 32, 0), (65, 33, 0), (100, 100, 0), (130, 257, 1), (200, 129, 0), (333, 1000, 2),
                                     (1497, 1024, 0), (700, 8192, 0), (2100, 64, 0),
                                     # the chip-balanced kernel (csrc/gram.hip: 16-byte aligned rows, k >= 192, d >= 64): 128 x 64 tiles ...
                                     (192, 64, 0), (193, 70, 2), (257, 1000, 0), (1025, 333, 1), (2559, 96, 0),
                                     (512, 4096, 0),      # tens of contributors per tile: their flags are polled all at once
                                     # ... and 128 x 128 tiles from k = 2560, ragged row lengths, tiles hanging over the edge
                                     (2560, 200, 0), (3000, 1030, 2), (4096, 1024, 0), (4100, 65, 1)))
def test_gram_operator_matches_numpy(bc, k, d, pad):
    """bcx_gram: G = V V^T of k rows of d doubles against NumPy, through both kernels -- csrc/gram.hip (tiles cut into equal
    ranges of stages over all workgroups; partial tiles added by the finisher) and csrc/moments.hip gram_tile_kernel (odd
    row strides, small supports): block edges, ragged row lengths, odd and padded row strides, both triangles written and
    bit-identical, nothing outside the k x k block touched."""
    import torch
    from bayesiancoresets_amd import _native as nat
    lib = nat.load()
    rs = np.random.RandomState(k * 7 + d)
    V = rs.randn(k, d)
    ld = d + pad
    buf = torch.zeros((k, ld), dtype=torch.float64, device="cuda")
    buf[:, :d] = torch.from_numpy(V).cuda()
    if pad:
        buf[:, d:] = float("nan")                      # (the pad is never read)
    ldg = k + 3
    G = torch.full((k, ldg), -5.0, dtype=torch.float64, device="cuda")
    need = int(lib.bcx_gram_scratch_bytes(k, d))
    assert need > 0
    work = torch.empty((need + 7) // 8, dtype=torch.float64, device="cuda")
    st = int(torch.cuda.current_stream().cuda_stream)
    rc = lib.bcx_gram(st, buf.data_ptr(), k, d, ld, G.data_ptr(), ldg, work.data_ptr(), work.numel() * 8)
    assert rc == 0, lib.bcx_project_last_error()
    assert lib.bcx_gram_check(st, work.data_ptr()) == 0      # (synchronises; no hand-off between workgroups timed out)
    got = G.cpu().numpy()
    want = V @ V.T
    np.testing.assert_allclose(got[:, :k], want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    assert np.array_equal(got[:, :k], got[:, :k].T)          # the mirror image is a copy
    assert np.all(got[:, k:] == -5.0)
    # argument checks: scratch too small, too many rows
    assert lib.bcx_gram(st, buf.data_ptr(), k, d, ld, G.data_ptr(), ldg, work.data_ptr(), need - 8) != 0
    assert lib.bcx_gram_scratch_bytes(20000, 8) == -1
    # the time-out word of the stream-K kernel: a call number of this process in the first 8 bytes of the scratch is reported,
    # anything else (stale contents, an earlier process) is not
    work[:1].view(torch.int64).fill_(0x1234567)
    assert lib.bcx_gram_check(st, work.data_ptr()) == 0
    assert lib.bcx_gram_check(st, None) == nat.ERR_ARG
