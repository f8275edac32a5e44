"""GPU: SparseVI's weight optimisation with the weights resident on the device (csrc/svi.hip; reference:
coreset/sparsevi.py:69-76, util/opt.py:4-28, the sampler of examples/linear_regression/main.py:124-147).

* the posterior-draw kernel against the NumPy restatement of the model's weighted conjugate posterior
  (tests/models.py linreg_weighted_post, pinned to reference outputs by tests/test_host_golden.py F12);
* the ADAM-step kernel against the package's ``nn_opt`` (bit-for-bit restatement of util/opt.py, tests/test_host_golden.py);
* the enqueued loop of ``SparseVICoreset`` against its host loop (the reference's own sequence) and against
  oracle/sparsevi_oracle.py, on the same draws."""
import numpy as np
import pytest

from models import make_linreg_data, linreg_weighted_post

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bc():
    import bayesiancoresets_amd as bc
    return bc


def _model(rs, D):
    mu0 = rs.randn(D)
    A0 = rs.randn(D, D)
    Sig0 = A0.dot(A0.T) / D + np.eye(D)
    return mu0, Sig0, 0.37


@pytest.mark.parametrize("D,k,S", ((7, 3, 16), (30, 1, 40), (31, 12, 64), (301, 4, 320), (301, 33, 320), (100, 64, 300),
                                   (302, 0, 310), (129, 17, 1024), (1000, 3, 1010), (2, 2, 5), (511, 16, 520), (40, 5, 3000),
                                   # more points than the rank-k kernels take: the D x D form (csrc/lrpost.hip)
                                   (301, 28, 320), (301, 65, 320), (301, 300, 320), (40, 130, 64), (33, 1000, 40), (7, 70, 16),
                                   (640, 100, 650), (1024, 40, 1030)))
def test_posterior_draw_kernel(bc, D, k, S):
    """theta = mu_w + R Uw^T with Uw Uw^T = Sigma_w: the rows of R are the unit vectors (they return Uw itself), a zero row
    (the mean) and standard-normal rows (compared with mu_w + R Uw^T for the Uw just read)."""
    import torch
    rs = np.random.RandomState(1000 * D + k)
    mu0, Sig0, sigsq = _model(rs, D)
    pts = rs.randn(k, D + 1)
    wts = np.abs(rs.randn(k)) * 30.0
    if k > 1:
        wts[0] = 0.0                       # a point that has just entered the coreset
    smp = bc.LinregPosteriorSampler(mu0, Sig0, sigsq, seed=3)
    ld = D + D % 2
    R = np.zeros((S, ld))
    R[:D, :D] = np.eye(D)
    R[D + 1:, :] = rs.randn(S - D - 1, ld)          # (row D stays zero; the padding column holds noise: it must not matter)
    Rd = torch.from_numpy(R).cuda()
    smp._noise = lambda n: Rd
    theta = smp(S, wts, pts)
    assert tuple(theta.shape) == (S, D) and theta.stride(0) == ld and theta.data_ptr() % 16 == 0
    th = theta.cpu().numpy()
    if k:
        mu_ref, U_ref = linreg_weighted_post(mu0, np.linalg.inv(Sig0), sigsq, pts, wts)
    else:
        mu_ref, U_ref = mu0, np.linalg.cholesky(Sig0)
    cov_ref = U_ref.dot(U_ref.T)
    mu = th[D]
    np.testing.assert_allclose(mu, mu_ref, rtol=1e-9, atol=1e-10 * np.abs(mu_ref).max())
    UwT = th[:D] - mu
    cov = UwT.T.dot(UwT)
    assert np.abs(cov - cov_ref).max() <= 1e-10 * np.abs(cov_ref).max()
    want = mu + R[:, :D].dot(UwT)
    assert np.abs(th - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
    np.testing.assert_allclose(smp.mean.cpu().numpy(), th.mean(axis=0), rtol=1e-12, atol=1e-13 * np.abs(th).max())
    # the same points again with other weights (cached point state), then other points of the same shape: nothing stale
    if k:
        w2 = wts * 0.5 + 1.0
        mu2 = smp(S, w2, pts).cpu().numpy()[D]
        np.testing.assert_allclose(mu2, linreg_weighted_post(mu0, np.linalg.inv(Sig0), sigsq, pts, w2)[0], rtol=1e-9, atol=1e-10)
        p3 = pts + 0.25
        mu3 = smp(S, w2, p3).cpu().numpy()[D]
        np.testing.assert_allclose(mu3, linreg_weighted_post(mu0, np.linalg.inv(Sig0), sigsq, p3, w2)[0], rtol=1e-9, atol=1e-10)
        # from device-resident weights: the same kernel, the same numbers
        plan = smp.enqueue_plan(S, p3, 2)
        assert plan is not None
        plan.set_noise(Rd[None, :, :])
        assert plan.factored == (not smp._low_rank(k)) and plan.fast == (smp._low_rank(k) and bool(smp._lib.bcx_linreg_posterior_apply_ok(k, ld)))
        plan.check()
        t4, m4 = plan.draw(torch.from_numpy(w2).cuda(), 0)
        t4, m4 = t4.cpu().numpy(), m4.cpu().numpy()
        np.testing.assert_allclose(t4[D], mu3, rtol=1e-12, atol=1e-13 * np.abs(mu3).max())
        U3 = smp(S, w2, p3).cpu().numpy()
        assert np.abs(t4 - U3).max() <= 1e-12 * max(1.0, np.abs(U3).max())      # (the call form at the same noise)
        np.testing.assert_allclose(m4, t4.mean(axis=0), rtol=1e-12, atol=1e-13 * np.abs(t4).max())


def test_posterior_sampler_limits(bc):
    rs = np.random.RandomState(0)
    mu0, Sig0, sigsq = _model(rs, 6)
    smp = bc.LinregPosteriorSampler(mu0, Sig0, sigsq)
    assert smp.enqueue_plan(16, rs.randn(4097, 7), 3) is None and smp.enqueue_plan(4097, rs.randn(3, 7), 3) is None
    assert smp.enqueue_plan(16, rs.randn(65, 7), 3) is not None
    smp.NOISE_BUDGET = 1000                                  # (a loop whose normal numbers would not fit: the host loop serves it)
    assert smp.enqueue_plan(16, rs.randn(3, 7), 3) is None
    del smp.NOISE_BUDGET
    with pytest.raises(ValueError):
        smp(16, np.ones(4097), rs.randn(4097, 7))
    with pytest.raises(ValueError):
        smp(16, np.ones(2), rs.randn(2, 9))
    assert tuple(smp(5, np.array([]), np.array([])).shape) == (5, 6)
    # draws have the posterior's first two moments
    pts, wts = rs.randn(4, 7), rs.rand(4) * 5
    mu, U = linreg_weighted_post(mu0, np.linalg.inv(Sig0), sigsq, pts, wts)
    big = np.vstack([smp(1024, wts, pts).cpu().numpy() for _ in range(20)])
    sd = np.sqrt(np.diag(U.dot(U.T)).max())
    np.testing.assert_allclose(big.mean(axis=0), mu, atol=6 * sd / np.sqrt(big.shape[0]))
    np.testing.assert_allclose(np.cov(big.T), U.dot(U.T), atol=0.08 * sd ** 2)


@pytest.mark.parametrize("k,S,raw", ((1, 16, 0), (5, 256, 1), (64, 100, 1), (17, 1000, 0), (4, 256, 1)))
def test_adam_step_kernel_against_nn_opt(bc, k, S, raw):
    """``raw``: the projected coreset points arrive uncentred (bcx_project_write_raw) and the kernel takes the row means."""
    import ctypes
    import torch
    from bayesiancoresets_amd import _native
    from bayesiancoresets_amd.util.opt import nn_opt
    lib = _native.load()
    rs = np.random.RandomState(k * 7 + S)
    core_raw = rs.randn(k, S) + 5.0 * rs.randn(k, 1)
    core = core_raw - core_raw.mean(axis=1)[:, None]
    colsum = 3.0 * rs.randn(S)
    w0 = np.abs(rs.randn(k))
    T, scaling = 25, 1.7
    sched_fn = lambda i: 0.3 / (1.0 + i)
    b1, b2, eps = 0.9, 0.999, 1e-8

    def grd(w):
        return -core.dot(scaling * colsum - w.dot(core)) / S
    want = nn_opt(w0, grd, opt_itrs=T, step_sched=sched_fn)
    sched = np.array([(sched_fn(i), 1.0 - b1 ** (i + 1), 1.0 - b2 ** (i + 1)) for i in range(T)])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    core_d, col_d, w, m1, m2, sc, tr = d(core_raw if raw else core), d(colsum), d(w0), d(np.zeros(k)), d(np.zeros(k)), d(sched), d(np.zeros((T, k)))
    stream = int(torch.cuda.current_stream().cuda_stream)
    for i in range(T):
        assert lib.bcx_sparsevi_adam_step(stream, k, S, col_d.data_ptr(), scaling, core_d.data_ptr(), S, w.data_ptr(), m1.data_ptr(),
                                          m2.data_ptr(), sc.data_ptr(), i, b1, b2, eps, tr.data_ptr(), raw) == 0
    got = w.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-13)
    assert np.array_equal(tr.cpu().numpy()[-1], got)
    assert (got >= 0).all()
    assert lib.bcx_sparsevi_adam_step(stream, 65, S, col_d.data_ptr(), scaling, core_d.data_ptr(), S, w.data_ptr(), m1.data_ptr(),
                                      m2.data_ptr(), sc.data_ptr(), 0, b1, b2, eps, None, 0) == _native.ERR_ARG


class _ReplaySampler(object):
    """Wraps the package sampler so that the call form and the enqueued form consume the SAME pre-drawn normal numbers."""

    def __init__(self, inner, noise):
        self.inner, self.noise, self.at = inner, noise, 0
        inner._noise = self._one
        inner._noise_block = self._block

    def _one(self, n):
        r = self.noise[self.at]
        self.at += 1
        return r

    def _block(self, steps, n):
        r = self.noise[self.at:self.at + steps]
        self.at += steps
        return r

    def __call__(self, n, wts, pts):
        return self.inner(n, wts, pts)

    def enqueue_plan(self, n, pts, steps):
        return self.inner.enqueue_plan(n, pts, steps)


@pytest.mark.parametrize("colsum,prior", (("mfma", "iso"), ("moments", "iso"), ("moments", "dense")))
def test_enqueued_loop_matches_the_host_loop(bc, colsum, prior):
    """Three greedy steps of SparseVICoreset, opt_itrs = 30: the ADAM loop enqueued on the device-resident weights against the
    host loop (nn_opt around projector.update + two projections, the reference's sequence) on the same normal draws."""
    import torch
    D, N, S, T, steps = 12, 20000, 64, 30, 3
    rs = np.random.RandomState(11)
    Z = make_linreg_data(11, N, D)
    mu0, Sig0, sigsq = np.zeros(D), 4.0 * np.eye(D), 1.0
    if prior == "dense":                   # (an isotropic prior takes a shortcut: R U0^T is a column scaling)
        A0 = rs.randn(D, D)
        mu0, Sig0 = 0.1 * rs.randn(D), 2.0 * (A0.dot(A0.T) / D + np.eye(D))
    g = torch.Generator(device="cuda")
    g.manual_seed(17)
    noise = torch.randn(steps * (T + 1) + 4, S, D + D % 2, dtype=torch.float64, device="cuda", generator=g)
    out = {}
    for mode in (True, False):
        smp = _ReplaySampler(bc.LinregPosteriorSampler(mu0, Sig0, sigsq), noise)
        prj = bc.DeviceProjector("linreg", smp, S, sigsq=sigsq, colsum=colsum)
        alg = bc.SparseVICoreset(Z, prj, opt_itrs=T)
        alg.ENQUEUE = mode
        alg.build(steps)
        out[mode] = (alg.wts.copy(), alg.idcs.copy(), smp.at)
    assert out[True][2] == out[False][2] == 1 + steps * (T + 1)      # (the constructor's draw, then select + T per step)
    assert np.array_equal(out[True][1], out[False][1]) and out[True][1].shape[0] >= 2
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-8, atol=1e-12)
    assert (out[True][0] > 0).any()


@pytest.mark.parametrize("colsum", ("mfma", "moments"))
def test_enqueued_loop_against_the_oracle(bc, colsum):
    """The same loop against oracle/sparsevi_oracle.py (the NumPy restatement of sparsevi.py + projector.py + opt.py, pinned to
    the reference's own runs F6 / F6b): the oracle's sampler callback is the device sampler's call form, fed the same normal
    numbers the enqueued loop consumes, so both sides see the same parameter draws and differ only in who does the arithmetic."""
    import torch
    from oracle.sparsevi_oracle import SparseVIOracle, linreg_loglik
    D, N, S, T, steps = 10, 12000, 48, 25, 3
    rs = np.random.RandomState(5)
    Z = make_linreg_data(7, N, D)
    A0 = rs.randn(D, D)
    mu0, Sig0, sigsq = 0.2 * rs.randn(D), 1.5 * (A0.dot(A0.T) / D + np.eye(D)), 0.8
    g = torch.Generator(device="cuda")
    g.manual_seed(23)
    noise = torch.randn(steps * (T + 1) + 4, S, D + D % 2, dtype=torch.float64, device="cuda", generator=g)
    ref_smp = _ReplaySampler(bc.LinregPosteriorSampler(mu0, Sig0, sigsq), noise)
    orc = SparseVIOracle(Z, lambda n, w, p: ref_smp(n, np.asarray(w, dtype=np.float64), p).cpu().numpy().copy(),
                         lambda z, th: linreg_loglik(z, th, sigsq), S, opt_itrs=T)
    for _ in range(steps):
        orc.step()
    smp = _ReplaySampler(bc.LinregPosteriorSampler(mu0, Sig0, sigsq), noise)
    alg = bc.SparseVICoreset(Z, bc.DeviceProjector("linreg", smp, S, sigsq=sigsq, colsum=colsum), opt_itrs=T)
    alg.build(steps)
    assert alg._enqueue_plan() is not None and smp.at - T == ref_smp.at == 1 + steps * (T + 1)      # (the probe above drew a plan's worth)
    assert np.array_equal(alg.idcs, orc.idcs) and orc.idcs.shape[0] >= 2
    np.testing.assert_allclose(alg.wts, orc.wts, rtol=1e-7, atol=1e-11)


def test_step_plan_refuses_draws_it_would_have_to_copy(bc):
    import torch
    D, S = 5, 16
    prj = bc.DeviceProjector("linreg", lambda n, w, p: np.zeros((n, D)), S, sigsq=1.0)
    Z = torch.randn(5000, D + 1, dtype=torch.float64, device="cuda")
    odd = torch.zeros(S, D, dtype=torch.float64, device="cuda")           # rows of 5 doubles: not 16-byte aligned
    with pytest.raises(ValueError):
        prj.enqueue_step_plan(Z, Z[:2], True, odd, None)
    even = torch.zeros(S, D + 1, dtype=torch.float64, device="cuda")[:, :D]
    run, buf, k = prj.enqueue_step_plan(Z, Z[:2], True, even, None)
    run()
    torch.cuda.synchronize()
    assert k == 2 and buf.numel() == S * 3 and bool(torch.isfinite(buf).all())


def test_enqueue_is_declined_where_the_host_has_to_act(bc):
    """Per-step sub-samples are drawn on the host (sparsevi.py:33) and a plain sampler has no device form: host loop."""
    D, N, S = 6, 5000, 32
    Z = make_linreg_data(3, N, D)
    mu0, Sig0, sigsq = np.zeros(D), np.eye(D), 1.0
    smp = bc.LinregPosteriorSampler(mu0, Sig0, sigsq, seed=1)
    np.random.seed(4)
    alg = bc.SparseVICoreset(Z, bc.DeviceProjector("linreg", smp, S, sigsq=sigsq), n_subsample_opt=1000, opt_itrs=5)
    alg.build(2)
    assert alg._enqueue_plan() is None and alg.size() >= 1
    alg2 = bc.SparseVICoreset(Z, bc.DeviceProjector("linreg", lambda n, w, p: smp(n, w, p), S, sigsq=sigsq), opt_itrs=5)
    alg2.build(2)
    assert alg2._enqueue_plan() is None
    alg3 = bc.SparseVICoreset(Z, bc.DeviceProjector("linreg", smp, S, sigsq=sigsq), opt_itrs=5)
    alg3.build(2)
    assert alg3._enqueue_plan() is not None


@pytest.mark.parametrize("D,k,prior", ((7, 3, "dense"), (31, 40, "dense"), (32, 5, "iso"), (33, 64, "dense"), (64, 200, "iso"), (100, 1, "dense"),
                                       (301, 300, "iso"), (301, 65, "dense"), (302, 0, "dense"), (511, 130, "dense"), (1000, 50, "iso"),
                                       (1024, 1100, "dense"), (96, 4096, "iso")))
def test_posterior_factor_kernel(bc, D, k, prior):
    """bcx_linreg_posterior_factor (csrc/lrpost.hip) against NumPy: L^-1 of the Cholesky factor of the weighted precision and the
    posterior mean, as examples/common/model_linreg.py:26-41 forms them (tests/models.py, pinned to reference outputs by F12)."""
    import torch
    from bayesiancoresets_amd import _native
    lib = _native.load()
    rs = np.random.RandomState(31 * D + k)
    mu0 = rs.randn(D)
    if prior == "dense":
        A0 = rs.randn(D, D)
        Sig0 = A0.dot(A0.T) / D + np.eye(D)
    else:
        Sig0 = 2.5 * np.eye(D)
    sigsq = 0.4
    S0inv = np.linalg.inv(Sig0)
    pts = rs.randn(max(k, 1), D + 1)[:k]
    w = np.abs(rs.randn(k)) * 20.0
    if k > 2:
        w[1] = 0.0
        w[2] = -3.0                                           # (clamped at zero, as the rank-k kernels do)
    ld = D + D % 2
    ldk = max((k + 31) // 32 * 32, 32)
    XT = np.zeros((D, ldk))
    XT[:, :k] = pts[:, :-1].T
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    need = int(lib.bcx_linreg_posterior_factor_scratch_bytes(D))
    assert need > 0 and lib.bcx_linreg_posterior_factor_scratch_bytes(1025) == -1
    work = torch.empty(need // 8, dtype=torch.float64, device="cuda")
    U, mu, uv = torch.zeros(D, ld, dtype=torch.float64, device="cuda"), torch.zeros(D, dtype=torch.float64, device="cuda"), torch.zeros(D, dtype=torch.float64, device="cuda")
    w_d, X_d, y_d, S_d, r_d = d(w if k else np.zeros(1)), d(XT), d(pts[:, -1] if k else np.zeros(1)), d(S0inv), d(S0inv.dot(mu0))
    st = int(torch.cuda.current_stream().cuda_stream)
    for rep in range(2):                                      # (the second call reuses the scratch: flags, tiles)
        rc = lib.bcx_linreg_posterior_factor(st, k, D, ldk, w_d.data_ptr(), X_d.data_ptr(), y_d.data_ptr(), S_d.data_ptr(), D, r_d.data_ptr(),
                                             sigsq, work.data_ptr(), work.numel() * 8, U.data_ptr(), ld, uv.data_ptr(), mu.data_ptr())
        assert rc == 0, lib.bcx_project_last_error()
        assert lib.bcx_linreg_posterior_factor_status(st, D, work.data_ptr()) == 0, lib.bcx_project_last_error()
    wc = np.maximum(w, 0.0)
    X, y = pts[:, :-1], pts[:, -1]
    P = S0inv + (wc[:, None] * X).T.dot(X) / sigsq
    L = np.linalg.cholesky(P)
    want = np.linalg.inv(L).T                                 # USigp of model_linreg.py:33
    got = U.cpu().numpy()
    assert np.all(got[:, D:] == 0.0) and np.all(np.tril(got[:, :D], -1) == 0.0)
    assert np.abs(got[:, :D] - want).max() <= 1e-13 * D * np.abs(want).max() * max(1.0, np.linalg.cond(L) * 1e-2)
    mu_ref = np.linalg.solve(P, S0inv.dot(mu0) + (wc * y).dot(X) / sigsq)
    np.testing.assert_allclose(mu.cpu().numpy(), mu_ref, rtol=1e-9, atol=1e-10 * np.abs(mu_ref).max())
    np.testing.assert_allclose(uv.cpu().numpy(), np.linalg.solve(L, S0inv.dot(mu0) + (wc * y).dot(X) / sigsq), rtol=1e-9, atol=1e-10 * np.abs(mu_ref).max() * np.abs(L).max())
    # the draws from this factor: theta = mu + R U^T = (R + 1 u^T) U^T, the mean of the draws from the column means of R
    S = 70
    R = torch.from_numpy(rs.randn(S, ld)).cuda()
    rbar = R.mean(dim=0)
    theta, tbar = torch.zeros(S, ld, dtype=torch.float64, device="cuda"), torch.zeros(D, dtype=torch.float64, device="cuda")
    assert lib.bcx_linreg_posterior_draw_factored(st, D, ld, U.data_ptr(), ld, uv.data_ptr(), R.data_ptr(), rbar.data_ptr(), S,
                                                  theta.data_ptr(), tbar.data_ptr()) == 0, lib.bcx_project_last_error()
    th_ref = mu_ref + R.cpu().numpy()[:, :D].dot(want.T)
    th = theta.cpu().numpy()
    assert np.abs(th[:, :D] - th_ref).max() <= 1e-11 * np.abs(th_ref).max() * max(1.0, np.linalg.cond(L) * 1e-2) and np.all(th[:, D:] == 0.0)
    np.testing.assert_allclose(tbar.cpu().numpy(), th[:, :D].mean(axis=0), rtol=1e-10, atol=1e-12 * np.abs(th).max())
    # argument checks
    assert lib.bcx_linreg_posterior_factor(st, k, D, ldk, w_d.data_ptr(), X_d.data_ptr(), y_d.data_ptr(), S_d.data_ptr(), D, r_d.data_ptr(),
                                           sigsq, work.data_ptr(), need - 8, U.data_ptr(), ld, uv.data_ptr(), mu.data_ptr()) == _native.ERR_ARG
    assert lib.bcx_linreg_posterior_factor(st, 4097, D, ldk, w_d.data_ptr(), X_d.data_ptr(), y_d.data_ptr(), S_d.data_ptr(), D, r_d.data_ptr(),
                                           sigsq, work.data_ptr(), need, U.data_ptr(), ld, uv.data_ptr(), mu.data_ptr()) == _native.ERR_ARG
    if k > 32:
        assert lib.bcx_linreg_posterior_factor(st, k, D, ldk - 32, w_d.data_ptr(), X_d.data_ptr(), y_d.data_ptr(), S_d.data_ptr(), D, r_d.data_ptr(),
                                               sigsq, work.data_ptr(), need, U.data_ptr(), ld, uv.data_ptr(), mu.data_ptr()) == _native.ERR_ARG


@pytest.mark.parametrize("k,S,raw", ((17, 64, 1), (33, 64, 0), (65, 256, 1), (130, 100, 0), (300, 256, 1), (1000, 48, 1), (67, 1000, 0)))
def test_adam_step_ws_kernels_against_nn_opt(bc, k, S, raw):
    """More than 16 weights: the two-launch form of the ADAM step (csrc/svi.hip svi_adam_a / b_kernel) against ``nn_opt``."""
    import torch
    from bayesiancoresets_amd import _native
    from bayesiancoresets_amd.util.opt import nn_opt
    lib = _native.load()
    rs = np.random.RandomState(k * 7 + S)
    core_raw = rs.randn(k, S) + 5.0 * rs.randn(k, 1)
    core = core_raw - core_raw.mean(axis=1)[:, None]
    colsum = 3.0 * rs.randn(S)
    w0 = np.abs(rs.randn(k))
    T, scaling = 12, 1.7
    sched_fn = lambda i: 0.3 / (1.0 + i)
    b1, b2, eps = 0.9, 0.999, 1e-8

    def grd(w):
        return -core.dot(scaling * colsum - w.dot(core)) / S
    want = nn_opt(w0, grd, opt_itrs=T, step_sched=sched_fn)
    sched = np.array([(sched_fn(i), 1.0 - b1 ** (i + 1), 1.0 - b2 ** (i + 1)) for i in range(T)])
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()
    core_d, col_d, w, m1, m2, sc, tr = d(core_raw if raw else core), d(colsum), d(w0), d(np.zeros(k)), d(np.zeros(k)), d(sched), d(np.zeros((T, k)))
    need = int(lib.bcx_sparsevi_adam_scratch_bytes(k, S))
    assert need > 0 and lib.bcx_sparsevi_adam_scratch_bytes(16, S) == 0 and lib.bcx_sparsevi_adam_scratch_bytes(4097, S) == -1
    work = torch.empty(need // 8, dtype=torch.float64, device="cuda")
    stream = int(torch.cuda.current_stream().cuda_stream)
    for i in range(T):
        assert lib.bcx_sparsevi_adam_step_ws(stream, k, S, col_d.data_ptr(), scaling, core_d.data_ptr(), S, w.data_ptr(), m1.data_ptr(),
                                             m2.data_ptr(), sc.data_ptr(), i, b1, b2, eps, tr.data_ptr(), raw, work.data_ptr(), need) == 0
    got = w.cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12)
    assert np.array_equal(tr.cpu().numpy()[-1], got) and (got >= 0).all()
    assert lib.bcx_sparsevi_adam_step_ws(stream, k, S, col_d.data_ptr(), scaling, core_d.data_ptr(), S, w.data_ptr(), m1.data_ptr(),
                                         m2.data_ptr(), sc.data_ptr(), 0, b1, b2, eps, None, raw, work.data_ptr(), need - 8) == _native.ERR_ARG


@pytest.mark.parametrize("D,k,colsum", ((301, 28, "moments"), (24, 65, "mfma"), (24, 130, "moments"), (301, 300, "moments"), (12, 300, "mfma")))
def test_enqueued_loop_against_the_oracle_at_reference_coreset_sizes(bc, D, k, colsum):
    """The weight optimisation (sparsevi.py:69-76) of a coreset of k points -- the sizes the reference's experiment reaches
    (examples/linear_regression/main.py:284 coreset_size_max = 300) -- enqueued on the device against oracle/sparsevi_oracle.py
    on the same parameter draws: both sides start from the same k seeded points and weights (what k greedy steps leave)."""
    import torch
    from oracle.sparsevi_oracle import SparseVIOracle, linreg_loglik
    N, S, T = 4000, 32, 12
    rs = np.random.RandomState(100 * D + k)
    Z = make_linreg_data(13, N, D)
    A0 = rs.randn(D, D)
    mu0, Sig0, sigsq = 0.2 * rs.randn(D), 1.5 * (A0.dot(A0.T) / D + np.eye(D)), 0.8
    idcs = np.sort(rs.choice(N, size=k, replace=False)).astype(np.int64)
    w0 = np.abs(rs.randn(k)) * (N / k)
    w0[::7] = 0.0
    g = torch.Generator(device="cuda")
    g.manual_seed(29)
    noise = torch.randn(T + 3, S, D + D % 2, dtype=torch.float64, device="cuda", generator=g)
    ref_smp = _ReplaySampler(bc.LinregPosteriorSampler(mu0, Sig0, sigsq), noise)
    orc = SparseVIOracle(Z, lambda n, w, p: ref_smp(n, np.asarray(w, dtype=np.float64), p).cpu().numpy().copy(),
                         lambda z, th: linreg_loglik(z, th, sigsq), S, opt_itrs=T)
    orc.wts, orc.idcs, orc.pts = w0.copy(), idcs.copy(), Z[idcs].copy()
    orc.optimize()
    smp = _ReplaySampler(bc.LinregPosteriorSampler(mu0, Sig0, sigsq), noise)
    alg = bc.SparseVICoreset(Z, bc.DeviceProjector("linreg", smp, S, sigsq=sigsq, colsum=colsum), opt_itrs=T)
    alg.wts, alg.idcs, alg.pts = w0.copy(), idcs.copy(), Z[idcs].copy()
    plan = alg._enqueue_plan()
    assert plan is not None                                   # (no host loop at these sizes)
    smp.at -= T                                               # (the probe drew a plan's worth of normal numbers)
    alg._optimize()
    assert smp.at == ref_smp.at == 1 + T
    assert (alg.wts > 0).sum() >= k // 2
    np.testing.assert_allclose(alg.wts, orc.wts, rtol=1e-7, atol=1e-9 * np.abs(orc.wts).max())


def test_standard_normal_generator_and_column_means(bc):
    """bcx_standard_normal (Philox-4x32-10 + Box-Muller): reproducible, the same numbers whatever the split into calls, first
    four moments and a Kolmogorov-Smirnov distance of a standard normal; bcx_column_means against NumPy."""
    import torch
    from scipy import stats
    from bayesiancoresets_amd import _native
    lib = _native.load()
    st = int(torch.cuda.current_stream().cuda_stream)
    n = 1_000_001                                             # (odd: the last pair is half used)
    a, b = torch.empty(n, dtype=torch.float64, device="cuda"), torch.empty(n, dtype=torch.float64, device="cuda")
    assert lib.bcx_standard_normal(st, 12345, 0, n, a.data_ptr()) == 0
    assert lib.bcx_standard_normal(st, 12345, 0, 400, b.data_ptr()) == 0
    assert lib.bcx_standard_normal(st, 12345, 200, n - 400, b[400:].data_ptr()) == 0      # (400 numbers = 200 pair counters)
    x = a.cpu().numpy()
    assert np.array_equal(x, b.cpu().numpy()) and np.isfinite(x).all()
    assert lib.bcx_standard_normal(st, 12346, 0, n, b.data_ptr()) == 0
    assert not np.array_equal(x[:100], b.cpu().numpy()[:100])
    assert abs(x.mean()) < 5e-3 and abs(x.var() - 1.0) < 5e-3 and abs(stats.skew(x)) < 1e-2 and abs(stats.kurtosis(x)) < 2e-2
    assert stats.kstest(x, "norm").statistic < 2.5e-3
    assert abs(np.corrcoef(x[:-1:2], x[1::2])[0, 1]) < 5e-3   # (the two numbers of a pair are independent)
    # column means of blocks
    rs = np.random.RandomState(0)
    for (nb, rows, ld) in ((1, 7, 5), (3, 256, 302), (5, 33, 64), (2, 1000, 130)):
        R = rs.randn(nb, rows, ld)
        Rd = torch.from_numpy(R).cuda()
        out = torch.zeros(nb, ld + 3, dtype=torch.float64, device="cuda")
        assert lib.bcx_column_means(st, Rd.data_ptr(), nb, rows, ld, rows * ld, out.data_ptr(), ld + 3) == 0
        got = out.cpu().numpy()
        np.testing.assert_allclose(got[:, :ld], R.mean(axis=1), rtol=1e-13, atol=1e-15)
        assert np.all(got[:, ld:] == 0.0)
    assert lib.bcx_column_means(st, Rd.data_ptr(), 0, 1, 1, 0, out.data_ptr(), 1) == _native.ERR_ARG


# ---- the Laplace sampler of the logistic / Poisson experiment on the device (csrc/laplace.hip) ---------------------------------
def _laplace_case(family, D, k, seed):
    sys_path_examples()
    import model_lr
    import model_poiss
    rs = np.random.RandomState(seed)
    if family == "logistic":
        pts = model_lr.synthetic_rows(max(k, 1), D, rs)[:k]
        fit = model_lr.laplace_fit
    else:
        pts = model_poiss.synthetic_rows(max(k, 1), D, rs)[:k]
        fit = model_poiss.laplace_fit
    wts = np.abs(rs.randn(k)) * 3.0
    if k > 3:
        wts[1] = 0.0
    return pts, wts, fit


def sys_path_examples():
    import os
    import sys
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bayesian-coresets_amd", "examples", "common")
    if p not in sys.path:
        sys.path.insert(0, p)


@pytest.mark.parametrize("family,D,k", (("logistic", 10, 40), ("logistic", 3, 1), ("logistic", 32, 300), ("logistic", 7, 0),
                                        ("poisson", 10, 60), ("poisson", 5, 2), ("poisson", 31, 250), ("poisson", 4, 900)))
def test_laplace_sampler_against_the_host_fit(bc, family, D, k):
    """bc.LaplacePosteriorSampler (one launch: damped Newton + in-register Cholesky + draws) against the package's host
    ``laplace_fit`` (examples/common/model_lr.py / model_poiss.py: the same objective, pinned to the reference's get_laplace
    outputs by tests/test_host_golden.py): mode, covariance, and the draws as mu + R W for the W just read."""
    import torch
    pts, wts, fit = _laplace_case(family, D, k, 50 * D + k)
    smp = bc.LaplacePosteriorSampler(family, D, seed=4)
    mu, W = smp.posterior(wts if k else None, pts if k else None)
    if k and (wts > 0).any():
        mu_ref, cov_ref = fit(pts[wts > 0], wts[wts > 0])
    else:
        mu_ref, cov_ref = np.zeros(D), np.eye(D)
    np.testing.assert_allclose(mu, mu_ref, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(W.T.dot(W), cov_ref, rtol=1e-6, atol=1e-9 * np.abs(cov_ref).max())
    assert np.all(np.triu(W, 1) == 0.0) and smp.newton_steps <= 60
    # the draws at the sampler's own normal numbers: right shape / alignment, the posterior's first two moments
    n = 2000
    th = smp(n, wts if k else None, pts if k else None)
    assert tuple(th.shape) == (n, D) and th.stride(0) == D + D % 2 and th.data_ptr() % 16 == 0
    t = th.cpu().numpy()
    sd = np.sqrt(np.diag(cov_ref))
    assert np.all(np.abs(t.mean(axis=0) - mu_ref) < 6 * sd / np.sqrt(n))
    np.testing.assert_allclose(smp.mean.cpu().numpy(), t.mean(axis=0), rtol=1e-10, atol=1e-12)
    assert np.abs(np.cov(t.T).reshape(D, D) - cov_ref).max() < 0.25 * (sd.max() ** 2)


def test_laplace_sampler_limits(bc):
    rs = np.random.RandomState(1)
    with pytest.raises(ValueError):
        bc.LaplacePosteriorSampler("gamma", 3)
    smp = bc.LaplacePosteriorSampler("logistic", 33)
    assert not smp.supports(8, 4) and smp.enqueue_plan(8, rs.randn(4, 33), 3) is None
    with pytest.raises(ValueError):
        smp(8, np.ones(4), rs.randn(4, 33))
    smp = bc.LaplacePosteriorSampler("poisson", 6)
    assert smp.enqueue_plan(8, rs.randn(3000, 7), 3) is None              # (the points would not fit the workgroup's LDS)
    with pytest.raises(ValueError):
        smp(8, np.ones(2), rs.randn(2, 6))                                 # Poisson rows carry the response: 7 columns
    assert tuple(smp(5, np.array([]), np.array([])).shape) == (5, 6)       # no points: the prior


@pytest.mark.parametrize("family", ("logistic", "poisson"))
def test_laplace_enqueued_loop_matches_the_host_loop(bc, family):
    """SparseVI on the logistic / Poisson model with bc.LaplacePosteriorSampler: the ADAM loop enqueued on the device-resident
    weights (one Laplace fit per step where they are, csrc/laplace.hip) against the host loop (nn_opt around projector.update
    + two projections, the reference's sequence) on the same normal numbers."""
    import torch
    sys_path_examples()
    import model_lr
    import model_poiss
    D, N, S, T, steps = 6, 6000, 64, 12, 3
    rs = np.random.RandomState(3)
    Z = model_lr.synthetic_rows(N, D, rs) if family == "logistic" else model_poiss.synthetic_rows(N, D, rs)
    g = torch.Generator(device="cuda")
    g.manual_seed(31)
    noise = torch.randn(steps * (T + 1) + 4, S, D + D % 2, dtype=torch.float64, device="cuda", generator=g)
    out = {}
    for mode in (True, False):
        smp = _ReplaySampler(bc.LaplacePosteriorSampler(family, D), noise)
        prj = bc.DeviceProjector(family, smp, S)
        alg = bc.SparseVICoreset(Z, prj, opt_itrs=T)
        alg.ENQUEUE = mode
        alg.build(steps)
        if mode:
            assert alg._enqueue_plan() is not None
            smp.at -= T
        out[mode] = (alg.wts.copy(), alg.idcs.copy(), smp.at)
    assert out[True][2] == out[False][2] == 1 + steps * (T + 1)
    assert np.array_equal(out[True][1], out[False][1]) and out[True][1].shape[0] >= 2
    np.testing.assert_allclose(out[True][0], out[False][0], rtol=1e-6, atol=1e-10)
    assert (out[True][0] > 0).any()
