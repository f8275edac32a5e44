#!/usr/bin/env python3
"""Golden vectors for the host-side pieces of the path, produced by the REFERENCE itself (imported read-only from
/root/reference; only outputs are stored):

  F10  GIGA / Frank-Wolfe with ``check_error_monotone = False`` on the F2 input, driven into the numeric limit
       (snnls.py:45,56-62: no error comparison, the retry flag is never refreshed)
  F11  sampling baselines (snnls/sampling.py:6-37, coreset/sampling.py:5-27) under a seeded global NumPy stream
  F12  example likelihoods (examples/common/model_lr.py:25-32, model_poiss.py:25-38, model_linreg.py:4-10 and
       weighted_post :24-37) on seeded inputs incl. extreme arguments
  F13  BlackBoxProjector.project centring (projector.py:19-21)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_host.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, "/root/reference/examples/common")
import bayesiancoresets as bc  # noqa: E402 (reference)
import model_lr  # noqa: E402
import model_poiss  # noqa: E402
import model_linreg  # noqa: E402

sys.path.insert(0, HERE)
from make_golden import traced_build, sparse  # noqa: E402

OUT = os.path.join(HERE, "host_golden.npz")


def main():
    g = {}
    # ---- F10 ---------------------------------------------------------------------------------------
    np.random.seed(1)
    X = np.random.randn(10000, 100)
    for name, cls, itrs in (("giga", bc.snnls.GIGA, 700), ("fw", bc.snnls.FrankWolfe, 450)):
        s = cls(X.T, X.sum(axis=0))
        s.check_error_monotone = False
        sel, err = traced_build(s, itrs)
        idx, w = sparse(s.weights())
        k = "F10_%s_" % name
        g[k + "itrs"] = np.array(itrs)
        g[k + "sel"], g[k + "err"], g[k + "idx"], g[k + "w"] = sel, err, idx, w
        g[k + "final_err"], g[k + "size"] = np.array(s.error()), np.array(int(s.size()))
        g[k + "limit"], g[k + "n_select_calls"] = np.array(bool(s.reached_numeric_limit)), np.array(len(sel))
        print(name, "monotone check off: select calls", len(sel), "size", s.size(), "limit", s.reached_numeric_limit,
              "err", s.error(), "non-monotone accepted steps", int((np.diff(err) > 0).sum()))
    # ---- F11 ---------------------------------------------------------------------------------------
    Xs = np.random.RandomState(0).randn(50, 4)
    for name, cls in (("unif", bc.snnls.UniformSampling), ("imp", bc.snnls.ImportanceSampling)):
        np.random.seed(3)
        s = cls(Xs.T, Xs.sum(axis=0))
        s.build(30)
        k = "F11_%s_" % name
        g[k + "ps"], g[k + "w30"], g[k + "err30"], g[k + "size30"] = s.ps.copy(), s.weights(), np.array(s.error()), np.array(int(s.size()))
        s.build(25)                                   # incremental build continues the stream
        g[k + "w55"] = s.weights()
        s.optimize()
        g[k + "wopt"], g[k + "erropt"], g[k + "limit_after_opt"] = s.weights(), np.array(s.error()), np.array(bool(s.reached_numeric_limit))
        s.reset()
        g[k + "w_reset_sum"] = np.array(s.weights().sum())
    np.random.seed(4)
    c = bc.UniformSamplingCoreset(Xs)
    c.build(20)
    c.build(7)
    wts, pts, idcs = c.get()
    g["F11_usc_wts"], g["F11_usc_idcs"] = wts, np.asarray(idcs, dtype=np.int64)
    # ---- F12 ---------------------------------------------------------------------------------------
    rs = np.random.RandomState(12)
    Zl = rs.randn(40, 6) * np.array([1, 5, 30, 80, 200, 1])[None, :]            # logistic: arguments from ~1 to > 100 (linear branch)
    thl = rs.randn(9, 6)
    g["F12_lr_Z"], g["F12_lr_th"], g["F12_lr_ll"] = Zl, thl, model_lr.log_likelihood(Zl, thl)
    Xp = np.hstack((rs.randn(40, 4) * np.array([1, 10, 60, 150])[None, :], np.ones((40, 1))))
    yp = rs.poisson(3.0, size=40).astype(float)
    Zp = np.hstack((Xp, yp[:, None]))
    thp = rs.randn(7, 5)
    g["F12_poiss_Z"], g["F12_poiss_th"], g["F12_poiss_ll"] = Zp, thp, model_poiss.log_likelihood(Zp, thp)
    Zr = np.hstack((rs.randn(40, 5), 3.0 * rs.randn(40, 1)))
    thr = rs.randn(8, 5)
    g["F12_linreg_Z"], g["F12_linreg_th"], g["F12_linreg_sigsq"] = Zr, thr, np.array(0.37)
    g["F12_linreg_ll"] = model_linreg.log_likelihood(Zr, thr, 0.37)
    wr = rs.rand(40) * 3
    mu0, Sig0inv = rs.randn(5), np.diag(1.0 / (0.5 + rs.rand(5)))
    mup, USigp, LSigpInv = model_linreg.weighted_post(mu0, Sig0inv, 0.37, Zr, wr)
    g["F12_post_w"], g["F12_post_mu0"], g["F12_post_Sig0inv"] = wr, mu0, Sig0inv
    g["F12_post_mu"], g["F12_post_Sigma"] = mup, USigp.dot(USigp.T)
    mue, USige, _ = model_linreg.weighted_post(mu0, Sig0inv, 0.37, np.zeros((0, 6)), np.zeros(0))
    g["F12_post_empty_mu"], g["F12_post_empty_Sigma"] = mue, USige.dot(USige.T)
    # ---- F13 ---------------------------------------------------------------------------------------
    prj = bc.BlackBoxProjector(lambda n, w, p: thl[:n], 9, model_lr.log_likelihood)
    g["F13_vecs"] = prj.project(Zl)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, len(g), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
