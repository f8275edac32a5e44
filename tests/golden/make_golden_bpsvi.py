#!/usr/bin/env python3
"""Golden vectors F17: the REFERENCE's BatchPSVICoreset (coreset/bpsvi.py) with a BlackBoxProjector on the Gaussian
linear-regression model of the examples (model_linreg: log-likelihood, its gradient in the data point, weighted posterior
sampler as in examples/linear_regression/main.py:141-147), full data and with a subsample per step, under a seeded NumPy
stream: the optimised weights and pseudo-points.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_bpsvi.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, "/root/reference/examples/common")
import bayesiancoresets as bc  # noqa: E402 (reference)
import model_linreg  # noqa: E402 (reference example model)

OUT = os.path.join(HERE, "bpsvi_golden.npz")


def main():
    rs = np.random.RandomState(17)
    N, D, S, sigsq = 3000, 5, 40, 0.49
    X = rs.randn(N, D)
    Z = np.hstack((X, (X.dot(rs.randn(D)) + 0.7 * rs.randn(N))[:, None]))
    mu0, Sig0 = np.zeros(D), 2.0 * np.eye(D)
    Sig0inv = np.linalg.inv(Sig0)

    def sampler_w(n, wts, pts):
        if wts is None or pts is None or pts.shape[0] == 0:
            muw, USigw = mu0, np.linalg.cholesky(Sig0)
        else:
            muw, USigw, _ = model_linreg.weighted_post(mu0, Sig0inv, sigsq, pts, wts)
        return muw + np.random.randn(n, muw.shape[0]).dot(USigw.T)

    ll = lambda z, th: model_linreg.log_likelihood(z, th, sigsq)
    gll = lambda z, th: model_linreg.grad_x_log_likelihood(z, th, sigsq)
    out = dict(Z=Z, mu0=mu0, Sig0=Sig0, sigsq=sigsq, S=S)
    for tag, nsub in (("full", None), ("sub", 500)):
        np.random.seed(7)
        prj = bc.BlackBoxProjector(sampler_w, S, ll, gll)
        alg = bc.BatchPSVICoreset(Z, prj, opt_itrs=25, n_subsample_opt=nsub, step_sched=lambda i: 0.5 / (1.0 + i))
        alg.build(6)
        wts, pts, idcs = alg.get()
        out[tag + "_wts"], out[tag + "_pts"], out[tag + "_idcs"] = wts, pts, idcs
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, out["full_wts"], out["sub_wts"])


if __name__ == "__main__":
    main()
