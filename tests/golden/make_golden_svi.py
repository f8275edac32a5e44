#!/usr/bin/env python3
"""Golden vectors for the config-5 style path: the REFERENCE's SparseVICoreset with a
BlackBoxProjector on synthetic Gaussian linear regression (model_linreg from /root/reference).
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_svi.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))          # tests/: data generator shared with the tests
sys.path.insert(0, "/root/reference")
sys.path.insert(1, "/root/reference/examples/common")
import bayesiancoresets as bc  # noqa: E402 (reference)
import model_linreg  # noqa: E402 (reference example model)
from models import make_linreg_data  # noqa: E402

OUT = os.path.join(HERE, "svi_golden.npz")


def main():
    N, D, S, opt_itrs, steps, sigsq = 50000, 30, 64, 20, 5, 1.0
    Z = make_linreg_data(1, N, D)
    mu0, Sig0 = np.zeros(D), np.eye(D)
    Sig0inv = np.linalg.inv(Sig0)

    def sampler_w(n, wts, pts):     # examples/linear_regression/main.py:141-147
        if wts is None or pts is None or pts.shape[0] == 0:
            muw, USigw = mu0, np.linalg.cholesky(Sig0)
        else:
            muw, USigw, _ = model_linreg.weighted_post(mu0, Sig0inv, sigsq, pts, wts)
        return muw + np.random.randn(n, muw.shape[0]).dot(USigw.T)

    np.random.seed(2)
    prj = bc.BlackBoxProjector(sampler_w, S, lambda z, th: model_linreg.log_likelihood(z, th, sigsq))
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=opt_itrs)
    hist_idcs, hist_wts = [], []
    for _ in range(steps):
        alg.build(1)
        hist_idcs.append(alg.idcs.copy())
        hist_wts.append(alg.wts.copy())
    wts, pts, idcs = alg.get()
    print("idcs (selection order):", alg.idcs, "wts:", alg.wts)
    g = {"N": np.array(N), "D": np.array(D), "S": np.array(S), "opt_itrs": np.array(opt_itrs), "steps": np.array(steps),
         "sigsq": np.array(sigsq), "idcs_order": alg.idcs.astype(np.int64), "wts_order": alg.wts.copy(),
         "get_wts": wts, "get_idcs": idcs.astype(np.int64)}
    for i, (a, b) in enumerate(zip(hist_idcs, hist_wts)):
        g["step%d_idcs" % i], g["step%d_wts" % i] = a.astype(np.int64), b
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
