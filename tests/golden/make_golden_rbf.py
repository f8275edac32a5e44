#!/usr/bin/env python3
"""Golden vectors F6b (config-5 workload as specified): the REFERENCE's SparseVICoreset with a
BlackBoxProjector (reference model_linreg likelihood + weighted posterior sampler, linear_regression/main.py:134-147)
on the synthetic RBF-basis regression of examples/common/rbf_workload.py (6 scales x 50 bases + 1 = 301 columns).
Also stores reference projections / correlations of one state so the fused SELECT / COLSUM kernels can be checked
against the reference's own centre-then-norm arithmetic (sparsevi.py:47-56) on this collinear design.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_rbf.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))   # OUR workload generator
import rbf_workload  # noqa: E402
sys.path.insert(0, "/root/reference")
sys.path.insert(1, "/root/reference/examples/common")
import bayesiancoresets as bc  # noqa: E402 (reference)
import model_linreg  # noqa: E402 (reference example model)

OUT = os.path.join(HERE, "rbf_golden.npz")


def main():
    N, nb, S, opt_itrs, steps = 50000, 50, 64, 20, 4
    wl = rbf_workload.make_rbf_regression(N, nb, seed=1)
    Z, mu0, Sig0, sigsq = wl["Z"], wl["mu0"], wl["Sig0"], wl["sigsq"]
    Sig0inv = np.linalg.inv(Sig0)

    def sampler_w(n, wts, pts):     # examples/linear_regression/main.py:141-147
        if wts is None or pts is None or pts.shape[0] == 0:
            muw, USigw = mu0, np.linalg.cholesky(Sig0)
        else:
            muw, USigw, _ = model_linreg.weighted_post(mu0, Sig0inv, sigsq, pts, wts)
        return muw + np.random.randn(n, muw.shape[0]).dot(USigw.T)

    def loglik(z, th):
        return model_linreg.log_likelihood(z, th, sigsq)

    np.random.seed(2)
    prj = bc.BlackBoxProjector(sampler_w, S, loglik)
    # one-state check vectors: prior samples (the first update), full projection, select arithmetic of sparsevi.py:47-56
    theta0 = prj.samples.copy()
    vecs = prj.project(Z)
    resid = vecs.sum(axis=0)
    corrs = vecs.dot(resid) / np.sqrt((vecs ** 2).sum(axis=1)) / vecs.shape[1]
    g = {"N": np.array(N), "nb": np.array(nb), "S": np.array(S), "opt_itrs": np.array(opt_itrs), "steps": np.array(steps),
         "sigsq": np.array(sigsq), "Z_sha": np.array(__import__("hashlib").sha256(Z.tobytes()).hexdigest()),
         "theta0": theta0, "colsum0": resid, "corr_argmax0": np.array(int(corrs.argmax())), "corr_max0": np.array(corrs.max()),
         "corr_head0": corrs[:4096].copy(), "rownorm_min0": np.array(np.sqrt((vecs ** 2).sum(axis=1)).min()),
         "rownorm_max0": np.array(np.sqrt((vecs ** 2).sum(axis=1)).max()),
         "ll_absmax0": np.array(np.abs(loglik(Z, theta0)).max())}
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=opt_itrs)
    for i in range(steps):
        alg.build(1)
        g["step%d_idcs" % i], g["step%d_wts" % i] = alg.idcs.astype(np.int64).copy(), alg.wts.copy()
        print("step", i, "idcs", alg.idcs, "wts", alg.wts)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    print("row norms of the centred projection: min %.3e max %.3e; |ll| max %.3e" % (g["rownorm_min0"], g["rownorm_max0"], g["ll_absmax0"]))


if __name__ == "__main__":
    main()
