#!/usr/bin/env python3
"""Golden vectors for the config-3 style workload: Laplace-projected logistic
regression log-likelihood vectors, whose row norms span many decades.  Runs the REFERENCE
(bayesiancoresets + examples/common/model_lr.py, imported read-only from /root/reference) and
stores inputs that are not trivially regenerable (Laplace mean/cov, the S parameter samples)
plus the reference's outputs.  Run in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_lr.py
"""
import os
import sys

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, "/root/reference")
sys.path.insert(1, "/root/reference/examples/common")
import bayesiancoresets as bc  # noqa: E402 (reference)
import model_lr  # noqa: E402 (reference example model)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lr_golden.npz")


def make_data(seed, N, D):
    """simple_lr-style generator (examples/simple_lr/main.py:22-35) with cov = I drawn via randn."""
    rs = np.random.RandomState(seed)
    X = rs.randn(N, D)
    th = 3.0 * np.ones(D)
    ps = 1.0 / (1.0 + np.exp(-(X * th).sum(axis=1)))
    y = (rs.rand(N) <= ps).astype(int)
    y[y == 0] = -1
    return y[:, np.newaxis] * X


def main():
    N, D, S, itrs = 20000, 10, 128, 30
    Z = make_data(1, N, D)
    ones = np.ones(N)
    res = minimize(lambda mu: -model_lr.log_joint(Z, mu, ones)[0], Z.mean(axis=0),
                   jac=lambda mu: -model_lr.grad_th_log_joint(Z, mu, ones)[0, :])
    mu = res.x
    cov = -np.linalg.inv(model_lr.hess_th_log_joint(Z, mu, ones)[0, :, :])
    samples = np.random.RandomState(2).multivariate_normal(mu, cov, S)
    projector = bc.BlackBoxProjector(lambda sz, w, p: samples, S, model_lr.log_likelihood)
    vecs = projector.project(Z)
    norms = np.sqrt((vecs ** 2).sum(axis=1))
    g = {"N": np.array(N), "D": np.array(D), "S": np.array(S), "itrs": np.array(itrs), "mu": mu, "cov": cov,
         "samples": samples, "vecs_sum": vecs.sum(axis=0), "vecs_abs_sum": np.array(np.abs(vecs).sum()),
         "vecs_head": vecs[:4].copy(), "norm_min": np.array(norms.min()), "norm_max": np.array(norms.max()),
         "norm_median": np.array(np.median(norms))}
    algs = {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe, "omp": bc.snnls.OrthoPursuit}
    for name, cls in algs.items():
        alg = bc.HilbertCoreset(Z, projector, snnls=cls)
        sel = []
        orig = alg.snnls._select

        def select(orig=orig, sel=sel):
            f = orig()
            sel.append(int(f))
            return f

        alg.snnls._select = select
        alg.build(itrs)
        wts, pts, idcs = alg.get()
        g[name + "_sel"] = np.array(sel, dtype=np.int64)
        g[name + "_wts"], g[name + "_idcs"] = wts, idcs.astype(np.int64)
        g[name + "_err"] = np.array(alg.error())
        print(name, "size", len(wts), "err", alg.error())
    np.savez_compressed(OUT, **g)
    print("norms: min %.3g median %.3g max %.3g" % (norms.min(), np.median(norms), norms.max()))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
