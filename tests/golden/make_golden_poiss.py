#!/usr/bin/env python3
"""Golden vectors F15: the Laplace approximation the reference's logistic / Poisson regression experiment places its tangent
spaces at -- `get_laplace` of examples/logistic_poisson_regression/main.py:15-41 with the reference's model_poiss / model_lr
(log joint, gradient, Hessian) -- on small seeded data sets, unweighted and weighted.  The example module itself cannot be
imported (it pulls in pystan for its MCMC evaluation), so the text of that one function is read from the reference tree AT
GENERATION TIME and executed with the names it uses (nothing of it is stored); the fixture holds inputs and outputs only.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_poiss.py"""
import os
import sys

import numpy as np
from scipy.linalg import solve_triangular
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(1, "/root/reference/examples/common")
import model_lr as ref_lr  # noqa: E402 (reference)
import model_poiss as ref_poiss  # noqa: E402 (reference)

OUT = os.path.join(HERE, "poiss_golden.npz")
REF_MAIN = "/root/reference/examples/logistic_poisson_regression/main.py"


def reference_get_laplace():
    lines = open(REF_MAIN).read().splitlines()
    a = next(i for i, l in enumerate(lines) if l.startswith("def get_laplace("))
    b = next(i for i, l in enumerate(lines) if i > a and l.startswith("def "))
    ns = dict(np=np, minimize=minimize, solve_triangular=solve_triangular)
    exec("\n".join(lines[a:b]), ns)
    return ns["get_laplace"]


def main():
    get_laplace = reference_get_laplace()
    rs = np.random.RandomState(15)
    out = {}
    # Poisson: rows [x, 1, y]
    N, D = 900, 4
    X = np.hstack((rs.randn(N, D - 1), np.ones((N, 1))))
    th = np.array([0.7, -0.4, 0.3, 0.2])
    y = rs.poisson(np.log1p(np.exp(X.dot(th)))).astype(np.float64)
    Zp = np.hstack((X, y[:, None]))
    w = np.zeros(N)
    sel = rs.choice(N, 25, replace=False)
    w[sel] = rs.uniform(5.0, 60.0, 25)
    for tag, wts in (("poiss_full", np.ones(N)), ("poiss_wtd", w)):
        mu, LSig, LSigInv = get_laplace(wts, Zp, np.zeros(D), ref_poiss)
        out[tag + "_mu"] = mu
        out[tag + "_cov"] = LSig.T.dot(LSig)            # (L L^T)^-1 with L = chol(-H): the inverse negative Hessian
    out.update(poiss_Z=Zp, poiss_w=w)
    # logistic: rows y x
    Xl = np.hstack((rs.randn(N, 2), np.ones((N, 1))))
    p = 1.0 / (1.0 + np.exp(-Xl.dot(np.array([1.5, -1.0, 0.3]))))
    yl = np.where(rs.rand(N) <= p, 1.0, -1.0)
    Zl = yl[:, None] * Xl
    for tag, wts in (("lr_full", np.ones(N)), ("lr_wtd", w)):
        mu, LSig, LSigInv = get_laplace(wts, Zl, np.zeros(3), ref_lr)
        out[tag + "_mu"] = mu
        out[tag + "_cov"] = LSig.T.dot(LSig)
    out.update(lr_Z=Zl)
    # the likelihood itself on arguments that reach both branches of compute_s (model_poiss.py:25-38)
    zz = np.hstack((np.array([[-300.0], [-120.0], [-99.0], [-5.0], [0.0], [3.0], [40.0], [800.0]]), np.array([[0.0], [2.0], [1.0], [0.0], [4.0], [7.0], [30.0], [750.0]])))
    out.update(ll_z=zz, ll_th=np.array([[1.0], [0.5]]), ll=ref_poiss.log_likelihood(zz.copy(), np.array([[1.0], [0.5]])))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: np.round(v, 4) for k, v in out.items() if k.endswith("_mu")})


if __name__ == "__main__":
    main()
