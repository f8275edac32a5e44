#!/usr/bin/env python3
"""Golden vectors F16: the reference's Gaussian experiment -- model_gaussian (log-likelihood, weighted posterior) and the
exact tangent-space projector `GaussianProjector` that examples/gaussian/main.py defines inside run() (:117-138) -- on a small
seeded data set, plus the reference's HilbertCoreset (GIGA) and SparseVICoreset runs on that projector.  The class text is
read from the reference tree AT GENERATION TIME and executed with the closure variables it expects (nothing of it is
stored); the fixture holds inputs and outputs only.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_gaussian.py"""
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, "/root/reference/examples/common")
import bayesiancoresets as bc  # noqa: E402 (reference)
import model_gaussian as gaussian  # noqa: E402 (reference example model)

OUT = os.path.join(HERE, "gaussian_golden.npz")
REF_MAIN = "/root/reference/examples/gaussian/main.py"


def reference_class(mu0, Sig0inv, Siginv, LSigInv):
    lines = open(REF_MAIN).read().splitlines()
    a = next(i for i, l in enumerate(lines) if l.strip().startswith("class GaussianProjector("))
    b = next(i for i, l in enumerate(lines) if i > a and l.strip().startswith("prj_optimal_exact"))
    ns = dict(bc=bc, np=np, gaussian=gaussian, mu0=mu0, Sig0inv=Sig0inv, Siginv=Siginv, LSigInv=LSigInv)
    exec(textwrap.dedent("\n".join(lines[a:b])), ns)
    return ns["GaussianProjector"]


def main():
    rs = np.random.RandomState(16)
    N, D = 500, 6
    A = rs.randn(D, D)
    Sig = A.dot(A.T) / D + 0.5 * np.eye(D)                      # a non-trivial known covariance
    Siginv = np.linalg.inv(Sig)
    LSigInv = np.linalg.cholesky(Siginv)
    mu0, Sig0inv = 0.2 * rs.randn(D), np.diag(rs.uniform(0.5, 2.0, D))
    x = rs.multivariate_normal(np.ones(D), Sig, N)
    th = rs.randn(5, D)
    out = dict(x=x, Sig=Sig, mu0=mu0, Sig0inv=Sig0inv, th=th,
               ll=gaussian.log_likelihood(x, th, Siginv, np.linalg.slogdet(Sig)[1]))
    idx = np.array([3, 100, 250, 499])
    w = np.array([40.0, 7.5, 0.0, 120.0])
    mu, U, _ = gaussian.weighted_post(mu0, Sig0inv, Siginv, x[idx], w)
    out.update(idx=idx, w=w, post_mu=mu, post_cov=U.dot(U.T))
    cls = reference_class(mu0, Sig0inv, Siginv, LSigInv)
    prj = cls()
    prj.update()
    out["v_prior"] = prj.project(x)
    prj.update(w, x[idx])
    out["v_core"] = prj.project(x)
    popt = cls()
    popt.update(np.ones(N), x)
    h = bc.HilbertCoreset(x, popt)
    h.build(10)
    wts, pts, idcs = h.get()
    out.update(giga_wts=wts, giga_idcs=idcs, giga_err=h.error())
    np.random.seed(5)
    s = bc.SparseVICoreset(x, cls(), opt_itrs=12, step_sched=lambda i: 1.0 / (1.0 + i))
    s.build(5)
    wts, pts, idcs = s.get()
    out.update(svi_wts=wts, svi_idcs=idcs)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "GIGA idcs", out["giga_idcs"], "SVI idcs", out["svi_idcs"])


if __name__ == "__main__":
    main()
