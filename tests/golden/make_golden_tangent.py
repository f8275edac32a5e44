#!/usr/bin/env python3
"""Golden vectors F14: the REFERENCE's exact tangent-space projector of the linear-regression experiment -- the class
`LinRegProjector` that examples/linear_regression/main.py defines inside run() (:158-185) -- on a small seeded data set: its
projections at the prior (update(None, None)), at a weighted coreset posterior, and the weights / selections of the
reference's HilbertCoreset (GIGA) and SparseVICoreset built on it.  The class lives in a function body, so its text is read
from the reference tree AT GENERATION TIME and executed with the closure variables it expects (nothing of it is stored);
the fixture holds inputs and outputs only.
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tangent.py"""
import argparse
import os
import sys
import textwrap

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.path.insert(1, "/root/reference/examples/common")
import bayesiancoresets as bc  # noqa: E402 (reference)
import model_linreg  # noqa: E402 (reference example model)

OUT = os.path.join(HERE, "tangent_golden.npz")
REF_MAIN = "/root/reference/examples/linear_regression/main.py"


def reference_class(mu0, Sig0, Sig0inv, datastd, proj_dim):
    lines = open(REF_MAIN).read().splitlines()
    a = next(i for i, l in enumerate(lines) if l.strip().startswith("class LinRegProjector("))
    b = next(i for i, l in enumerate(lines) if i > a and l.strip().startswith("prj_optimal_exact"))
    ns = dict(bc=bc, np=np, model_linreg=model_linreg, mu0=mu0, Sig0=Sig0, Sig0inv=Sig0inv, datastd=datastd,
              arguments=argparse.Namespace(proj_dim=proj_dim))
    exec(textwrap.dedent("\n".join(lines[a:b])), ns)
    return ns["LinRegProjector"]


def main():
    rs = np.random.RandomState(14)
    N, D, p = 600, 7, 4
    X = rs.randn(N, D) * np.array([1.0, 0.5, 2.0, 1.0, 0.3, 1.5, 1.0])
    theta = rs.randn(D)
    datastd = 0.7
    Y = X.dot(theta) + datastd * rs.randn(N)
    Z = np.hstack((X, Y[:, None]))
    mu0 = 0.3 * np.ones(D)
    Sig0 = 1.7 * np.eye(D)
    Sig0inv = np.linalg.inv(Sig0)
    bV = np.linalg.eigh(X.T.dot(X))[1][:, -p:]                      # main.py:110-111
    cls = reference_class(mu0, Sig0, Sig0inv, datastd, p)
    prj = cls(bV)
    prj.update(None, None)
    v_prior = prj.project(Z)
    idx = np.array([5, 77, 300, 431])
    w = np.array([120.0, 0.0, 250.5, 33.0])
    prj.update(w, Z[idx])
    v_core = prj.project(Z)
    out = dict(Z=Z, mu0=mu0, Sig0=Sig0, datastd=datastd, bV=bV, idx=idx, w=w, v_prior=v_prior, v_core=v_core)
    # GIGA-OPT-EXACT as main.py:189-197 builds it: tangent space at the full-data posterior
    popt = cls(bV)
    popt.update(np.ones(N), Z)
    h = bc.HilbertCoreset(Z, popt)
    h.build(12)
    wts, pts, idcs = h.get()
    out.update(giga_wts=wts, giga_idcs=idcs, giga_err=h.error())
    # SVI-EXACT (main.py:191): fresh projector, updated by the coreset at every step
    np.random.seed(3)
    s = bc.SparseVICoreset(Z, cls(bV), opt_itrs=15, step_sched=lambda i: 1.0 / (1.0 + i))
    s.build(6)
    wts, pts, idcs = s.get()
    out.update(svi_wts=wts, svi_idcs=idcs)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "GIGA idcs", out["giga_idcs"], "SVI idcs", out["svi_idcs"])


if __name__ == "__main__":
    main()
