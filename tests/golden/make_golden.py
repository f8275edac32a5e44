#!/usr/bin/env python3
"""Generate golden vectors by running the *reference itself* (imported read-only
from /root/reference) on seeded synthetic inputs.  Run in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Only inputs' seeds/shapes, SHA-256 digests of the input bytes and the
reference's OUTPUTS (selection order, per-iteration error, final weights) are
stored -- no reference source.  The reference never travels to the GPU box;
tests read ``snnls_golden.npz`` only.

Fixture ids follow SURVEY.md section 8c:
  F1  axis (X = eye(N)) N=100, 12 and 100 iterations       -> tie-breaking
  F2  normal seed 1, N=10k, d=100, 100 iterations            -> main parity case
  F3  config-1 harness (Ms schedule) seeds 1..3              -> csize / err per M
  F4  numeric-limit behaviour on the F2 input                -> latch semantics
  F7  optimize() before/after on the F2 state
  F9  normal seed 7, N=3000, d=64 (small, fast GPU/oracle case), 60 iterations
"""
import hashlib
import os
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
import bayesiancoresets as bc  # noqa: E402  (the reference)

ALGS = {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe, "omp": bc.snnls.OrthoPursuit}
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "snnls_golden.npz")


def digest(x):
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


class IDProjector(bc.Projector):
    def update(self, wts, pts):
        pass

    def project(self, pts, grad=False):
        return pts


def traced_build(solver, itrs):
    """Run solver.build(itrs) once, recording every _select() result and the
    error after every completed _reweight()."""
    sel, err = [], []
    orig_select, orig_reweight = solver._select, solver._reweight

    def select():
        f = orig_select()
        sel.append(int(f))
        return f

    def reweight(f):
        orig_reweight(f)
        err.append(float(solver.error()))

    solver._select, solver._reweight = select, reweight
    try:
        solver.build(itrs)
    finally:
        solver._select, solver._reweight = orig_select, orig_reweight
    return np.array(sel, dtype=np.int64), np.array(err)


def sparse(w):
    idx = np.flatnonzero(w > 0)
    return idx.astype(np.int64), w[idx]


def main():
    g = {}
    # ---------------- F1: axis -------------------------------------------
    X = np.eye(100)
    for name, cls in ALGS.items():
        for itrs in (12, 100):
            s = cls(X.T, X.sum(axis=0))
            sel, err = traced_build(s, itrs)
            idx, w = sparse(s.weights())
            k = "F1_%s_%d_" % (name, itrs)
            g[k + "sel"], g[k + "err"], g[k + "idx"], g[k + "w"] = sel, err, idx, w
            g[k + "final_err"] = np.array(s.error())
    # ---------------- F2 / F4 / F7: normal seed 1, N=10k, d=100 ----------
    np.random.seed(1)
    X = np.random.randn(10000, 100)
    g["F2_input_sha256"] = np.array(digest(X))
    for name, cls in ALGS.items():
        s = cls(X.T, X.sum(axis=0))
        sel, err = traced_build(s, 100)
        idx, w = sparse(s.weights())
        k = "F2_%s_" % name
        g[k + "sel"], g[k + "err"], g[k + "idx"], g[k + "w"] = sel, err, idx, w
        g[k + "final_err"] = np.array(s.error())
        # F7: optimize() on that state
        s.optimize()
        idx, w = sparse(s.weights())
        k = "F7_%s_" % name
        g[k + "idx"], g[k + "w"] = idx, w
        g[k + "final_err"] = np.array(s.error())
        g[k + "limit"] = np.array(bool(s.reached_numeric_limit))
    # F4: drive into the numeric limit
    for name, itrs in (("giga", 2000), ("fw", 400), ("omp", 140)):
        s = ALGS[name](X.T, X.sum(axis=0))
        sel, err = traced_build(s, itrs)
        idx, w = sparse(s.weights())
        k = "F4_%s_" % name
        g[k + "itrs"] = np.array(itrs)
        g[k + "sel"], g[k + "err"], g[k + "idx"], g[k + "w"] = sel, err, idx, w
        g[k + "final_err"] = np.array(s.error())
        g[k + "size"] = np.array(int(s.size()))
        g[k + "limit"] = np.array(bool(s.reached_numeric_limit))
        g[k + "n_select_calls"] = np.array(len(sel))
    # ---------------- F3: config-1 harness --------------------------------
    Ms = np.unique(np.logspace(0.0, np.log10(1000), 50, dtype=np.int32))
    g["F3_Ms"] = Ms
    for trial in (1, 2, 3):
        np.random.seed(trial)
        X = np.random.randn(10000, 100)
        g["F3_t%d_input_sha256" % trial] = np.array(digest(X))
        for name, cls in ALGS.items():
            alg = bc.HilbertCoreset(X, IDProjector(), snnls=cls)
            csize, err = np.zeros(Ms.shape[0]), np.zeros(Ms.shape[0])
            for m in range(Ms.shape[0]):
                alg.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
                wts, pts, idcs = alg.get()
                csize[m] = (wts > 0).sum()
                err[m] = alg.error()
            k = "F3_t%d_%s_" % (trial, name)
            g[k + "csize"], g[k + "err"] = csize, err
            g[k + "limit"] = np.array(bool(alg.snnls.reached_numeric_limit))
            wts, pts, idcs = alg.get()
            g[k + "idcs"], g[k + "wts"] = idcs.astype(np.int64), wts
    # ---------------- F9: small fast case ---------------------------------
    np.random.seed(7)
    X = np.random.randn(3000, 64)
    g["F9_input_sha256"] = np.array(digest(X))
    for name, cls in ALGS.items():
        s = cls(X.T, X.sum(axis=0))
        sel, err = traced_build(s, 60)
        idx, w = sparse(s.weights())
        k = "F9_%s_" % name
        g[k + "sel"], g[k + "err"], g[k + "idx"], g[k + "w"] = sel, err, idx, w
        g[k + "final_err"] = np.array(s.error())
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "with", len(g), "arrays,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
