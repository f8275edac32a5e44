#!/usr/bin/env python3
"""F3 fixtures for seeds 4 and 5 of the config-1 harness (SURVEY.md section 8c asks for seeds 1-5; snnls_golden.npz holds
1-3), produced by the REFERENCE itself (imported read-only from /root/reference) exactly as make_golden.py does:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_harness45.py

Stores the Ms schedule, SHA-256 digests of the seeded inputs and the reference's per-M coreset size / error, final points and
weights -- outputs only, no reference source."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
import bayesiancoresets as bc  # noqa: E402  (the reference)

ALGS = {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe, "omp": bc.snnls.OrthoPursuit}
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness45_golden.npz")


class IDProjector(bc.Projector):
    def update(self, wts, pts):
        pass

    def project(self, pts, grad=False):
        return pts


def main():
    g = {}
    Ms = np.unique(np.logspace(0.0, np.log10(1000), 50, dtype=np.int32))     # examples/synthetic_vectors/main.py:51-54
    g["F3_Ms"] = Ms
    for trial in (4, 5):
        np.random.seed(trial)
        X = np.random.randn(10000, 100)
        g["F3_t%d_input_sha256" % trial] = np.array(hashlib.sha256(np.ascontiguousarray(X).tobytes()).hexdigest())
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
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "with", len(g), "arrays,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
