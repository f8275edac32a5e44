#!/usr/bin/env python3
"""Synthetic-vector experiment on the device engine: the workload of the reference's
examples/synthetic_vectors/main.py:31-107 (seeded randn(N, d) or eye(N) vectors, identity projector,
HilbertCoreset with FW / GIGA / OMP / uniform sampling, coreset built incrementally over a log- or
linearly-spaced size schedule) with results printed as CSV: M, csize, err, wall-clock seconds.

    python synthetic_vectors.py --alg GIGA --data_num 1000000 --data_dim 256 --trial 1
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bayesiancoresets_amd as bc  # noqa: E402


class IDProjector(bc.Projector):
    def update(self, wts, pts):
        pass

    def project(self, pts, grad=False):
        return pts


def main():
    ap = argparse.ArgumentParser("sparse non-negative regression on synthetic vectors")
    ap.add_argument("--alg", default="GIGA", choices=["FW", "GIGA", "OMP", "US"])
    ap.add_argument("--data_num", type=int, default=10000)
    ap.add_argument("--data_dim", type=int, default=100)
    ap.add_argument("--data_type", default="normal", choices=["normal", "axis"])
    ap.add_argument("--coreset_size_max", type=int, default=1000)
    ap.add_argument("--coreset_num_sizes", type=int, default=50)
    ap.add_argument("--coreset_size_spacing", default="log", choices=["log", "linear"])
    ap.add_argument("--trial", type=int, default=1)
    ap.add_argument("--verbosity", default="error", choices=["error", "warning", "critical", "info", "debug"])
    a = ap.parse_args()

    np.random.seed(a.trial)
    bc.util.set_verbosity(a.verbosity)
    algs = {"FW": bc.snnls.FrankWolfe, "GIGA": bc.snnls.GIGA, "OMP": bc.snnls.OrthoPursuit,
            "US": bc.snnls.UniformSampling}
    if a.coreset_size_spacing == "log":
        Ms = np.unique(np.logspace(0.0, np.log10(a.coreset_size_max), a.coreset_num_sizes, dtype=np.int32))
    else:
        Ms = np.unique(np.linspace(1, a.coreset_size_max, a.coreset_num_sizes, dtype=np.int32))
    X = np.random.randn(a.data_num, a.data_dim) if a.data_type == "normal" else np.eye(a.data_num)

    t0 = time.perf_counter()
    alg = bc.HilbertCoreset(X, IDProjector(), snnls=algs[a.alg])
    t_init = time.perf_counter() - t0
    print("# alg=%s N=%d d=%d trial=%d constructor %.3f s" % (a.alg, a.data_num, a.data_dim, a.trial, t_init))
    print("M,csize,err,wall_s")
    wall = 0.0
    for m, M in enumerate(Ms):
        t0 = time.perf_counter()
        alg.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
        wall += time.perf_counter() - t0
        wts, pts, idcs = alg.get()
        print("%d,%d,%.10g,%.4f" % (M, (wts > 0).sum(), alg.error(), wall))


if __name__ == "__main__":
    main()
