#!/usr/bin/env python3
"""Logistic-regression coreset end to end on the GPU (workload of the reference's
examples/simple_lr/main.py): synthetic data, Laplace approximation at the MAP, S Monte-Carlo samples,
log-likelihood projection on the device (DeviceProjector "logistic"), greedy Hilbert coreset.

    python main.py --rows 1000000 --samples 512 --alg OMP --size 512
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(1, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "common"))
import bayesiancoresets_amd as bc  # noqa: E402
import model_lr  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100000)
    ap.add_argument("--dim", type=int, default=10)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--alg", default="GIGA", choices=["FW", "GIGA", "OMP"])
    ap.add_argument("--size", type=int, default=500)
    a = ap.parse_args()
    np.random.seed(1)
    N, D = a.rows, a.dim
    Z = model_lr.synthetic_rows(N, D, np.random)                 # simple_lr/main.py:22-35
    mu, cov = model_lr.laplace_fit(Z, device="cuda")             # MAP + inverse negative Hessian (main.py:57-63)
    sampler = lambda n, w, p: np.atleast_2d(np.random.multivariate_normal(mu, cov, n))
    algs = {"FW": bc.snnls.FrankWolfe, "GIGA": bc.snnls.GIGA, "OMP": bc.snnls.OrthoPursuit}
    t0 = time.perf_counter()
    prj = bc.DeviceProjector("logistic", sampler, a.samples)
    coreset = bc.HilbertCoreset(Z, prj, snnls=algs[a.alg])
    t1 = time.perf_counter()
    coreset.build(a.size)
    wts, pts, idcs = coreset.get()
    t2 = time.perf_counter()
    print("N=%d D=%d S=%d %s: projection+ingest %.3f s, %d greedy iterations %.3f s, coreset size %d, error %.6g"
          % (N, D, a.samples, a.alg, t1 - t0, a.size, t2 - t1, len(wts), coreset.error()))


if __name__ == "__main__":
    main()
