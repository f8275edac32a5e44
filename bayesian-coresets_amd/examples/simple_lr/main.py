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
from scipy.optimize import minimize

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bayesiancoresets_amd as bc  # noqa: E402


def log_joint_and_grad(Z, th, w):
    m = -Z.dot(th)
    ll = np.where(m < 100, -np.log1p(np.exp(np.minimum(m, 100))), -m)
    sig = np.where(m < 100, np.exp(np.minimum(m, 100)) / (1.0 + np.exp(np.minimum(m, 100))), 1.0)
    val = (w * ll).sum() - 0.5 * th.shape[0] * np.log(2 * np.pi) - 0.5 * (th ** 2).sum()
    grad = ((w * sig)[:, None] * Z).sum(axis=0) - th
    return val, grad, sig


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
    X = np.random.randn(N, D)
    th_true = 3.0 * np.ones(D)
    y = (np.random.rand(N) <= 1.0 / (1.0 + np.exp(-X.dot(th_true)))).astype(int)
    y[y == 0] = -1
    Z = y[:, None] * X
    ones = np.ones(N)
    res = minimize(lambda t: -log_joint_and_grad(Z, t, ones)[0], Z.mean(axis=0),
                   jac=lambda t: -log_joint_and_grad(Z, t, ones)[1])
    mu = res.x
    sig = log_joint_and_grad(Z, mu, ones)[2]
    H = (Z * (sig * (1 - sig))[:, None]).T.dot(Z) + np.eye(D)        # negative Hessian of the log joint
    cov = np.linalg.inv(H)
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
