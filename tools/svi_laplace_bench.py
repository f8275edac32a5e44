#!/usr/bin/env python3
"""dev: SparseVI on the logistic / Poisson regression model (examples/logistic_poisson_regression) with the Laplace sampler on
the device (bc.LaplacePosteriorSampler, csrc/laplace.hip): seconds per greedy step with the ADAM loop enqueued against the
host loop (the same device sampler called from the host every step) and against the host Laplace fit of the reference's
sequence.   python tools/svi_laplace_bench.py [--family logistic --rows 1000000 --dim 10 --samples 512 --steps 6]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="logistic")
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=10)
    ap.add_argument("--samples", type=int, default=512)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--opt-itrs", type=int, default=100)
    a = ap.parse_args()
    import torch
    import bayesiancoresets_amd as bc
    import model_lr
    import model_poiss
    mod = model_lr if a.family == "logistic" else model_poiss
    rs = np.random.RandomState(1)
    Z = mod.synthetic_rows(a.rows, a.dim, rs)
    Zd = torch.from_numpy(Z).cuda()
    out = {"family": a.family, "rows": a.rows, "D": a.dim, "S": a.samples, "opt_itrs": a.opt_itrs}

    def host_sampler(n, wts, pts):
        if wts is None or pts is None or np.asarray(pts).shape[0] == 0 or not (np.asarray(wts) > 0).any():
            return np.random.randn(n, a.dim)
        keep = np.asarray(wts) > 0
        mu, Sig = mod.laplace_fit(np.atleast_2d(pts)[keep], np.asarray(wts)[keep])
        return np.atleast_2d(np.random.multivariate_normal(mu, Sig, n))

    for name, make, enq in (("enqueued", lambda: bc.LaplacePosteriorSampler(a.family, a.dim, seed=2), True),
                            ("device_sampler_host_loop", lambda: bc.LaplacePosteriorSampler(a.family, a.dim, seed=2), False),
                            ("host_sampler_host_loop", lambda: host_sampler, False)):
        np.random.seed(3)
        prj = bc.DeviceProjector(a.family, make(), a.samples)

        class DevData(object):
            shape = Z.shape
            def __getitem__(self, i):
                return Z[i]
        data = DevData()
        prj._dev = lambda pts, _orig=prj._dev: Zd if pts is data else _orig(pts)
        alg = bc.SparseVICoreset(data, prj, opt_itrs=a.opt_itrs)
        alg.ENQUEUE = enq
        alg.build(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        alg.build(a.steps)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        out[name] = {"s_per_greedy_step": dt, "adam_step_us_incl_projection": dt / a.opt_itrs * 1e6, "points": int(alg.wts.shape[0])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
