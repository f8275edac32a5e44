#!/usr/bin/env python3
"""Secondary measurement (BASELINE.json configs[4] shape, scaled to one GPU): SparseVICoreset greedy
steps on synthetic Gaussian linear regression with the fused device projection.  Prints one JSON line.
    python tools/svi_bench.py [--rows 1000000 --dim 300 --samples 256 --opt-itrs 20 --steps 3]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=300)
    ap.add_argument("--samples", type=int, default=256)
    ap.add_argument("--opt-itrs", type=int, default=20)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--device-sampler", action="store_true", help="draw the posterior samples on the GPU (examples/common/model_linreg.py)")
    a = ap.parse_args()
    import torch
    import bayesiancoresets_amd as bc
    from models import linreg_sampler
    N, D, S = a.rows, a.dim, a.samples
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    X = torch.randn(N, D, device="cuda", dtype=torch.float64, generator=g)
    th = torch.randn(D, device="cuda", dtype=torch.float64, generator=g)
    y = X @ th + torch.randn(N, device="cuda", dtype=torch.float64, generator=g)
    Zd = torch.cat((X, y[:, None]), dim=1).contiguous()
    del X
    np.random.seed(2)
    if a.device_sampler:
        sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
        from model_linreg import posterior_sampler
        smp = posterior_sampler(np.zeros(D), np.eye(D), 1.0, device="cuda", seed=2)
    else:
        smp = linreg_sampler(np.zeros(D), np.eye(D), 1.0)
    prj = bc.DeviceProjector("linreg", smp, S, sigsq=1.0)

    class DevData(object):      # device-resident data set with ndarray-style row access for the coreset points
        shape = (N, D + 1)
        def __getitem__(self, i):
            return Zd[i].cpu().numpy()
    data = DevData()
    prj._dev = lambda pts, _orig=prj._dev: Zd if pts is data else _orig(pts)
    alg = bc.SparseVICoreset(data, prj, opt_itrs=a.opt_itrs)
    alg.build(1)   # warm-up step
    # time the projection kernels alone
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    col = torch.empty(S, dtype=torch.float64, device="cuda")
    args = prj._common(Zd)
    e0.record()
    for _ in range(reps):
        prj._check(prj._lib.bcx_project_colsum(*args, col.data_ptr(), prj._workspace(S).data_ptr()))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    flops = 2.0 * N * D * S
    torch.cuda.synchronize(); t0 = time.perf_counter()
    alg.build(a.steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nproj = a.steps * (1 + a.opt_itrs)
    print(json.dumps({
        "metric": "SparseVI greedy steps/sec (N=%d, D=%d, S=%d, opt_itrs=%d)" % (N, D, S, a.opt_itrs),
        "value": a.steps / dt, "unit": "steps/s", "n_gpus": 1, "steps": a.steps, "s_per_step": dt / a.steps,
        "full_data_projections_per_step": 1 + a.opt_itrs, "ms_per_projection_end_to_end": dt / nproj * 1e3,
        "dtype": "f64", "data": "synthetic", "sampler": "device (torch)" if a.device_sampler else "host (NumPy/SciPy)",
        "roofline": {"bound": "mfma", "kernel": "proj_kernel<linreg, colsum>", "achieved": flops / (ms * 1e-3) / 1e12,
                     "peak": 78.6, "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / 78.6,
                     "avg_launch_ms": ms, "flops_per_launch": flops},
        "coreset": {"idcs": [int(i) for i in alg.idcs], "wts": [float(w) for w in alg.wts]},
    }))

if __name__ == "__main__":
    main()
