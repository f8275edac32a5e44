"""dev: time the fused projection kernels (csrc/proj.hip)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
import bayesiancoresets_amd as bc

def run(family, N, D, S, reps=20):
    rs = np.random.RandomState(0)
    cols = D if family == "logistic" else D + 1
    Z = torch.randn(N, cols, dtype=torch.float64, device="cuda")
    if family == "poisson":
        Z[:, -1] = torch.poisson(torch.ones(N, dtype=torch.float64, device="cuda"))
    theta = 0.1 * rs.randn(S, D)
    prj = bc.DeviceProjector(family, lambda n, w, p: theta, S, sigsq=1.0, colsum="mfma")      # (time the projection kernel, not the moments form)
    resid = rs.randn(S)
    for name, fn in (("colsum", lambda: prj.project_colsum(Z)), ("select", lambda: prj.project_select(Z, resid)),
                     ("write", lambda: prj.project(Z))):
        fn(); fn(); torch.cuda.synchronize()
        prj.profile(True)
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        kms, launches, flops = prj.profile_read()          # hipEvents around the projection kernel alone
        prj.profile(False)
        fl = 2.0 * N * D * S
        print("%-8s %-6s N=%d D=%d S=%d: wall %.2f ms  %.1f TFLOP/s | kernel %.3f ms  %.1f TFLOP/s  (%d launches)"
              % (family, name, N, D, S, dt * 1e3, fl / dt / 1e12, kms / max(launches, 1), flops / max(kms, 1e-9) / 1e9, launches), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        for spec in sys.argv[1:]:
            fam, n, dd, ss = spec.split(",")
            run(fam, int(n), int(dd), int(ss))
        sys.exit(0)
    run("linreg", 1000000, 300, 256)
    run("logistic", 1000000, 10, 512)
    run("poisson", 1000000, 16, 256)
    run("linreg", 1000000, 32, 64)
    run("linreg", 625000, 301, 256)     # BASELINE configs[4] per-GPU shard: N=5M/8, D=301 (300 bases + const), S=256
    run("linreg", 1000000, 304, 256)
    run("linreg", 1000000, 512, 512)
    run("logistic", 1000000, 100, 500)
