"""dev / profiling: the fused projection kernel alone at one shape, many launches (for rocprofv3 kernel traces and
PMC passes).  usage: python tools/proj_shape.py [--rows 625000 --dim 301 --samples 256 --mode colsum|select|write --reps 30]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
import bayesiancoresets_amd as bc

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=625000)
ap.add_argument("--dim", type=int, default=301)
ap.add_argument("--samples", type=int, default=256)
ap.add_argument("--family", default="linreg")
ap.add_argument("--mode", default="colsum")
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
rs = np.random.RandomState(0)
cols = a.dim if a.family == "logistic" else a.dim + 1
# even leading dimension (16-byte aligned rows), as DeviceProjector._dev lays host arrays out
Z = torch.randn(a.rows, cols + (cols % 2), dtype=torch.float64, device="cuda")[:, :cols]
if a.family == "poisson":
    Z[:, -1] = torch.poisson(torch.ones(a.rows, dtype=torch.float64, device="cuda"))
theta = 0.1 * rs.randn(a.samples, a.dim)
prj = bc.DeviceProjector(a.family, lambda n, w, p: theta, a.samples, sigsq=1.0)
prj.profile(True)
resid = rs.randn(a.samples)
lib, S = prj._lib, a.samples
col = torch.empty(S, dtype=torch.float64, device="cuda")
res = torch.empty(2, dtype=torch.float64, device="cuda")
r = torch.from_numpy(resid).cuda()
out = torch.empty((a.rows, S), dtype=torch.float64, device="cuda") if a.mode.startswith("write") else None
swork = prj._select_scratch(a.rows, S)
rowsum = torch.empty(a.rows, dtype=torch.float64, device="cuda")
work = prj._workspace(S)


def launch():
    if a.mode == "colsum":
        prj._check(lib.bcx_project_colsum(*prj._common(Z), col.data_ptr(), work.data_ptr()))
    elif a.mode == "select":
        prj._check(lib.bcx_project_select_ws(*prj._common(Z), r.data_ptr(), float(resid.sum()), res.data_ptr(), swork.data_ptr(),
                                             swork.numel() * 8))
    elif a.mode == "write_raw":
        prj._check(lib.bcx_project_write_raw(*prj._common(Z), out.data_ptr(), S))
    else:
        prj._check(lib.bcx_project_write(*prj._common(Z), out.data_ptr(), S, rowsum.data_ptr()))


for _ in range(3):
    launch()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.reps):
    launch()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.reps
fl = 2.0 * a.rows * a.dim * S
kms, kn, kfl = prj.profile_read()          # hipEvents around proj_kernel alone (3 warm-up + reps launches)
print("%s %s N=%d D=%d S=%d: %.3f ms per call (all kernels of the call) = %.1f TFLOP/s; proj_kernel alone %.3f ms = %.1f TFLOP/s fp64"
      % (a.family, a.mode, a.rows, a.dim, S, ms, fl / ms / 1e9, kms / max(kn, 1), kfl / max(kms, 1e-9) / 1e9), flush=True)
