"""dev: does the SparseVI select projection (N = 5M, D = 301, S = 256) run slower right after a phase of small kernels (the ADAM
loop) or an idle gap than back to back?  hipEvents around each call."""
import os, sys, time
os.environ.setdefault("BCX_DEV", "1")
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
import bayesiancoresets_amd as bc
N, D, S = 5_000_000, 301, 256
Z = torch.randn(N, D + 1, dtype=torch.float64, device="cuda")
theta = 0.05 * np.random.RandomState(0).randn(S, D)
prj = bc.DeviceProjector("linreg", lambda n, w, p: theta, S, sigsq=1.0)
resid = np.random.RandomState(1).randn(S)
small = torch.zeros(256, device="cuda")


def sel():
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    prj.project_select(Z, resid)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)


for _ in range(3):
    sel()
print("back to back:          ", " ".join("%.2f" % sel() for _ in range(5)), "ms")
for gap in (0.005, 0.05, 0.5):
    out = []
    for _ in range(4):
        time.sleep(gap)
        out.append(sel())
    print("after %3.0f ms idle:      " % (gap * 1e3), " ".join("%.2f" % v for v in out), "ms")
out = []
for _ in range(4):
    for _ in range(400):
        small.add_(1.0)
    out.append(sel())
print("after 400 tiny kernels:", " ".join("%.2f" % v for v in out), "ms")
out = []
for _ in range(4):
    prj.project_colsum(Z)
    out.append(sel())
print("after a column-sum projection:", " ".join("%.2f" % v for v in out), "ms")
