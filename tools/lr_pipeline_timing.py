"""dev: wall-clock pieces of the config-3 pipeline (device projection -> HilbertCoreset constructor -> build)."""
import os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
import torch
import bayesiancoresets_amd as bc
def T(): torch.cuda.synchronize(); return time.perf_counter()
N, D, S = 1_000_000, 10, 512
g = torch.Generator(device="cuda"); g.manual_seed(1)
Z = torch.randn(N, D, dtype=torch.float64, device="cuda", generator=g)
theta = np.random.RandomState(0).randn(S, D) * 0.3
torch.zeros(1, device="cuda")
for rep in range(3):
    t0 = T()
    prj = bc.DeviceProjector("logistic", lambda n, w, p: theta, S)
    t1 = T()
    vecs = prj.project(Z)
    t2 = T()
    s = bc.snnls.OrthoPursuit(vecs.t(), None)
    t3 = T()
    s.build(100)
    t4 = T()
    print("rep %d: projector %.3f s, project %.3f s, solver constructor %.3f s, build(100) %.3f s" % (rep, t1 - t0, t2 - t1, t3 - t2, t4 - t3), flush=True)
    del s, vecs, prj
