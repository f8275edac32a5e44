"""dev: where does the host time of a SparseVI greedy step go? (cProfile around build() only)"""
import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import bayesiancoresets_amd as bc
from models import linreg_sampler
N, D, S, OPT = 625000, 301, 256, 30
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.randn(N, D, device="cuda", dtype=torch.float64, generator=g)
th = torch.randn(D, device="cuda", dtype=torch.float64, generator=g)
y = X @ th + torch.randn(N, device="cuda", dtype=torch.float64, generator=g)
Zd = torch.cat((X, y[:, None]), dim=1).contiguous(); del X
np.random.seed(2)
prj = bc.DeviceProjector("linreg", linreg_sampler(np.zeros(D), np.eye(D), 1.0), S, sigsq=1.0)
class DevData(object):
    shape = (N, D + 1)
    def __getitem__(self, i): return Zd[i].cpu().numpy()
data = DevData()
prj._dev = lambda pts, _orig=prj._dev: Zd if pts is data else _orig(pts)
alg = bc.SparseVICoreset(data, prj, opt_itrs=OPT)
alg.build(1)
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
alg.build(1)
pr.disable(); print("step %.3f s" % (time.perf_counter() - t0))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:5000])
