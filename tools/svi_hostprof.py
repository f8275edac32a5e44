"""dev: where does the host time of a SparseVI greedy step go (enqueued ADAM loop, closed-form column sums)?  cProfile around
build() only; the GPU work is small here (N = 625k) so that the host shows."""
import cProfile, pstats, io, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
import torch
import bayesiancoresets_amd as bc
N, D, S, OPT = 625000, 301, 256, 100
g = torch.Generator(device="cuda"); g.manual_seed(1)
X = torch.randn(N, D, device="cuda", dtype=torch.float64, generator=g)
th = torch.randn(D, device="cuda", dtype=torch.float64, generator=g)
y = X @ th + torch.randn(N, device="cuda", dtype=torch.float64, generator=g)
Zd = torch.cat((X, y[:, None]), dim=1).contiguous(); del X
np.random.seed(2)
smp = bc.LinregPosteriorSampler(np.zeros(D), 30.0 * np.eye(D), 1.0, seed=3)
prj = bc.DeviceProjector("linreg", smp, S, sigsq=1.0, colsum="moments")
alg = bc.SparseVICoreset(Zd, prj, opt_itrs=OPT)
alg.build(2)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); alg.build(1); torch.cuda.synchronize(); print("greedy step %.3f ms (coreset size %d)" % ((time.perf_counter() - t0) * 1e3, alg.size()))
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
alg.build(1)
pr.disable(); print("profiled step %.3f ms" % ((time.perf_counter() - t0) * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(32); print(s.getvalue()[:7000])
