"""dev: time optimize() (fp64-MFMA Gram + cold-start NNLS) after a greedy build."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
from bayesiancoresets_amd import _native as nat

def run(alg, N, d, its):
    eng = nat.Engine(alg, N, d)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn(N, d, device="cuda", dtype=torch.float64, generator=g)
    eng.load_device_rows(x.data_ptr(), N, d, True); torch.cuda.synchronize()
    assert eng.finalize(None) == 0
    eng.run_build(its, 1e-12)
    idx, w = eng.sparse_weights()
    e0 = eng.error()
    t0 = time.perf_counter()
    ok = eng.optimize(1e-12)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = len(idx)
    print("alg %d N=%d d=%d k=%d: optimize %.2f ms accepted=%s err %.6g -> %.10g  (Gram flops %.2e; refined-solve fallbacks %d)"
          % (alg, N, d, k, dt * 1e3, ok, e0, eng.error(), 2.0 * k * k * d, eng.omp_stats()["resolves"]))

if __name__ == "__main__":
    run(nat.ALG_FW, 1000000, 512, 400)
    run(nat.ALG_GIGA, 1000000, 512, 1000)
    run(nat.ALG_FW, 1000000, 1024, 1500)
