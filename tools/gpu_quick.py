"""Quick device check: scan-kernel bandwidth at a few shapes (dev tool, not part of the product)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
from bayesiancoresets_amd import _native as nat

def run(alg, N, d, store=nat.F32, keep=False, iters=50):
    eng = nat.Engine(alg, N, d, store_dtype=store, keep_exact_rows=keep)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    step = 1 << 18
    for r0 in range(0, N, step):
        m = min(step, N - r0)
        x = torch.randn(m, d, device="cuda", dtype=torch.float64, generator=g)
        eng.load_device_rows(x.data_ptr(), m, d, True, row_begin=r0)
        torch.cuda.synchronize()
    rc = eng.finalize(None)
    assert rc == 0, rc
    assert not eng.build_begin(iters, 1e-12)
    ms, by = eng.time_scan(20)
    t0 = time.perf_counter()
    eng.enqueue(iters)
    done, ne, lim = eng.poll()
    t1 = time.perf_counter()
    sel, err, st = eng.trace(iters)
    print("alg=%d N=%d d=%d store=%d keep=%d: scan %.3f ms -> %.1f GB/s (%.1f%% of 8TB/s); build %d it in %.1f ms = %.3f ms/it; need_exact=%s done=%d err[-1]=%.6g status_bad=%d"
          % (alg, N, d, store, keep, ms, by / ms / 1e6, by / ms / 1e6 / 80.0, iters, (t1 - t0) * 1e3, (t1 - t0) * 1e3 / iters, ne, done, err[-1] if len(err) else -1, int((st != 0).sum())), flush=True)
    eng.close()

if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep":
    run(nat.ALG_GIGA, 1000000, 256, iters=20)
    run(nat.ALG_FW, 1250000, 512, iters=20)
    run(nat.ALG_FW, 4000000, 512, iters=20)
    sys.exit(0)

if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    run(nat.ALG_FW, 1000000, 256)
    run(nat.ALG_GIGA, 1000000, 256)
    run(nat.ALG_FW, 1000000, 512)
    run(nat.ALG_GIGA, 1000000, 512)
    run(nat.ALG_GIGA, 1000000, 256, keep=True)
    run(nat.ALG_FW, 4000000, 512)
    run(nat.ALG_GIGA, 1000000, 256, store=nat.F64)
