"""dev: per-phase time stamps of the fused OMP step (library built with EXTRA=-DBCX_TIMING)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
from bayesiancoresets_amd import _native as nat

NAMES = ["entry", "rows", "B1", "decide", "u=Hg", "B2", "step", "combine", "rank-1", "B4", "finish", "next query"]

def run(N, d, its):
    eng = nat.Engine(nat.ALG_OMP, N, d)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn(N, d, device="cuda", dtype=torch.float64, generator=g)
    eng.load_device_rows(x.data_ptr(), N, d, True); torch.cuda.synchronize()
    assert eng.finalize(None) == 0
    eng.build_begin(its + 5, 1e-12); eng.enqueue(its); eng.poll()
    buf = (C.c_longlong * 32)()
    eng.lib.bcx_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    eng.lib.bcx_debug_stamps(eng.h, buf)
    t = np.array(list(buf), dtype=np.int64)
    print("OMP N", N, "d", d, "k", its)
    for i in range(1, 12):
        print("  %-12s +%6.2f us (abs %6.2f)" % (NAMES[i], (t[i] - t[i - 1]) / 100.0, (t[i] - t[0]) / 100.0))

if __name__ == "__main__":
    run(200000, 512, 100)
    run(200000, 512, 400)
