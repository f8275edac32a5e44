"""dev: per-phase time stamps of the merged resolve+apply tail (library built with EXTRA=-DBCX_TIMING)."""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
from bayesiancoresets_amd import _native as nat
nat.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd", "lib_timing", "libbcx.so")   # tools/build_timing.sh

NAMES = ["entry", "partials+L*", "candidates", "rescored", "winner row", "slot/size", "step sums", "new state", "commit", "next query"]

def run(alg, N, d, its=300):
    eng = nat.Engine(alg, N, d)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn(N, d, device="cuda", dtype=torch.float64, generator=g)
    eng.load_device_rows(x.data_ptr(), N, d, True); torch.cuda.synchronize()
    assert eng.finalize(None) == 0
    eng.build_begin(its + 5, 1e-12); eng.enqueue(its); eng.poll()
    buf = (C.c_longlong * 32)()
    eng.lib.bcx_debug_stamps.argtypes = [C.c_void_p, C.c_void_p]
    eng.lib.bcx_debug_stamps(eng.h, buf)
    t = np.array(list(buf), dtype=np.int64)
    print("alg", alg, "N", N, "d", d, "k", its)
    for i in range(1, 10):
        print("  %-12s +%6.2f us (abs %6.2f)" % (NAMES[i], (t[i] - t[i - 1]) / 100.0, (t[i] - t[0]) / 100.0))

if __name__ == "__main__":
    run(nat.ALG_FW, 1000000, 256)
    run(nat.ALG_GIGA, 1000000, 256)
    run(nat.ALG_GIGA, 1000000, 512)
    run(nat.ALG_FW, 1250304, 512)
    run(nat.ALG_FW, 10000000, 512, its=60)
