"""dev: scan bandwidth across row widths (perf cliffs?)  usage: shape_sweep.py [--store f32|f16|f64] [d ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.gpu_quick import run
from bayesiancoresets_amd import _native as nat
if __name__ == "__main__":
    args = sys.argv[1:]
    store, esz = nat.F32, 4
    if args and args[0] == "--store":
        store, esz = {"f32": (nat.F32, 4), "f16": (nat.F16, 2), "f64": (nat.F64, 8)}[args[1]]
        args = args[2:]
    dims = [int(v) for v in args] or [8, 16, 17, 33, 64, 100, 128, 200, 256, 300, 512, 777, 1000, 1024, 2048, 4096, 5000, 8192]
    for d in dims:
        N = max(100000, int(2.0e9 / (esz * d)))
        for alg in (nat.ALG_FW, nat.ALG_GIGA):
            run(alg, N, d, store=store, iters=20)
