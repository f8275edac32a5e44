"""dev: scan bandwidth across row widths (perf cliffs?)  usage: shape_sweep.py [d ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools.gpu_quick import run
from bayesiancoresets_amd import _native as nat
if __name__ == "__main__":
    dims = [int(v) for v in sys.argv[1:]] or [8, 16, 33, 64, 100, 128, 200, 256, 300, 512, 777, 1000, 1024, 2048]
    for d in dims:
        N = max(100000, int(2.0e9 / (4 * d)))
        for alg in (nat.ALG_FW, nat.ALG_GIGA):
            run(alg, N, d, iters=20)
