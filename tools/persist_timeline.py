"""Dev tool: where an iteration of the several-iterations-per-launch form (csrc/persist.hip) spends its time, from in-kernel
time stamps (100 MHz) of the tail's workgroup and of the scan workgroups.

    python tools/persist_timeline.py [giga|fw] [rows] [d]
"""
import ctypes
import os
import sys

os.environ["BCX_DEV"] = "1"
os.environ["BCX_PERSIST"] = "1"
os.environ["BCX_PERSIST_DBG"] = "1"
os.environ["BCX_PERSIST_CHUNK"] = "24"
os.environ["BCX_PERSIST_MAX_GB"] = "100"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bayesian-coresets_amd")]
import numpy as np
import torch
import bayesiancoresets_amd as bc
from bayesiancoresets_amd import _native

alg = sys.argv[1] if len(sys.argv) > 1 else "giga"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 256
X = torch.empty(n, d, dtype=torch.float64, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
for r in range(0, n, 1 << 20):
    X[r:r + (1 << 20)].normal_(generator=g)
cls = {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe}[alg]
s = cls(X.T, X.sum(dim=0).cpu().numpy())
s.build(24)
s.build(24)
lib = _native.load()
NW = 1024 + 2 * 2048
out = (ctypes.c_longlong * NW)()
lib.bcx_debug_persist.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
rc = lib.bcx_debug_persist(s._eng.h, out)
raw = np.array(out[:], dtype=np.int64)
t = raw[:1024].reshape(128, 8).astype(np.float64) / 100.0   # us
ok = [i for i in range(2, 22) if t[i, 0] > 0 and t[i + 1, 6] > 0 and t[i, 6] > 0]
print("rc", rc, "iterations with stamps:", len(ok), "(%s, %d x %d)" % (alg, n, d))
def col(f):
    v = np.array([f(i) for i in ok])
    return "mean %7.2f  median %7.2f  p90 %7.2f us" % (v.mean(), np.median(v), np.percentile(v, 90))
print("iteration (GO to GO)                       ", col(lambda i: t[i + 1, 4] - t[i, 4]))
print("scan wg 0: GO seen -> its partials done    ", col(lambda i: t[i + 1, 6] - t[i + 1, 5]))
print("tail: GO stored -> scan wg 0 sees GO       ", col(lambda i: t[i + 1, 5] - t[i, 4]))
print("tail: scan wg 0 done -> all stamps seen    ", col(lambda i: t[i, 1] - t[i, 6]))
print("tail: stamps seen -> resolve done          ", col(lambda i: t[i, 2] - t[i, 1]))
print("tail: resolve done -> re-weight done       ", col(lambda i: t[i, 3] - t[i, 2]))
print("tail: re-weight done -> GO stored (release)", col(lambda i: t[i, 4] - t[i, 3]))
print("tail: stamps seen -> GO stored             ", col(lambda i: t[i, 4] - t[i, 1]))
w = raw[1024:].reshape(2048, 2).astype(np.float64) / 100.0
have = w[:, 1] > 0
nb = int(have.sum())
if nb:
    b0 = w[have, 0].min()
    beg, end = w[have, 0] - b0, w[have, 1] - b0
    print("last iteration, %d scan workgroups: pass begins %.2f .. %.2f us after the first, ends min %.2f  p10 %.2f  median %.2f  p90 %.2f  max %.2f us"
          % (nb, beg.min(), beg.max(), end.min(), np.percentile(end, 10), np.median(end), np.percentile(end, 90), end.max()))
    print("  idle share of the pass (mean over workgroups of max end - own end) / max end: %.4f" % ((end.max() - end).mean() / end.max()))
    for x in range(8):
        e = end[x::8]   # scan workgroup b is block b + 1: XCD (b + 1) % 8
        print("  blocks = %d mod 8 (XCD %d): median end %.2f  max %.2f" % (x, (x + 1) % 8, np.median(e), e.max()))
