"""dev: where the fused OMP step's time goes on the REAL configs[2] vectors (Laplace-projected logistic log-likelihoods),
by path and active-set size.  Needs a library built with EXTRA=-DBCX_TIMING (tools/build_timing.sh -> lib_timing/):
    python tools/omp_hist.py [--rows 1000000] [--itrs 140]
Prints one line per step and a histogram by (mode before, mode after the step phase)."""
import argparse, os, sys, ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
sys.path.insert(0, ROOT)
from bayesiancoresets_amd import _native as nat
nat.LIB_PATH = os.path.join(ROOT, "bayesian-coresets_amd", "lib_timing", "libbcx.so")
import torch
import bayesiancoresets_amd as bc
import model_lr
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1000000)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--itrs", type=int, default=140)
ap.add_argument("--randn", action="store_true", help="Gaussian rows instead of the projected vectors")
ap.add_argument("--quiet", action="store_true")
a = ap.parse_args()
if a.randn:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    vecs = torch.randn(a.rows, a.dim, device="cuda", dtype=torch.float64, generator=g)
else:
    args = argparse.Namespace(rows=a.rows, features=10, seed=1, dim=a.dim)
    Z = bench.logistic_rows(args, torch, 0, a.rows)
    mu, cov = model_lr.laplace_fit(Z)
    samples = np.random.RandomState(2).multivariate_normal(mu, cov, a.dim)
    prj = bc.DeviceProjector("logistic", lambda n, w, p: samples[:n], a.dim)
    vecs = prj.project(Z)
s = bc.snnls.OrthoPursuit(vecs.t(), None)
s.build(a.itrs)
sel, err, status = s.last_trace
n = len(sel)
lib = nat.load()
lib.bcx_debug_omp_log.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros((n, 24), dtype=np.int64)
assert lib.bcx_debug_omp_log(buf.ctypes.data, n) == 0
MODE = {0: "idle", 1: "done", 2: "fast_try", 3: "fast_accept", 4: "general"}
NAMES = ["entry", "rows", "B1", "decide", "u=Hg", "B2", "step", "gen+comb", "rank-1", "B5", "finish", "next"]
tot = {}
for r in buf:
    it, k, p, m0, m1, st_, np1, ill = r[:8]
    t = r[8:20] / 100.0
    key = (MODE.get(int(m0)), MODE.get(int(m1)))
    tot.setdefault(key, []).append((t[11] if t[11] > 0 else t[10], int(k), int(p), t[7] - t[6], r[20], r[21], r[22], r[23]))
    if not a.quiet:
        print("it %3d k %3d p %3d %-8s -> %-11s st %d np %3d ill %d | total %6.1f us: " % (it, k, p, MODE.get(int(m0)), MODE.get(int(m1)), st_, np1, ill, t[11] if t[11] > 0 else t[10])
              + " ".join("%s %.1f" % (NAMES[i], t[i] - t[i - 1]) for i in range(1, 12))
              + " | outer %d inner %d refine %d del %d" % tuple(r[20:24]))
print("size %d error %.6g limit %s; statuses %s" % (s.size(), s.error(), s.reached_numeric_limit, {int(k): int((status == k).sum()) for k in set(status)}))
for key, v in sorted(tot.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    v = np.array(v, dtype=float)
    print("%-10s -> %-12s n %3d  in-kernel mean %6.1f us max %6.1f  (gen+comb mean %6.1f)  k %3d..%3d  outer %.2f inner %.2f refine %.2f del %.2f"
          % (key[0], key[1], len(v), v[:, 0].mean(), v[:, 0].max(), v[:, 3].mean(), v[:, 1].min(), v[:, 1].max(),
             v[:, 4].mean(), v[:, 5].mean(), v[:, 6].mean(), v[:, 7].mean()))
print("all steps: mean %.1f us in-kernel" % np.mean([x[0] for v in tot.values() for x in v]))
