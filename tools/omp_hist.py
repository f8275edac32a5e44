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
NAMES = ["entry", "rows", "B1", "decide", "u,dz=H[g,grad]", "B2", "LH step", "combine", "B-last", "finish", "next"]
tot = {}
for r in buf:
    it, k, p, m0, m1, st_, np1, nbar = r[:8]
    t = r[8:20] / 100.0
    ent, rem, res = int(r[20]), int(r[21]), int(r[22])
    key = ("done" if m0 == 1 else "step", "resolve" if res else ("left %d" % min(rem, 3) if rem else "closed form"))
    tot.setdefault(key, []).append((t[10], int(k), int(p), t[6] - t[5], ent, rem, res, int(nbar)))
    if not a.quiet:
        print("it %3d k %3d np %3d %-5s st %d barriers %2d entered %d left %d resolve %d | total %6.1f us: " % (it, k, p, key[0], st_, nbar, ent, rem, res, t[10])
              + " ".join("%s %.1f" % (NAMES[i], t[i] - t[i - 1]) for i in range(1, 11)) + " | threads %d" % r[19])
print("size %d error %.6g limit %s; statuses %s; omp stats %s" % (s.size(), s.error(), s.reached_numeric_limit,
      {int(k): int((status == k).sum()) for k in set(status)}, s._eng.omp_stats()))
for key, v in sorted(tot.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    v = np.array(v, dtype=float)
    print("%-5s %-12s n %3d  in-kernel mean %6.1f us max %6.1f  (LH step mean %6.1f)  k %3d..%3d  barriers %.2f"
          % (key[0], key[1], len(v), v[:, 0].mean(), v[:, 0].max(), v[:, 3].mean(), v[:, 1].min(), v[:, 1].max(), v[:, 7].mean()))
print("all steps: mean %.1f us in-kernel" % np.mean([x[0] for v in tot.values() for x in v]))
