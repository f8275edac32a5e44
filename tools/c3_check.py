"""dev: config-3 pipeline (device Laplace fit + device projection, as bench.py --config c3) at a size the CPU oracle
can follow: engine OMP vs oracle OMP on the same projected vectors -- selections, statuses, errors."""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
sys.path.insert(0, ROOT)
import torch
import bayesiancoresets_amd as bc
import model_lr
import bench
from oracle.snnls_oracle import SnnlsOracle

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=200000)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--itrs", type=int, default=200)
ap.add_argument("--alg", default="omp")
a = ap.parse_args()
args = argparse.Namespace(rows=a.rows, features=10, seed=1, dim=a.dim)
Z = bench.logistic_rows(args, torch, 0, a.rows)
mu, cov = model_lr.laplace_fit(Z)
samples = np.random.RandomState(2).multivariate_normal(mu, cov, a.dim)
prj = bc.DeviceProjector("logistic", lambda n, w, p: samples[:n], a.dim)
vecs = prj.project(Z)
cls = {"omp": bc.snnls.OrthoPursuit, "giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe}[a.alg]
s = cls(vecs.t(), None)
t0 = time.perf_counter()
s.build(a.itrs)
print("engine: %.3f s, size %d, error %.6g, limit %s" % (time.perf_counter() - t0, s.size(), s.error(), s.reached_numeric_limit))
sel, err, status = s.last_trace
print("engine statuses: ok %d, fail %s; iterations run %d" % ((status == 0).sum(), {int(k): int((status == k).sum()) for k in set(status) if k}, len(sel)))
V = vecs.cpu().numpy()
nr = np.sqrt((V ** 2).sum(axis=1))
print("row norms: min %.3e median %.3e max %.3e" % (nr.min(), np.median(nr), nr.max()))
o = SnnlsOracle(V.T, V.sum(axis=0), alg=a.alg, mode="faithful")
t0 = time.perf_counter()
o.build(a.itrs)
print("oracle: %.1f s, size %d, error %.6g, limit %s" % (time.perf_counter() - t0, o.size(), o.error(), o.reached_numeric_limit))
osel = np.array([t[0] for t in o.trace]); ost = np.array([t[2] for t in o.trace]); oerr = np.array([t[1] for t in o.trace])
n = min(len(osel), len(sel))
same = (osel[:n] == sel[:n]) & (ost[:n] == status[:n])
first = int(np.argmin(same)) if not same.all() else n
print("oracle statuses: ok %d fail %d; first difference at iteration %d of %d" % ((ost == 0).sum(), (ost != 0).sum(), first, n))
if first < n:
    lo = max(0, first - 2)
    for i in range(lo, min(n, first + 4)):
        print(i, "engine", sel[i], status[i], err[i], "| oracle", osel[i], ost[i], oerr[i])
