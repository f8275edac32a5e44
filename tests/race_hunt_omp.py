"""test infrastructure (by hand): the OMP step kernel (csrc/omp_lh.hip: 16 workgroups, exchange vectors, owner-computes
double-double inverse, columns entering and leaving) run many times on the same ill-conditioned vectors -- any number of
distinct outcomes other than 1 is a data race (a stale line in some XCD's L2, a missed barrier) or an uninitialised read.
    python tests/race_hunt_omp.py [reps] [rows]"""
import argparse, hashlib, os, sys
os.environ.setdefault("BCX_DEV", "1")   # BCX_OMP_THREADS is a dev switch (csrc/dev_util.h)
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
sys.path.insert(0, ROOT)
import torch
import bayesiancoresets_amd as bc
import model_lr
import bench

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
args = argparse.Namespace(rows=rows, features=10, seed=1, dim=512)
Z = bench.logistic_rows(args, torch, 0, rows)
mu, cov = model_lr.laplace_fit(Z)
samples = np.random.RandomState(2).multivariate_normal(mu, cov, 512)
vecs = bc.DeviceProjector("logistic", lambda n, w, p: samples[:n], 512).project(Z)
seen = {}
left = None
for r in range(reps):
    s = bc.snnls.OrthoPursuit(vecs.t(), None)
    s.build(90)
    s.build(70)                      # (a second call: begin_kernel, barrier base reset)
    tr = s.last_trace
    h = hashlib.md5(tr[0].tobytes() + tr[1].tobytes() + s.weights().tobytes() + np.float64(s.error()).tobytes()).hexdigest()
    seen.setdefault(h, []).append(r)
    left = s._eng.omp_stats()
print("OMP N=%d d=512, 160 iterations x %d runs (BCX_OMP_THREADS=%s): %d distinct outcome(s); last run's stats %s%s"
      % (rows, reps, os.environ.get("BCX_OMP_THREADS", "auto"), len(seen), left,
         "" if len(seen) == 1 else "  <-- " + str([v[:5] for v in seen.values()])), flush=True)
