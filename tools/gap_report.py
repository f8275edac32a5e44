"""dev: where an iteration's time goes on the GPU's own clock -- kernel durations and the idle gaps between consecutive
kernels of the stream, from a rocprofv3 --kernel-trace rocpd database.
    rocprofv3 --kernel-trace -d out -o t -- python bench.py --config c2 --steps 400 --warmup 20 --no-cpu-baseline
    python tools/gap_report.py out/**/t_results.db"""
import sqlite3, sys
import numpy as np
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0] for r in rows]; st = np.array([r[1] for r in rows], dtype=np.int64); en = np.array([r[2] for r in rows], dtype=np.int64)
def short(n):
    for k in ("scan_kernel", "scan_long_kernel", "tail_kernel", "tail_exchange_kernel", "resolve_kernel", "omp_lh_kernel", "apply_kernel", "resolve_exchange_kernel"):
        if k in n: return k
    return None
sn = [short(n) for n in names]
# steady state: the longest run of consecutive engine kernels
idx = [i for i, s in enumerate(sn) if s]
best, cur = [], []
for a, b in zip(idx, idx[1:] + [None]):
    cur.append(a)
    if b is None or b != a + 1:
        if len(cur) > len(best): best = cur
        cur = []
idx = best[len(best) // 10:]          # skip the first tenth
dur = {}; gap = {}
for a, b in zip(idx, idx[1:]):
    dur.setdefault(sn[a], []).append((en[a] - st[a]) / 1e3)
    gap.setdefault(sn[a] + " -> " + sn[b], []).append((st[b] - en[a]) / 1e3)
print("steady-state kernels: %d" % len(idx))
for k, v in dur.items(): print("  %-24s n %5d  mean %8.2f us  median %8.2f" % (k, len(v), np.mean(v), np.median(v)))
for k, v in gap.items(): print("  gap %-40s n %5d  mean %6.2f us  median %6.2f  p90 %6.2f" % (k, len(v), np.mean(v), np.median(v), np.percentile(v, 90)))
tot = (en[idx[-1]] - st[idx[0]]) / 1e3
nscan = sum(1 for i in idx if sn[i] in ("scan_kernel", "scan_long_kernel"))
print("  per iteration: %.2f us over %d iterations" % (tot / max(nscan, 1), nscan))
