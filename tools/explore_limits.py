"""dev: how far do GPU selections match the golden sequences in the numeric-limit regime?"""
import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
import bayesiancoresets_amd as bc
g = np.load(os.path.join(ROOT, "tests/golden/snnls_golden.npz"))
X = np.random.RandomState(1).randn(10000, 100)
cls = {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe, "omp": bc.snnls.OrthoPursuit}
for dtype in ("float32", "float64"):
    for alg in ("giga", "fw", "omp"):
        k = "F4_%s_" % alg
        itrs = int(g[k + "itrs"])
        s = cls[alg](X.T, X.sum(axis=0), dtype=dtype)
        s.build(itrs)
        sel, err, st = s.last_trace
        gs = g[k + "sel"]
        ssel = sel[sel >= 0]
        n = min(len(ssel), len(gs))
        neq = np.flatnonzero(ssel[:n] != gs[:n])
        first = int(neq[0]) if len(neq) else n
        ge = g[k + "err"]
        print(dtype, alg, "itrs", itrs, "trace len", len(sel), "ref selects", len(gs), "first mismatch at", first,
              "ref err there", ge[min(first, len(ge) - 1)], "| size", s.size(), "ref", int(g[k + "size"]), "| err", s.error(),
              "ref", float(g[k + "final_err"]), "| limit", s.reached_numeric_limit, "ref", bool(g[k + "limit"]),
              "| bad status", int((st != 0).sum()))
# harness
Ms = g["F3_Ms"]
for alg in ("giga", "fw", "omp"):
    for trial in (1,):
        X = np.random.RandomState(trial).randn(10000, 100)
        class IDP(bc.Projector):
            def update(self, w, p): pass
            def project(self, pts, grad=False): return pts
        a = bc.HilbertCoreset(X, IDP(), snnls=cls[alg])
        cs, er = [], []
        for m in range(len(Ms)):
            a.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
            cs.append(a.size()); er.append(a.error())
        k = "F3_t%d_%s_" % (trial, alg)
        cs, er = np.array(cs, dtype=float), np.array(er)
        bad = np.flatnonzero(cs != g[k + "csize"])
        print(alg, "harness: first csize mismatch at M index", (int(bad[0]), int(Ms[bad[0]])) if len(bad) else None,
              "final csize", cs[-1], "ref", g[k + "csize"][-1], "final err", er[-1], "ref", g[k + "err"][-1],
              "limit", a.snnls.reached_numeric_limit, "ref", bool(g[k + "limit"]))
        rel = np.abs(er - g[k + "err"]) / np.maximum(g[k + "err"], 1e-300)
        print("   rel err diff by M:", " ".join("%d:%.1e" % (M, r) for M, r in zip(Ms[::4], rel[::4])))
