"""test infrastructure (by hand): find and dissect an optimize() disagreement of tests/stress_parity.py"""
import sys, os
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd")); sys.path.insert(0, ROOT)
import bayesiancoresets_amd as bc
from oracle.snnls_oracle import SnnlsOracle
cls = {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe, "omp": bc.snnls.OrthoPursuit}
want = (5000, 256, "fw", "float64", "plain")
for seed in range(501, 521):
    rs = np.random.RandomState(seed)
    for case in range(60):
        N = int(rs.choice([700, 5000, 30000, 120000])); d = int(rs.choice([3, 17, 64, 100, 256, 300, 512]))
        alg = str(rs.choice(["giga", "fw", "omp"])); dtype = str(rs.choice(["float32", "float16", "float64"]))
        kind = str(rs.choice(["plain", "dups", "scaled", "parallel", "lowrank"]))
        X = rs.randn(N, d)
        if kind == "dups":
            src = rs.randint(0, N, size=N // 10); dst = rs.randint(0, N, size=N // 10); X[dst] = X[src]
        elif kind == "scaled":
            X *= 10.0 ** rs.uniform(-8, 3, size=(N, 1))
        elif kind == "parallel":
            base = rs.randn(d); X[: N // 20] = base * rs.uniform(0.5, 2.0, size=(N // 20, 1)) + 1e-7 * rs.randn(N // 20, d)
        elif kind == "lowrank":
            r = max(1, d // 8); X = rs.randn(N, r).dot(rs.randn(r, d)) + 1e-3 * rs.randn(N, d)
        if (N, d, alg, dtype, kind) != want or case != 8:
            continue
        itrs = int(min(30, d + 5))
        o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="onepass"); o.build(itrs)
        for single in (False, True):
            if single: os.environ["BCX_OPT_SINGLE"] = "1"
            else: os.environ.pop("BCX_OPT_SINGLE", None)
            s = cls[alg](X.T, X.sum(axis=0), dtype=dtype); s.build(itrs)
            w0 = s.weights().copy()
            s.optimize()
            gw = s.weights()
            o2 = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="onepass"); o2.build(itrs); o2.optimize()
            ow = o2.weights()
            sup = ow > 0
            # exact LS on the support in numpy for reference
            A = X[sup].T
            ls = np.linalg.lstsq(A, X.sum(axis=0), rcond=None)[0]
            print("seed", seed, "single" if single else "grid", "err gpu %.12f oracle %.12f" % (s.error(), o2.error()),
                  "max rel dw gpu-vs-oracle %.3e" % np.max(np.abs(gw[sup] - ow[sup]) / ow[sup]),
                  "gpu-vs-lstsq %.3e oracle-vs-lstsq %.3e" % (np.max(np.abs(gw[sup] - ls) / np.abs(ls)), np.max(np.abs(ow[sup] - ls) / np.abs(ls))),
                  "support equal", np.array_equal(gw > 0, sup), "pre-opt w equal", np.allclose(w0[sup], o.weights()[sup], rtol=1e-9))
