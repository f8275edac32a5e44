"""test infrastructure (run by hand on the GPU box: python tests/stress_parity.py SEED CASES): randomized parity stress -- HIP engine vs CPU oracle over random shapes / algorithms / storage types,
including duplicated rows (exact ties), wildly scaled rows and nearly parallel rows."""
import sys, os, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")  # tests/ -> repo root
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd")); sys.path.insert(0, ROOT)
import bayesiancoresets_amd as bc
from oracle.snnls_oracle import SnnlsOracle

cls = {"giga": bc.snnls.GIGA, "fw": bc.snnls.FrankWolfe, "omp": bc.snnls.OrthoPursuit}
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad = 0
t0 = time.time()
for case in range(ncase):
    N = int(rs.choice([700, 5000, 30000, 120000]))
    d = int(rs.choice([3, 17, 64, 100, 256, 300, 512]))
    alg = str(rs.choice(["giga", "fw", "omp"]))
    dtype = str(rs.choice(["float32", "float16", "float64"]))
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
    itrs = int(min(30, d + 5))
    if os.environ.get('STRESS_VERBOSE'): print('case', case, N, d, alg, dtype, kind, flush=True)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="onepass"); o.build(itrs)
    s = cls[alg](X.T, X.sum(axis=0), dtype=dtype); s.build(itrs)
    sel = s.last_trace[0]; osel = np.array([t[0] for t in o.trace]); oerr = np.array([t[1] for t in o.trace])
    scale = np.sqrt((X.sum(axis=0) ** 2).sum())
    n = 0
    for t in o.trace:
        if t[2] != 0 or t[1] < 1e-7 * scale: break
        n += 1
    n = min(n, len(sel))
    # bit-identical duplicate rows tie exactly; which of them NumPy's arg-max names depends on the BLAS kernel of
    # the host CPU (observed: same input, different pick on two hosts), the engine takes the lowest index.
    # Compare up to that relabelling: every row is represented by the first row with the same bytes.
    if kind == "dups":
        _, first = np.unique(X, axis=0, return_index=True)
        canon = {}
        for i in np.sort(first): canon.setdefault(X[i].tobytes(), i)
        cmap = lambda v: np.array([canon[X[i].tobytes()] if i >= 0 else i for i in v])
        sel, osel = cmap(sel), cmap(osel)
    ok = np.array_equal(sel[:n], osel[:n])
    if ok and n:
        ok = np.allclose(s.last_trace[1][:n], oerr[:n], rtol=1e-6, atol=1e-9 * scale)
    if not ok:
        bad += 1
        k = int(np.argmax(sel[:n] != osel[:n])) if not np.array_equal(sel[:n], osel[:n]) else -1
        print("MISMATCH case %d: N=%d d=%d %s %s %s first diff at %d (n=%d) gpu %s oracle %s" % (case, N, d, alg, dtype, kind, k, n, sel[max(0,k-1):k+2], osel[max(0,k-1):k+2]), flush=True)
    # optimize() on the state both sides agree on (plain data, selections matched to the end, error well above
    # the numeric limit): same support afterwards, weights to 1e-5, cost to 1e-7
    if ok and kind in ("plain", "scaled") and n == len(osel) == len(sel) and oerr[-1] > 1e-6 * scale:
        o_ok = o.optimize(); s.optimize()
        ge, oe = s.error(), o.error()
        tol_e = 1e-7 * oe + 1e-9 * scale
        if ge < oe - tol_e and not s.reached_numeric_limit:
            continue   # the engine's optimum is strictly better than SciPy's answer (rows scaled over 11 decades: SciPy's
                       # dual tolerance 10 max(m,n) eps ||A||_1 leaves tiny columns out / stops early; with k >= d it may
                       # even come back worse than its start and the reference latches) -- not claimed as a mismatch
        ow, gw = o.weights(), s.weights()
        same = ge <= oe + tol_e and bool(s.reached_numeric_limit) == (not o_ok)
        if same and (ow > 0).sum() < d:       # unique minimiser only for independent columns
            same = np.array_equal(np.flatnonzero(ow > 0), np.flatnonzero(gw > 0)) and \
                   np.allclose(gw[ow > 0], ow[ow > 0], rtol=1e-5, atol=1e-10 * ow.max())
        if not same:
            bad += 1
            sup = ow > 0
            rel = np.max(np.abs(gw[sup] - ow[sup]) / ow[sup]) if sup.any() and np.array_equal(gw > 0, sup) else float("nan")
            print("OPTIMIZE MISMATCH case %d: N=%d d=%d %s %s %s err gpu %.15g oracle %.15g support %d / %d (equal %s) accepted %s / %s max rel dw %.3e"
                  % (case, N, d, alg, dtype, kind, ge, oe, (gw > 0).sum(), (ow > 0).sum(), np.array_equal(gw > 0, sup),
                     not s.reached_numeric_limit, o_ok, rel), flush=True)
print("cases %d bad %d in %.1f s" % (ncase, bad, time.time() - t0))
