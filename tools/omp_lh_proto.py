"""dev (CPU only): NumPy model of the incremental OMP / NNLS step of csrc/omp_lh.hip, to settle its numerics before
spending GPU time -- the carried least-squares solution, closed-form enter / leave updates of the inverse H, and the
refinement policy -- against the CPU oracle (scipy.optimize.nnls per step) on configs[2]-style vectors (Laplace-projected
logistic log-likelihoods: numerical rank ~100, nearly dependent columns).

    python tools/omp_lh_proto.py --rows 100000 --itrs 200 --policy deferred|gram2|gramconv|none
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import model_lr
from lr_workload import make_data, log_likelihood
from oracle.snnls_oracle import SnnlsOracle

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=100000)
ap.add_argument("--dim", type=int, default=512)
ap.add_argument("--itrs", type=int, default=200)
ap.add_argument("--policy", default="deferred")
ap.add_argument("--gram-its", type=int, default=2)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--thr", type=float, default=1e-10)
ap.add_argument("--enter-only", action="store_true")
ap.add_argument("--stag", type=float, default=0.25)
ap.add_argument("--conv", type=float, default=0.0, help="stop refining once the correction is below this fraction of the weights")
a = ap.parse_args()

cache = "/tmp/omp_proto_%d_%d_%d.npz" % (a.rows, a.dim, a.seed)
if os.path.exists(cache):
    z = np.load(cache)
    V, osel, oerr = z["V"], z["osel"], z["oerr"]
else:
    Z = make_data(a.seed, a.rows, 10)
    mu, cov = model_lr.laplace_fit(Z)
    samples = np.random.RandomState(a.seed + 1).multivariate_normal(mu, cov, a.dim)
    V = log_likelihood(Z, samples)
    V -= V.mean(axis=1)[:, None]
    t0 = time.time()
    o = SnnlsOracle(V.T, V.sum(axis=0), alg="omp", mode="onepass")
    o.build(max(a.itrs, 250))
    print("oracle: %.1f s" % (time.time() - t0))
    osel = np.array([t[0] for t in o.trace]); oerr = np.array([t[1] for t in o.trace])
    np.savez(cache, V=V, osel=osel, oerr=oerr)

N, d = V.shape
b = V.sum(axis=0)
norms = np.sqrt((V ** 2).sum(axis=1))
An = V / norms[:, None]
eps = 2.220446049250313e-16


class LH(object):
    """state of the device step: slots (selection order), passive list P (positions), H = inv(G[P,P]), x by slot"""

    def __init__(self):
        self.rows = np.zeros((0, d)); self.idx = []; self.x = np.zeros(0)
        self.G = np.zeros((0, 0)); self.c = np.zeros(0)
        self.P = []                 # slots by position
        self.H = np.zeros((0, 0))
        self.stats = dict(left=0, resolve=0, gram_its=0)
        self.ratios = []; self.ratio = 0.0; self.ill = False

    def xw(self):
        return self.x.dot(self.rows) if len(self.x) else np.zeros(d)

    def step(self, f, policy):
        r = b - self.xw()
        k = len(self.idx)
        if f in self.idx:
            slot = self.idx.index(f)
        else:
            slot = k
            row = V[f]
            g_all = self.rows.dot(row) if k else np.zeros(0)
            self.rows = np.vstack([self.rows, row[None]])
            G2 = np.zeros((k + 1, k + 1)); G2[:k, :k] = self.G; G2[k, :k] = g_all; G2[:k, k] = g_all; G2[k, k] = row.dot(row)
            self.G = G2; self.c = np.append(self.c, row.dot(b)); self.idx.append(f); self.x = np.append(self.x, 0.0)
        if slot in self.P:
            return
        P = self.P; p = len(P)
        tolscale = 10.0 * eps * max(d, len(self.idx)) * np.sqrt(b.dot(b))
        x = self.x
        xP = x[P].copy()
        S = set(P) | {slot}
        rej = set()
        # deferred refinement from the gradient of the carried solution
        z = xP.copy()
        self.ratio = 0.0
        if policy not in ("none", "gramonly") and p:
            gamma = self.rows[P].dot(r)
            dz = self.H.dot(gamma)
            z = xP + dz
            self.ratio = np.abs(dz).max() / np.abs(xP).max()
            self.ratios.append(self.ratio)
        xs = xP.copy()
        cand = slot
        first = True
        while cand is not None:
            g = self.G[cand, P] if p else np.zeros(0)
            u = self.H.dot(g) if p else np.zeros(0)
            sc = self.G[cand, cand] - g.dot(u)
            wv = self.c[cand] - g.dot(z)
            entered = False
            if not (wv > tolscale * np.sqrt(self.G[cand, cand])):
                if first:
                    z = xs.copy()
                cand_done = True
            elif not (sc > 1e-12 * self.G[cand, cand]):
                rej.add(cand)
                if first:
                    z = xs.copy()
                cand_done = False
            else:
                t = wv / sc; inv = 1.0 / sc
                z = np.append(z - t * u, t); xs = np.append(xs, 0.0)
                H2 = np.zeros((p + 1, p + 1)); H2[:p, :p] = self.H + np.outer(u, u) * inv; H2[p, :p] = -u * inv; H2[:p, p] = -u * inv; H2[p, p] = inv
                self.H = H2; P.append(cand); p += 1; entered = True
                cand_done = False
            first = False
            # refinement of the carried solution on the current passive set (Gram space), policy dependent
            def gram_refine(z, its):
                prev = np.inf
                for _ in range(its):
                    res = self.c[P] - self.G[np.ix_(P, P)].dot(z)
                    dzz = self.H.dot(res)
                    z = z + dzz
                    self.stats["gram_its"] += 1
                    m = np.abs(dzz).max()
                    if a.conv > 0 and (m <= a.conv * np.abs(z).max() or m > a.stag * prev):
                        break
                    prev = m
                return z
            if policy == "adaptive":
                if self.ratio > a.thr:
                    self.ill = True
                do_gram = self.ill
            else:
                do_gram = policy.startswith("gram")
            if entered and do_gram:
                z = gram_refine(z, a.gram_its)
            # inner loop: columns leave until z > 0
            inner = 0
            while p:
                bad = np.flatnonzero(~(z > 0))
                if not len(bad):
                    break
                al = xs[bad] / (xs[bad] - z[bad]); al[np.isnan(al)] = 0.0
                j = bad[np.argmin(al)]; alpha = al.min()
                xn = xs + alpha * (z - xs)
                rm = ~(xn > 0); rm[j] = True
                xs = np.where(rm, 0.0, xn)
                for q in sorted(np.flatnonzero(rm), reverse=True):
                    gone = P[q]
                    if gone == cand and inner == 0 and entered:
                        rej.add(gone)
                    h = self.H[:, q].copy(); hqq = h[q]
                    z = z - (z[q] / hqq) * h
                    Hn = self.H - np.outer(h, h) / hqq
                    keep = [i for i in range(p) if i != q]
                    # (device: move last into q; same set)
                    last = p - 1
                    if q != last:
                        order = list(range(p)); order[q] = last; order = order[:last]
                    else:
                        order = keep
                    self.H = Hn[np.ix_(order, order)]; z = z[order]; xs = xs[order]
                    P[:] = [P[i] for i in order]; p -= 1
                    self.stats["left"] += 1
                if do_gram and p and not a.enter_only:
                    z = gram_refine(z, a.gram_its)
                inner += 1
            x[:] = 0.0
            x[P] = z
            xs = z.copy()
            if cand_done:
                break
            # next candidate
            out = [j for j in S if j not in P and j not in rej]
            cand = None
            if out:
                duals = [self.c[j] - self.G[j, P].dot(x[P]) for j in out]
                jb = int(np.argmax(duals))
                if duals[jb] > tolscale * np.sqrt(self.G[out[jb], out[jb]]):
                    cand = out[jb]


lh = LH()
w_active = lambda: [lh.idx[s] for s in lh.P]
first_diff = None
t0 = time.time()
sel = []
for it in range(a.itrs):
    xw = lh.xw()
    r = b - xw
    dots = An.dot(r)
    fpos = int(dots.argmax())
    f = fpos
    if lh.P:
        act = np.array([lh.idx[s] for s in lh.P])
        # orthopursuit.py:26-35: compare the best positive direction with the best negative one among the active points
        order = np.argsort(act)                      # (lowest global index wins ties as in the dense argmax)
        neg = -dots[act]
        jn = order[np.argmax(neg[order])]
        if not (dots[fpos] >= neg[jn]):
            f = int(act[jn])
    lh.step(f, a.policy)
    err = np.sqrt(((lh.xw() - b) ** 2).sum())
    sel.append(f)
    relmax = max(globals().get("relmax", 0.0), abs(err - oerr[it]) / oerr[it])
    if f != osel[it] and first_diff is None:
        first_diff = it
        print("first selection difference at iteration %d: proto %d oracle %d" % (it, f, osel[it]))
        break
    if it % 20 == 0 or it > a.itrs - 5:
        print("it %3d f %6d p %3d err %.12g oracle %.12g rel %.2e" % (it, f, len(lh.P), err, oerr[it], abs(err - oerr[it]) / oerr[it]))
rt = np.array(lh.ratios)
print("deferred correction / weights by 20-step window:", " ".join("%.0e" % rt[i:i + 20].max() for i in range(0, len(rt), 20)))
print("max relative error difference over all steps: %.2e" % relmax)
print("policy %s: %d iterations in %.1f s, first difference %s, stats %s" % (a.policy, len(sel), time.time() - t0, first_diff, lh.stats))
