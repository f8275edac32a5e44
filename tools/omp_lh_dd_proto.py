"""dev (CPU only): the incremental OMP / NNLS step of csrc/omp_lh.hip with the inverse H held in DOUBLE-DOUBLE (hi, lo),
everything else in doubles -- the NumPy model that settled the kernel's numerics: closed-form
enter / leave updates alone, no refinement, against the CPU oracle's selections and error on configs[2]-style vectors.
The data (vectors + oracle trace) is the cache file tools/omp_lh_proto.py writes:
    python tools/omp_lh_proto.py --rows 100000 --itrs 250 --policy deferred      # writes /tmp/omp_proto_100000_512_1.npz
    python tools/omp_lh_dd_proto.py --cache /tmp/omp_proto_100000_512_1.npz [--apply-deferred]
Double-double arithmetic is emulated with Dekker / Veltkamp error-free transformations (NumPy has no FMA)."""
import sys, os, time, argparse
import numpy as np
ap = argparse.ArgumentParser()
ap.add_argument("--cache", required=True)
ap.add_argument("--itrs", type=int, default=250)
ap.add_argument("--apply-deferred", action="store_true")
a = ap.parse_args()
zf = np.load(a.cache); V, osel, oerr = zf["V"], zf["osel"], zf["oerr"]
N, d = V.shape; b = V.sum(axis=0); norms = np.sqrt((V**2).sum(axis=1)); An = V / norms[:, None]; eps = 2.220446049250313e-16

SPL = 134217729.0
def split(a):
    t = SPL * a; hi = t - (t - a); return hi, a - hi
def two_prod(a, b):
    p = a * b; ah, al = split(a); bh, bl = split(b)
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl
def two_sum(a, b):
    s = a + b; bb = s - a; return s, (a - (s - bb)) + (b - bb)
def dd_add(ah, al, bh, bl):
    s, e = two_sum(ah, bh); e = e + (al + bl); return two_sum(s, e)   # fast renorm
def dd_mul_d(ah, al, b):
    p, e = two_prod(ah, b); e = e + al * b; return two_sum(p, e)
def dd_mul(ah, al, bh, bl):
    p, e = two_prod(ah, bh); e = e + (ah * bl + al * bh); return two_sum(p, e)
def dd_recip(ah, al):
    x = 1.0 / ah
    # r = 1 - a*x in dd
    ph, pl = dd_mul_d(ah, al, x)
    rh, rl = dd_add(1.0, 0.0, -ph, -pl)
    ch, cl = dd_mul_d(rh, rl, x)
    return dd_add(x, 0.0, ch, cl)
def dd_sum_rows(Ph, Pl):   # sum over axis 1 of dd matrix -> dd vector
    sh = np.zeros(Ph.shape[0]); sl = np.zeros(Ph.shape[0])
    for j in range(Ph.shape[1]):
        sh, sl = dd_add(sh, sl, Ph[:, j], Pl[:, j])
    return sh, sl
def dd_mv(Hh, Hl, v):      # dd matrix x double vector
    Ph, Pe = two_prod(Hh, v[None, :]); Pe = Pe + Hl * v[None, :]
    Ph, Pe = two_sum(Ph, Pe)
    return dd_sum_rows(Ph, Pe)
def dd_dot_d(uh, ul, g):   # sum u_i g_i -> dd scalar
    ph, pe = two_prod(uh, g); pe = pe + ul * g
    sh, sl = 0.0, 0.0
    for i in range(len(g)):
        sh, sl = dd_add(sh, sl, ph[i], pe[i])
    return sh, sl

rows = np.zeros((0, d)); idx = []; x = np.zeros(0); G = np.zeros((0, 0)); c = np.zeros(0); P = []
Hh = np.zeros((0, 0)); Hl = np.zeros((0, 0))
relmax = 0.0; ratios = []; left = 0
t0 = time.time()
for it in range(a.itrs):
    xw = x.dot(rows) if len(x) else np.zeros(d)
    r = b - xw
    dots = An.dot(r); fpos = int(dots.argmax()); f = fpos
    if P:
        act = np.array([idx[s] for s in P]); order = np.argsort(act); neg = -dots[act]; jn = order[np.argmax(neg[order])]
        if not (dots[fpos] >= neg[jn]): f = int(act[jn])
    k = len(idx)
    if f in idx: slot = idx.index(f)
    else:
        slot = k; row = V[f]; g_all = rows.dot(row) if k else np.zeros(0)
        rows = np.vstack([rows, row[None]])
        G2 = np.zeros((k+1, k+1)); G2[:k, :k] = G; G2[k, :k] = g_all; G2[:k, k] = g_all; G2[k, k] = row.dot(row); G = G2
        c = np.append(c, row.dot(b)); idx.append(f); x = np.append(x, 0.0)
    if slot not in P:
        p = len(P); tolscale = 10.0 * eps * max(d, len(idx)) * np.sqrt(b.dot(b))
        xP = x[P].copy(); S = set(P) | {slot}; rej = set(); z = xP.copy()
        if p:
            gamma = rows[P].dot(r); dzh, dzl = dd_mv(Hh, Hl, gamma); ratios.append(np.abs(dzh).max() / np.abs(xP).max())
            if a.apply_deferred: z = xP + dzh
        xs = xP.copy(); cand = slot; first = True
        while cand is not None:
            g = G[cand, P] if p else np.zeros(0)
            if p:
                uh, ul = dd_mv(Hh, Hl, g); guh, gul = dd_dot_d(uh, ul, g)
            else:
                uh = ul = np.zeros(0); guh = gul = 0.0
            sch, scl = dd_add(G[cand, cand], 0.0, -guh, -gul)
            wv = c[cand] - g.dot(z); entered = False
            if not (wv > tolscale * np.sqrt(G[cand, cand])):
                if first: z = xs.copy()
                done = True
            elif not (sch > 1e-12 * G[cand, cand]):
                rej.add(cand); done = False
                if first: z = xs.copy()
            else:
                t = wv / sch; ih, il = dd_recip(sch, scl)
                z = np.append(z - t * uh, t); xs = np.append(xs, 0.0)
                wh, wl = dd_mul(uh, ul, ih, il)                       # u * inv
                oh, ol = dd_mul(wh[:, None], wl[:, None], uh[None, :], ul[None, :])
                Nh, Nl = dd_add(Hh, Hl, oh, ol)
                H2h = np.zeros((p+1, p+1)); H2l = np.zeros((p+1, p+1))
                H2h[:p, :p] = Nh; H2l[:p, :p] = Nl; H2h[p, :p] = -wh; H2l[p, :p] = -wl; H2h[:p, p] = -wh; H2l[:p, p] = -wl; H2h[p, p] = ih; H2l[p, p] = il
                Hh, Hl = H2h, H2l; P.append(cand); p += 1; entered = True; done = False
            first = False; inner = 0
            while p:
                bad = np.flatnonzero(~(z > 0))
                if not len(bad): break
                al = xs[bad] / (xs[bad] - z[bad]); al[np.isnan(al)] = 0.0
                j = bad[np.argmin(al)]; alpha = al.min(); xn = xs + alpha * (z - xs); rm = ~(xn > 0); rm[j] = True
                xs = np.where(rm, 0.0, xn)
                for q in sorted(np.flatnonzero(rm), reverse=True):
                    gone = P[q]
                    if gone == cand and inner == 0 and entered: rej.add(gone)
                    hh, hl = Hh[:, q].copy(), Hl[:, q].copy()
                    qh, ql = dd_recip(hh[q], hl[q])
                    z = z - (z[q] / hh[q]) * hh
                    fh, fl = dd_mul(hh, hl, qh, ql)
                    oh, ol = dd_mul(fh[:, None], fl[:, None], hh[None, :], hl[None, :])
                    Nh, Nl = dd_add(Hh, Hl, -oh, -ol)
                    last = p - 1
                    if q != last:
                        order = list(range(p)); order[q] = last; order = order[:last]
                    else:
                        order = [i for i in range(p) if i != q]
                    Hh = Nh[np.ix_(order, order)]; Hl = Nl[np.ix_(order, order)]; z = z[order]; xs = xs[order]; P[:] = [P[i] for i in order]; p -= 1; left += 1
                inner += 1
            x[:] = 0.0; x[P] = z; xs = z.copy()
            if done: break
            out = [j for j in S if j not in P and j not in rej]; cand = None
            if out:
                duals = [c[j] - G[j, P].dot(x[P]) for j in out]; jb = int(np.argmax(duals))
                if duals[jb] > tolscale * np.sqrt(G[out[jb], out[jb]]): cand = out[jb]
    err = np.sqrt(((x.dot(rows) - b) ** 2).sum())
    rel = abs(err - oerr[it]) / oerr[it]; relmax = max(relmax, rel)
    if f != osel[it]:
        print("first selection difference at iteration %d: proto %d oracle %d" % (it, f, osel[it])); break
    if it % 40 == 0: print("it %3d p %3d err %.12g rel %.2e" % (it, len(P), err, rel))
rt = np.array(ratios)
print("deferred correction / weights by 20-step window:", " ".join("%.0e" % rt[i:i+20].max() for i in range(0, len(rt), 20)))
print("dd-H apply_deferred=%s: %d its %.1f s, max rel err diff %.2e, left %d" % (a.apply_deferred, it + 1, time.time() - t0, relmax, left))
