#!/usr/bin/env python3
"""Gaussian-mean coreset experiment on the device engine, command-line compatible with the reference's
examples/gaussian/main.py:228-266 for the `run` sub-command:

    python main.py --alg GIGA-OPT --data_num 1000 --data_dim 200 --proj_dim 100 --coreset_size_max 200 --trial 1 run

Everything follows main.py:40-222: synthetic observations x ~ N(1, I), conjugate prior, exact full-data posterior, the
black-box projectors (`SVI`: refreshed at the weighted coreset posterior; `GIGA-OPT`: samples from the true posterior;
`GIGA-REAL`: from the posterior of a sqrt(N)-point subsample), their exact tangent-space counterparts (`*-EXACT`:
common/model_gaussian.py `tangent_space_projector`), `US`, the incremental build over the size schedule and the closed-form
evaluation -- reverse / forward KL to the true posterior, relative errors of mean and covariance -- stored with the
arguments in results/<arg-hash>.csv.  This likelihood is not one of the device projector's families: the projection is the
reference's NumPy callback (`bc.BlackBoxProjector`), the greedy construction runs on the device engine."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(1, os.path.join(HERE, "..", "common"))
import results  # noqa: E402
import model_gaussian as gaussian  # noqa: E402


def run(a):
    if results.check_exists(a, a.results_folder):
        print("Results already exist for arguments " + str(a))
        print("Quitting.")
        return
    import bayesiancoresets_amd as bc
    np.random.seed(a.trial)
    bc.util.set_verbosity(a.verbosity)
    if a.coreset_size_spacing == "log":
        Ms = np.unique(np.logspace(0.0, np.log10(a.coreset_size_max), a.coreset_num_sizes, dtype=np.int32))
    else:
        Ms = np.unique(np.linspace(1, a.coreset_size_max, a.coreset_num_sizes, dtype=np.int32))
    if Ms[0] != 0:
        Ms = np.hstack((0, Ms))                                    # the first recorded size is the empty coreset
    D = a.data_dim
    mu0, Sig0inv, Sig = np.zeros(D), np.eye(D), np.eye(D)          # main.py:59-62: change these to change prior / likelihood
    Siginv = np.linalg.inv(Sig)
    logdetSig = np.linalg.slogdet(Sig)[1]
    x = np.random.multivariate_normal(np.ones(D), Sig, a.data_num)
    mup, Up = gaussian.weighted_posterior(mu0, Sig0inv, Siginv, x, np.ones(x.shape[0]))
    Sigp = Up.dot(Up.T)
    SigpInv = np.linalg.inv(Sigp)
    loglik = lambda pts, th: gaussian.log_likelihood(pts, th, Siginv, logdetSig)
    fixed = lambda mu, U: (lambda n, w, p: mu + np.random.randn(n, mu.shape[0]).dot(U.T))
    xhat = x[np.random.randint(0, x.shape[0], int(np.sqrt(x.shape[0])))]
    muh, Uh = gaussian.weighted_posterior(mu0, Sig0inv, Siginv, xhat, np.ones(xhat.shape[0]))

    def sampler_w(n, wts, pts):                                     # main.py:107-113
        if wts is None or pts is None or np.asarray(pts).shape[0] == 0:
            wts, pts = np.zeros(1), np.zeros((1, D))
        mu, U = gaussian.weighted_posterior(mu0, Sig0inv, Siginv, pts, wts)
        return mu + np.random.randn(n, D).dot(U.T)

    bb = lambda sampler: bc.BlackBoxProjector(sampler, a.proj_dim, loglik)

    def exact(at=None):
        prj = gaussian.tangent_space_projector(bc, mu0, Sig0inv, Siginv)
        if at is not None:
            prj.update(np.ones(at.shape[0]), at)
        return prj

    sched = eval(a.step_sched)
    build = {
        "SVI": lambda: bc.SparseVICoreset(x, bb(sampler_w), opt_itrs=a.opt_itrs, step_sched=sched),
        "SVI-EXACT": lambda: bc.SparseVICoreset(x, exact(), opt_itrs=a.opt_itrs, step_sched=sched),
        "GIGA-OPT": lambda: bc.HilbertCoreset(x, bb(fixed(mup, Up))),
        "GIGA-OPT-EXACT": lambda: bc.HilbertCoreset(x, exact(x)),
        "GIGA-REAL": lambda: bc.HilbertCoreset(x, bb(fixed(muh, Uh))),
        "GIGA-REAL-EXACT": lambda: bc.HilbertCoreset(x, exact(xhat)),
        "US": lambda: bc.UniformSamplingCoreset(x),
    }
    alg = build[a.alg]()
    n = Ms.shape[0]
    cputs, walls, csizes = np.zeros(n), np.zeros(n), np.zeros(n)
    rklw, fklw, mu_errs, Sig_errs = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    for m in range(n):
        print("M = %d: coreset construction, %s %d" % (Ms[m], a.alg, a.trial))
        c0, t0 = time.process_time(), time.perf_counter()
        alg.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
        cputs[m] = time.process_time() - c0 + (cputs[m - 1] if m else 0.0)
        walls[m] = time.perf_counter() - t0 + (walls[m - 1] if m else 0.0)
        wts, pts, idcs = alg.get()
        csizes[m] = (wts > 0).sum()
        if len(wts):
            muw, Uw = gaussian.weighted_posterior(mu0, Sig0inv, Siginv, pts, wts)
        else:
            muw, Uw = gaussian.weighted_posterior(mu0, Sig0inv, Siginv, np.zeros((1, D)), np.zeros(1))
        Sigw = Uw.dot(Uw.T)
        rklw[m] = gaussian.gaussian_kl(muw, Sigw, mup, SigpInv)
        fklw[m] = gaussian.gaussian_kl(mup, Sigp, muw, np.linalg.inv(Sigw))
        mu_errs[m] = np.sqrt(((mup - muw) ** 2).sum()) / np.sqrt((mup ** 2).sum())
        Sig_errs[m] = np.sqrt(((Sigp - Sigw) ** 2).sum()) / np.sqrt((Sigp ** 2).sum())
    print("final: csize %d, reverse KL %.6g, forward KL %.6g, %.2f s wall" % (csizes[-1], rklw[-1], fklw[-1], walls[-1]))
    results.save(a, a.results_folder, csizes=csizes, Ms=Ms, cputs=cputs, walls=walls, rklw=rklw, fklw=fklw, mu_errs=mu_errs,
                 Sig_errs=Sig_errs)


def parser():
    ap = argparse.ArgumentParser("Runs Riemannian linear regression (employing coreset contruction) on the specified dataset")
    sub = ap.add_subparsers(help="sub-command help")
    rp = sub.add_parser("run", help="Runs the main computational code")
    rp.set_defaults(func=run)
    ap.add_argument("--data_num", type=int, default=1000)
    ap.add_argument("--data_dim", type=int, default=200)
    ap.add_argument("--alg", type=str, default="SVI", choices=["SVI", "SVI-EXACT", "GIGA-OPT", "GIGA-OPT-EXACT", "GIGA-REAL", "GIGA-REAL-EXACT", "US"])
    ap.add_argument("--proj_dim", type=int, default=100)
    ap.add_argument("--coreset_size_max", type=int, default=200)
    ap.add_argument("--coreset_num_sizes", type=int, default=7)
    ap.add_argument("--coreset_size_spacing", type=str, choices=["log", "linear"], default="log")
    ap.add_argument("--opt_itrs", type=int, default=100)
    ap.add_argument("--step_sched", type=str, default="lambda i : 1./(1+i)")
    ap.add_argument("--trial", type=int, default=1)
    ap.add_argument("--results_folder", type=str, default="results/")
    ap.add_argument("--verbosity", type=str, default="error", choices=["error", "warning", "critical", "info", "debug"])
    return ap


if __name__ == "__main__":
    args = parser().parse_args()
    if not hasattr(args, "func"):
        parser().error("choose a sub-command: run")
    args.func(args)
