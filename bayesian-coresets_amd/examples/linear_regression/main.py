#!/usr/bin/env python3
"""Linear-regression coreset experiment on the device engine (BASELINE.json configs[4]), command-line compatible with
the reference's examples/linear_regression/main.py:264-308 for the `run` sub-command:

    python main.py --alg SVI --trial 1 --data_num 100000 --proj_dim 256 --opt_itrs 100 run

The reference regresses log10 house prices (prices2018.npy, not distributed with the repository) on radial basis
functions of the location; here the observations are synthetic (common/rbf_workload.py), everything after the data load
follows main.py:56-259: RBF design matrix, conjugate prior, exact full-data posterior, the projectors (`SVI`: black-box
projector refreshed at the weighted coreset posterior, `GIGA-OPT`: samples from the true posterior, `GIGA-REAL`: samples
from the posterior of a sqrt(N)-point subsample), incremental build over the size schedule, and per size the forward /
reverse KL to the true posterior and the relative errors of mean and covariance, stored with the arguments in
results/<arg-hash>.csv.  Projection runs on the GPU (bc.DeviceProjector "linreg"); --host-sampler keeps the per-step
weighted-posterior sampler in NumPy exactly as main.py:141-147 writes it, the default draws it on the device too.
`--alg US` is the uniform-sampling baseline.  The `*-EXACT` variants (main.py:158-191) use the exact tangent-space projector
of this model (common/model_linreg.py: `tangent_space_projector`) on the host -- rows of D + proj_dim^2 numbers, no samples --
with the greedy construction on the device as for any other projector.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(1, os.path.join(HERE, "..", "common"))
import results  # noqa: E402
import rbf_workload  # noqa: E402
import model_linreg  # noqa: E402


def gaussian_kl(mu0, Sig0, mu1, Sig1inv):
    """KL(N(mu0, Sig0) || N(mu1, Sig1)) (the evaluation metric of main.py:248-249)."""
    diff = mu1 - mu0
    return 0.5 * (np.trace(Sig1inv.dot(Sig0)) + diff.dot(Sig1inv).dot(diff)
                  - np.linalg.slogdet(Sig1inv)[1] - np.linalg.slogdet(Sig0)[1] - mu0.shape[0])


def schedule(a):
    if a.coreset_size_spacing == "log":
        Ms = np.unique(np.logspace(0.0, np.log10(a.coreset_size_max), a.coreset_num_sizes, dtype=np.int32))
    else:
        Ms = np.unique(np.linspace(1, a.coreset_size_max, a.coreset_num_sizes, dtype=np.int32))
    return Ms if Ms[0] == 0 else np.hstack((0, Ms))              # the first recorded size is the empty coreset


def run(a):
    if results.check_exists(a, a.results_folder):
        print("Results already exist for arguments " + str(a))
        print("Quitting.")
        return
    import bayesiancoresets_amd as bc
    np.random.seed(a.trial)
    bc.util.set_verbosity(a.verbosity)
    Ms = schedule(a)
    wl = rbf_workload.make_rbf_regression(a.data_num, a.n_bases_per_scale, seed=a.trial)
    Z, mu0, Sig0, sigsq = wl["Z"], wl["mu0"], wl["Sig0"], wl["sigsq"]
    Sig0inv = np.linalg.inv(Sig0)
    print("dataset size : %s, basis dimension: %d, trial %d" % (Z.shape, mu0.shape[0], a.trial))
    mup, Up = model_linreg.weighted_posterior(mu0, Sig0inv, sigsq, Z, np.ones(Z.shape[0]))
    Sigp = Up.dot(Up.T)
    SigpInv = np.linalg.inv(Sigp)
    S = a.proj_dim
    fixed = lambda mu, U: (lambda n, w, p: mu + np.random.randn(n, mu.shape[0]).dot(U.T))
    Zhat = Z[np.random.randint(0, Z.shape[0], int(np.sqrt(Z.shape[0])))]
    muh, Uh = model_linreg.weighted_posterior(mu0, Sig0inv, sigsq, Zhat, np.ones(Zhat.shape[0]))
    sampler_w = model_linreg.posterior_sampler(mu0, Sig0, sigsq, device=None if a.host_sampler else "cuda", seed=a.trial)
    dev = lambda sampler: bc.DeviceProjector("linreg", sampler, S, sigsq=sigsq)
    def exact(at=None):
        # main.py:110-111, 158-191: quadratic block in the span of the leading proj_dim eigenvectors of X^T X
        X = Z[:, :-1]
        bV = np.linalg.eigh(X.T.dot(X))[1][:, -min(a.proj_dim, X.shape[1]):]
        prj = model_linreg.tangent_space_projector(bc, bV, mu0, Sig0, sigsq)
        if at is not None:
            prj.update(np.ones(at.shape[0]), at)
        return prj
    build = {
        "SVI": lambda: bc.SparseVICoreset(Z, dev(sampler_w), opt_itrs=a.opt_itrs, step_sched=eval(a.step_sched)),
        "SVI-EXACT": lambda: bc.SparseVICoreset(Z, exact(), opt_itrs=a.opt_itrs, step_sched=eval(a.step_sched)),
        "GIGA-OPT": lambda: bc.HilbertCoreset(Z, dev(fixed(mup, Up))),
        "GIGA-OPT-EXACT": lambda: bc.HilbertCoreset(Z, exact(Z)),
        "GIGA-REAL": lambda: bc.HilbertCoreset(Z, dev(fixed(muh, Uh))),
        "GIGA-REAL-EXACT": lambda: bc.HilbertCoreset(Z, exact(Zhat)),
        "US": lambda: bc.UniformSamplingCoreset(Z),
    }
    alg = build[a.alg]()
    w, p = [], []
    cputs, walls = np.zeros(Ms.shape[0]), np.zeros(Ms.shape[0])
    for m in range(Ms.shape[0]):
        print("M = %d: coreset construction, %s %d" % (Ms[m], a.alg, a.trial))
        c0, t0 = time.process_time(), time.perf_counter()
        alg.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
        cputs[m] = time.process_time() - c0 + (cputs[m - 1] if m else 0.0)
        walls[m] = time.perf_counter() - t0 + (walls[m - 1] if m else 0.0)
        wts, pts, idcs = alg.get()
        w.append(wts)
        p.append(pts)
    rklw, fklw = np.zeros(Ms.shape[0]), np.zeros(Ms.shape[0])
    mu_errs, Sig_errs, csizes = np.zeros(Ms.shape[0]), np.zeros(Ms.shape[0]), np.zeros(Ms.shape[0])
    for m in range(Ms.shape[0]):
        csizes[m] = (w[m] > 0).sum()
        muw, Uw = model_linreg.weighted_posterior(mu0, Sig0inv, sigsq, p[m], w[m])
        Sigw = Uw.dot(Uw.T)
        rklw[m] = gaussian_kl(muw, Sigw, mup, SigpInv)
        fklw[m] = gaussian_kl(mup, Sigp, muw, np.linalg.inv(Sigw))
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
    ap.add_argument("--data_num", type=int, default=10000)
    ap.add_argument("--alg", type=str, default="SVI", choices=["SVI", "SVI-EXACT", "GIGA-OPT", "GIGA-OPT-EXACT", "GIGA-REAL", "GIGA-REAL-EXACT", "US"])
    ap.add_argument("--proj_dim", type=int, default=100)
    ap.add_argument("--coreset_size_max", type=int, default=300)
    ap.add_argument("--coreset_num_sizes", type=int, default=6)
    ap.add_argument("--coreset_size_spacing", type=str, choices=["log", "linear"], default="log")
    ap.add_argument("--n_bases_per_scale", type=int, default=50)
    ap.add_argument("--opt_itrs", type=int, default=100)
    ap.add_argument("--step_sched", type=str, default="lambda i : 1./(1+i)")
    ap.add_argument("--host-sampler", action="store_true")
    ap.add_argument("--trial", type=int, default=1)
    ap.add_argument("--results_folder", type=str, default="results/")
    ap.add_argument("--verbosity", type=str, default="error", choices=["error", "warning", "critical", "info", "debug"])
    return ap


if __name__ == "__main__":
    args = parser().parse_args()
    if not hasattr(args, "func"):
        parser().error("choose a sub-command: run")
    args.func(args)
