#!/usr/bin/env python3
"""Logistic / Poisson regression coreset experiment on the device engine (BASELINE.json configs[2] names this harness),
command-line compatible with the reference's examples/logistic_poisson_regression/main.py:232-289 for the `run` sub-command:

    python main.py --model lr --dataset synth_lr --alg GIGA-OPT --proj_dim 500 --coreset_size_max 1000 run
    python main.py --model poiss --dataset synth_poiss --alg SVI --opt_itrs 100 run

Construction follows main.py:66-185: the model (`lr`: rows y x, `poiss`: rows [x, y] with a softplus rate), the three
projectors -- `GIGA-OPT` samples from the Laplace approximation of the full-data posterior, `GIGA-REAL` from that of a
sqrt(N)-point subsample, `SVI` from the Laplace approximation of the weighted coreset, refreshed at every step -- the
incremental build over the size schedule, and `US` as the uniform baseline.  The projections run on the GPU
(bc.DeviceProjector "logistic" / "poisson"), the greedy construction on the device engine.
`--dataset synth_lr | synth_poiss` generates the data (`--data_num` rows, `--data_dim` columns); a path to an .npz with
arrays X, y (the reference's data/*.npz layout, last column of X the intercept) is standardised as load_data does.
The reference evaluates a coreset by Stan MCMC on it (pystan: not available here, and not on the path this repository
is about); this harness reports the same metric columns -- reverse / forward KL to the full-data posterior, relative errors
of mean and covariance -- between the LAPLACE approximations of the coreset posterior and of the full-data posterior."""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(1, os.path.join(HERE, "..", "common"))
import results  # noqa: E402
import model_lr  # noqa: E402
import model_poiss  # noqa: E402


def gaussian_kl(mu0, Sig0, mu1, Sig1inv):
    """KL(N(mu0, Sig0) || N(mu1, Sig1)) (model_gaussian.py KL: the metric of main.py:226-227)."""
    diff = mu1 - mu0
    return 0.5 * (np.trace(Sig1inv.dot(Sig0)) + diff.dot(Sig1inv).dot(diff)
                  - np.linalg.slogdet(Sig1inv)[1] - np.linalg.slogdet(Sig0)[1] - mu0.shape[0])


def load(a, rs):
    """Z (rows as the model wants them) for the data set named on the command line."""
    if os.path.exists(a.dataset):
        d = np.load(a.dataset)
        X, y = model_poiss.standardized(d["X"]), np.asarray(d["y"], dtype=np.float64)
        return y[:, None] * X if a.model == "lr" else np.hstack((X, y[:, None]))
    if a.model == "lr":
        Zx = model_lr.synthetic_rows(a.data_num, a.data_dim - 1, rs)          # y x
        y = np.sign(Zx[:, :1] / np.where(Zx[:, :1] == 0.0, 1.0, Zx[:, :1]))   # (recover y to append the intercept column y * 1)
        return np.hstack((Zx, y))
    return model_poiss.synthetic_rows(a.data_num, a.data_dim, rs)


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
    Z = load(a, np.random)
    D = Z.shape[1] if a.model == "lr" else Z.shape[1] - 1
    family = "logistic" if a.model == "lr" else "poisson"
    print("dataset %s: %d rows, %d parameters, model %s, trial %d" % (a.dataset, Z.shape[0], D, a.model, a.trial))

    def laplace(pts, wts):
        if a.model == "lr":
            return model_lr.laplace_fit(pts, wts)
        return model_poiss.laplace_fit(pts, wts)

    mup, Sigp = laplace(Z, None)                                               # main.py:145 (tangent space of GIGA-OPT)
    SigpInv = np.linalg.inv(Sigp)
    Zhat = Z[np.random.randint(0, Z.shape[0], int(np.sqrt(Z.shape[0])))]      # main.py:150-152
    muh, Sigh = laplace(Zhat, None)
    gauss = lambda mu, Sig: (lambda n, w, p: np.atleast_2d(np.random.multivariate_normal(mu, Sig, n)))

    def sampler_w(n, wts, pts):                                                # main.py:155-162
        if wts is None or pts is None or np.asarray(pts).shape[0] == 0:
            return np.random.randn(n, D)                                       # the prior N(0, I)
        keep = np.asarray(wts) > 0
        if not keep.any():
            return np.random.randn(n, D)
        mu, Sig = laplace(np.atleast_2d(pts)[keep], np.asarray(wts)[keep])
        return np.atleast_2d(np.random.multivariate_normal(mu, Sig, n))

    # the same sampler on the device (csrc/laplace.hip: one launch per call, and -- through enqueue_plan -- from weights that
    # never leave the device, so SparseVI enqueues its whole ADAM loop) where the model fits it: D <= 32 parameters
    sampler_host = sampler_w
    if D <= 32 and not getattr(a, "host_sampler", False):
        try:
            sampler_w = bc.LaplacePosteriorSampler(family, D, seed=a.trial)
        except RuntimeError:
            sampler_w = sampler_host
    dev = lambda sampler: bc.DeviceProjector(family, sampler, a.proj_dim)
    build = {
        "SVI": lambda: bc.SparseVICoreset(Z, dev(sampler_w), opt_itrs=a.opt_itrs, step_sched=eval(a.step_sched)),
        "GIGA-OPT": lambda: bc.HilbertCoreset(Z, dev(gauss(mup, Sigp))),
        "GIGA-REAL": lambda: bc.HilbertCoreset(Z, dev(gauss(muh, Sigh))),
        "US": lambda: bc.UniformSamplingCoreset(Z),
    }
    alg = build[a.alg]()
    n = Ms.shape[0]
    cputs, walls, csizes = np.zeros(n), np.zeros(n), np.zeros(n)
    rklw, fklw, mu_errs, Sig_errs = np.zeros(n), np.zeros(n), np.zeros(n), np.zeros(n)
    for m in range(n):
        print("M = %d: coreset construction, %s %s %d" % (Ms[m], a.alg, a.dataset, a.trial))
        c0, t0 = time.process_time(), time.perf_counter()
        alg.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
        cputs[m] = time.process_time() - c0 + (cputs[m - 1] if m else 0.0)
        walls[m] = time.perf_counter() - t0 + (walls[m - 1] if m else 0.0)
        wts, pts, idcs = alg.get()
        csizes[m] = (wts > 0).sum()
        if csizes[m] > 0:
            muw, Sigw = laplace(pts[wts > 0], wts[wts > 0])
        else:
            muw, Sigw = np.zeros(D), np.eye(D)
        rklw[m] = gaussian_kl(muw, Sigw, mup, SigpInv)
        fklw[m] = gaussian_kl(mup, Sigp, muw, np.linalg.inv(Sigw))
        mu_errs[m] = np.sqrt(((mup - muw) ** 2).sum()) / np.sqrt((mup ** 2).sum())
        Sig_errs[m] = np.sqrt(((Sigp - Sigw) ** 2).sum()) / np.sqrt((Sigp ** 2).sum())
    print("final: csize %d, reverse KL %.6g, forward KL %.6g, %.2f s wall" % (csizes[-1], rklw[-1], fklw[-1], walls[-1]))
    results.save(a, a.results_folder, csizes=csizes, Ms=Ms, cputs=cputs, walls=walls, rklw=rklw, fklw=fklw, mu_errs=mu_errs,
                 Sig_errs=Sig_errs)


def parser():
    ap = argparse.ArgumentParser("Runs logistic or poisson regression (employing coreset contruction) on the specified dataset")
    sub = ap.add_subparsers(help="sub-command help")
    rp = sub.add_parser("run", help="Runs the main computational code")
    rp.set_defaults(func=run)
    ap.add_argument("--model", type=str, choices=["lr", "poiss"], default="lr")
    ap.add_argument("--dataset", type=str, default="synth_lr")
    ap.add_argument("--data_num", type=int, default=10000)
    ap.add_argument("--data_dim", type=int, default=3)
    ap.add_argument("--alg", type=str, default="SVI", choices=["SVI", "GIGA-OPT", "GIGA-REAL", "US"])
    ap.add_argument("--mcmc_samples_full", type=int, default=10000, help="accepted for command-line compatibility; unused (no MCMC evaluation)")
    ap.add_argument("--mcmc_samples_coreset", type=int, default=10000, help="accepted for command-line compatibility; unused")
    ap.add_argument("--proj_dim", type=int, default=500)
    ap.add_argument("--coreset_size_max", type=int, default=1000)
    ap.add_argument("--coreset_num_sizes", type=int, default=7)
    ap.add_argument("--coreset_size_spacing", type=str, choices=["log", "linear"], default="log")
    ap.add_argument("--opt_itrs", type=int, default=100)
    ap.add_argument("--step_sched", type=str, default="lambda i : 1./(1+i)")
    ap.add_argument("--trial", type=int, default=1)
    ap.add_argument("--results_folder", type=str, default="results/")
    ap.add_argument("--verbosity", type=str, default="error", choices=["error", "warning", "critical", "info", "debug"])
    ap.add_argument("--host_sampler", action="store_true", help="SVI: the Laplace fit of every sampler call on the host (NumPy) instead of csrc/laplace.hip")
    return ap


if __name__ == "__main__":
    args = parser().parse_args()
    if not hasattr(args, "func"):
        parser().error("choose a sub-command: run")
    args.func(args)
