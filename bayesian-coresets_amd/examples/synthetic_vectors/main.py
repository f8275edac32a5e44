#!/usr/bin/env python3
"""Synthetic-vector experiment on the device engine, command-line compatible with the reference's
examples/synthetic_vectors/main.py:31-150 so its run_experiment.sh loops run unchanged:

    python main.py --alg GIGA --trial 1 --data_type normal run
    python main.py --alg GIGA plot Ms err --summarize trial --groupby Ms --plot_legend alg

`run`: seeded randn(N, d) or eye(N) vectors, identity projector, HilbertCoreset with FW / GIGA / OMP /
uniform sampling built incrementally over the log/linear size schedule; per size M it records csize, err
and cput and stores them with the arguments in results/<arg-hash>.csv (common/results.py).
`cput` stays `time.process_time()` for column compatibility (main.py:92); `wall` is added next to it
because CPU seconds say nothing about a build that runs on the GPU.
`plot`: prints / writes the matching series as text (common/summary.py); the Bokeh canvas is out of scope.
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
import summary  # noqa: E402


def schedule(a):
    if a.coreset_size_spacing == "log":
        return np.unique(np.logspace(0.0, np.log10(a.coreset_size_max), a.coreset_num_sizes, dtype=np.int32))
    return np.unique(np.linspace(1, a.coreset_size_max, a.coreset_num_sizes, dtype=np.int32))


def run(a):
    if results.check_exists(a, a.results_folder):
        print("Results already exist for arguments " + str(a))
        print("Quitting.")
        return
    import bayesiancoresets_amd as bc

    class IDProjector(bc.Projector):
        def update(self, wts, pts):
            pass

        def project(self, pts, grad=False):
            return pts

    np.random.seed(a.trial)
    bc.util.set_verbosity(a.verbosity)
    algs = {"FW": bc.snnls.FrankWolfe, "GIGA": bc.snnls.GIGA, "OMP": bc.snnls.OrthoPursuit,
            "US": bc.snnls.UniformSampling}
    Ms = schedule(a)
    X = np.random.randn(a.data_num, a.data_dim) if a.data_type == "normal" else np.eye(a.data_num)
    print("data: %s, trial %s, alg: %s" % (a.data_type, a.trial, a.alg))

    t0 = time.perf_counter()
    alg = bc.HilbertCoreset(X, IDProjector(), snnls=algs[a.alg])
    t_init = time.perf_counter() - t0
    err, csize = np.zeros(Ms.shape[0]), np.zeros(Ms.shape[0])
    cput, wall = np.zeros(Ms.shape[0]), np.zeros(Ms.shape[0])
    for m in range(Ms.shape[0]):
        c0, w0 = time.process_time(), time.perf_counter()
        alg.build(int(Ms[m] if m == 0 else Ms[m] - Ms[m - 1]))
        cput[m] = time.process_time() - c0 + (cput[m - 1] if m else 0.0)
        wall[m] = time.perf_counter() - w0 + (wall[m - 1] if m else 0.0)
        wts, pts, idcs = alg.get()
        csize[m] = (wts > 0).sum()
        err[m] = alg.error()
    print("constructor %.3f s, %d iterations in %.3f s wall; final csize %d err %.6g"
          % (t_init, int(Ms[-1]), wall[-1], csize[-1], err[-1]))
    results.save(a, a.results_folder, err=err, csize=csize, Ms=Ms, cput=cput, wall=wall)


def plot(a):
    match = dict(vars(a))
    for name in (a.summarize or []):
        match.pop(name, None)
    match.pop(a.plot_legend, None)
    for k in [k for k in match if k.startswith("plot_") or k in ("func", "summarize", "groupby", "out")]:
        match.pop(k)
    table = results.load_matching(match, a.results_folder)
    if table is None:
        print("No matching results to plot, skipping")
        return
    summary.summarize(a, table, a.out)


def parser():
    ap = argparse.ArgumentParser("Runs sparse nonnegative regression")
    sub = ap.add_subparsers(help="sub-command help")
    rp = sub.add_parser("run", help="Runs the main computational code")
    rp.set_defaults(func=run)
    pp = sub.add_parser("plot", help="Summarises stored results")
    pp.set_defaults(func=plot)
    ap.add_argument("--alg", type=str, default="GIGA", choices=["FW", "GIGA", "OMP", "US"])
    ap.add_argument("--data_num", type=int, default=10000)
    ap.add_argument("--data_dim", type=int, default=100)
    ap.add_argument("--data_type", type=str, default="normal", choices=["normal", "axis"])
    ap.add_argument("--coreset_size_max", type=int, default=1000)
    ap.add_argument("--coreset_num_sizes", type=int, default=50)
    ap.add_argument("--coreset_size_spacing", type=str, choices=["log", "linear"], default="log")
    ap.add_argument("--trial", type=int)
    ap.add_argument("--results_folder", type=str, default="results/")
    ap.add_argument("--verbosity", type=str, default="error",
                    choices=["error", "warning", "critical", "info", "debug"])
    pp.add_argument("plot_x", type=str)
    pp.add_argument("plot_y", type=str)
    pp.add_argument("--plot_legend", type=str)
    pp.add_argument("--summarize", type=str, nargs="*")
    pp.add_argument("--groupby", type=str)
    pp.add_argument("--out", type=str, help="write the summary to this file instead of stdout")
    return ap


if __name__ == "__main__":
    args = parser().parse_args()
    if not hasattr(args, "func"):
        parser().error("choose a sub-command: run | plot")
    args.func(args)
