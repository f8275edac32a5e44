#!/usr/bin/env python3
"""Headline benchmark: greedy coreset iterations/sec on the workloads BASELINE.json names.

    python bench.py [--gpus N --steps K --warmup W] [--config c4|c2|c3|c5] [--alg fw|giga|omp --rows N --dim d]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one greedy iteration: one pass of the hot path over the resident rows.

  c4 (default)  BASELINE.json configs[3] and the metric's own config: synthetic randn N = 10,000,000, d = 512,
                Frank-Wolfe, row-sharded over --gpus ranks (total N fixed => "scaling": "strong").  On one GPU the line
                also carries two side legs of the same workload, neither of them the headline: rows stored in fp64
                (`exact_mode_its`: the reference's arithmetic end to end) and rows stored in fp16 (`f16_rows_its`, with
                `f16_rows_same_selections`: whether every selection equals the fp32 run's).
  c2            configs[1]: synthetic randn N = 1,000,000, d = 256, GIGA.
  c3            configs[2]: Laplace-projected logistic-regression vectors (examples/simple_lr pipeline: data, Laplace fit
                at the MAP, S = 512 posterior samples, log-likelihood projection ON THE DEVICE), N = 1,000,000, OMP.
  c5            configs[4]: SparseVICoreset on the RBF-basis linear regression (D = 301 = 6 scales x 50 bases + 1,
                S = 256, opt_itrs = 100), N = 5,000,000 rows sharded over --gpus ranks; a step is one greedy SparseVI
                step = 1 + opt_itrs full-data projections; the dominant kernel is the fp64-MFMA projection.
  --alg/--rows/--dim without --config: ad-hoc synthetic randn workload (sweeps, tests).

Inputs are generated on the device (seeded per 8192-row block, so the matrix is the same for every shard count) and are
resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# multi-process GPU work on this platform needs dmabuf IPC (RCCL and the peer mailbox share device memory across
# processes: hipIpcGetMemHandle fails in legacy mode on this host driver) -- set it whatever the launcher exported
os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd", "examples", "common"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
F64_MFMA_PEAK_TF = 78.6   # MI355X fp64 matrix peak
GEN_BLOCK = 8192          # rows per seeded generation block (multiple of the 1024-row engine chunk)

CONFIGS = {
    "c2": dict(kind="synthetic", alg="giga", rows=1_000_000, dim=256, what="BASELINE.json configs[1]"),
    "c3": dict(kind="logistic", alg="omp", rows=1_000_000, dim=512, what="BASELINE.json configs[2]"),
    "c4": dict(kind="synthetic", alg="fw", rows=10_000_000, dim=512, what="BASELINE.json configs[3] (the metric's config)"),
    "c5": dict(kind="sparsevi", alg=None, rows=5_000_000, dim=256, what="BASELINE.json configs[4]"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS))
    ap.add_argument("--alg", default=None, choices=["fw", "giga", "omp"])
    ap.add_argument("--rows", type=int, default=None)
    ap.add_argument("--dim", type=int, default=None, help="row length d of the vectors (c3 / c5: the number of samples S)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64", "float16"],
                    help="storage of the normalised rows (state and re-score are always fp64)")
    ap.add_argument("--no-exact-rows", action="store_true", help="do not keep the raw fp64 rows resident")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exact-mode", action="store_true", help="skip the fp64-stored-rows figure of the c4 line")
    ap.add_argument("--cpu-rows", type=int, default=None,
                    help="rows of the CPU-baseline sample.  Default: a 1,000,000-row sample first (halved until three copies "
                         "of the fp64 sample fit the free memory), then -- single-GPU runs, host memory and --cpu-wall-cap "
                         "permitting -- the oracle on ALL rows (SURVEY 8d); given explicitly: that sample only")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-wall-cap", type=float, default=300.0,
                    help="wall-clock cap of the full-N CPU baseline (generation + constructor + iterations); the number of "
                         "timed iterations is cut to fit, and below 2 the leg is skipped (the scaled sample stands)")
    ap.add_argument("--cpu-full", default="auto", choices=["auto", "always", "never"],
                    help="full-N CPU baseline: auto = single-GPU runs only (a multi-rank line carries the scaled sample)")
    ap.add_argument("--no-one-rank-leg", action="store_true",
                    help="multi-rank synthetic runs: skip the one-shard run of the same workload on rank 0's GPU "
                         "(one_rank_its / scaling_efficiency on the line)")
    ap.add_argument("--no-f16-leg", action="store_true", help="c4: skip the extra leg with fp16-stored rows")
    ap.add_argument("--opt-itrs", type=int, default=100, help="c5: ADAM steps per greedy step (sparsevi.py:7)")
    ap.add_argument("--no-side-legs", action="store_true", help="default line only: skip the compact c2 / c3 / c5 legs")
    ap.add_argument("--k-curve", default=None,
                    help="c5: comma-separated coreset sizes; every greedy step is timed on its own and the line gains "
                         "steps/s and ADAM-step microseconds at these sizes (run enough --steps to reach them)")
    ap.add_argument("--colsum", default="mfma", choices=["mfma", "moments", "auto"],
                    help="c5: column sums of the full-data projection by the fp64-MFMA projection kernel (101 per greedy step), "
                         "or in closed form from the one-time (D+1)x(D+1) moments of the data (linear-regression family)")
    ap.add_argument("--features", type=int, default=10, help="c3: regression features D (simple_lr/main.py:24)")
    a = ap.parse_args()
    explicit = a.config is not None
    if a.config is None:
        a.config = "c4"
    c = dict(CONFIGS[a.config])
    adhoc = not explicit and (a.alg or a.rows or a.dim)
    a.kind = c["kind"]
    a.alg = a.alg or c["alg"]
    a.rows = a.rows or c["rows"]
    a.dim = a.dim or c["dim"]
    a.what = "ad-hoc synthetic workload" if adhoc else c["what"]
    a.adhoc = bool(adhoc)
    # (c3: OMP on these vectors reaches its numeric floor after ~100 points -- the log-likelihood functions of a
    # 10-parameter model span a space of low numerical rank -- and latches near iteration 160: stay below that)
    if a.steps is None:
        a.steps = {"sparsevi": 3, "logistic": 100}.get(a.kind, 300)
    if a.warmup is None:
        a.warmup = {"sparsevi": 1, "logistic": 20}.get(a.kind, 30)
    return a


def gen_block(torch, seed, block_id, rows, d, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1_000_003 + block_id)
    return torch.randn(rows, d, device=device, dtype=torch.float64, generator=g)


def host_threads():
    """threads the NumPy restatement actually computes on: the BLAS pool behind ndarray.dot (OpenBLAS caps at 64)"""
    try:
        from threadpoolctl import threadpool_info
        pools = threadpool_info()
        blas = [p.get("num_threads", 1) for p in pools if p.get("user_api") == "blas"]
        return max(blas or [p.get("num_threads", 1) for p in pools] or [1])
    except Exception:
        return os.cpu_count() or 1


def blas_info():
    """'<library> <version> (<threads> threads, <threading layer>)' of the BLAS behind ndarray.dot, and the host's core count."""
    try:
        from threadpoolctl import threadpool_info
        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"] or threadpool_info()
        p0 = max(pools, key=lambda p: p.get("num_threads", 0))
        return "%s %s (%d threads, %s)" % (p0.get("internal_api", "?"), p0.get("version", "?"), p0.get("num_threads", 0),
                                           p0.get("threading_layer", "?")), os.cpu_count() or 0
    except Exception:
        return "unknown", os.cpu_count() or 0


def blas_threads_for_baseline():
    """A launcher may have pinned the BLAS pool to one thread per rank (torch.distributed.run exports OMP_NUM_THREADS=1): the
    CPU baseline is timed on the host's cores, as many as the BLAS build takes (NumPy's OpenBLAS: 64), whatever the launcher
    set for the GPU ranks.  Stays in force for the rest of the process (only rank 0 computes on the host afterwards)."""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=max(1, min(os.cpu_count() or 1, 64)), user_api="blas")
    except Exception:
        pass


def host_rows(torch, args, n_s):
    """The first n_s rows of the synthetic matrix as a C-contiguous fp64 host array: generated on the device per seeded block
    exactly as load_synthetic does (same block sizes, so the same numbers) and copied block by block into one allocation."""
    X = np.empty((n_s, args.dim))
    for b0 in range(0, n_s, GEN_BLOCK):
        m = min(GEN_BLOCK, args.rows - b0)
        blk = gen_block(torch, args.seed, b0 // GEN_BLOCK, m, args.dim, "cuda")
        k = min(m, n_s - b0)
        X[b0:b0 + k] = blk[:k].cpu().numpy()
    return X


def _time_oracle(args, X, mode, seconds, warm, stride, cap_its):
    """(iterations, elapsed s, constructor s) of the oracle in `mode` on X: `warm` untimed iterations, then `stride` at a time
    until `seconds` have passed or `cap_its` are done."""
    from oracle.snnls_oracle import SnnlsOracle
    t0 = time.perf_counter()
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=args.alg, mode=mode)
    ctor = time.perf_counter() - t0
    if warm:
        o.build(warm)
    done, t1 = 0, time.perf_counter()
    while True:
        o.build(stride)
        done += stride
        el = time.perf_counter() - t1
        if el > seconds or done >= cap_its:
            break
    del o
    return done, el, ctor


def cpu_baseline_snnls(args, torch, world, what, sample=None):
    """The oracle's faithful mode (reference op sequence: 5 passes of OpenBLAS dgemv/dgemm per iteration, fp64) timed on
    this box's host cores.  First on a bounded sample of the same workload (scaled linearly in N); then, when the host can
    hold it and the wall-clock cap allows, on ALL rows -- that figure replaces the scaled one and the scaled
    one stays beside it (`sample_value_scaled`: how linear the cost is in N).  `sample`: ready-made host rows (config 3)."""
    t_leg = time.perf_counter()
    blas_threads_for_baseline()
    explicit = args.cpu_rows is not None
    if sample is not None:
        X, gen_s = sample, 0.0
    else:
        n_s = min(args.cpu_rows if explicit else 1_000_000, args.rows)
        try:      # the oracle holds A, An and temporaries: three copies of the fp64 sample
            import psutil
            while n_s > 8 * GEN_BLOCK and 3.5 * n_s * args.dim * 8 > 0.6 * psutil.virtual_memory().available:
                n_s //= 2
        except ImportError:
            n_s = min(n_s, 200_000)
        if n_s < args.rows:
            n_s = (n_s // GEN_BLOCK) * GEN_BLOCK or n_s
        t0 = time.perf_counter()
        X = host_rows(torch, args, n_s)
        gen_s = time.perf_counter() - t0
    n_s = X.shape[0]
    done, el, ctor_s = _time_oracle(args, X, "faithful", args.cpu_seconds, 3, 2, 200)
    its_sample = done / el
    scale = n_s / float(args.rows)   # cost per iteration is linear in N
    # the fairer CPU number: the same arithmetic with A.dot(w) maintained incrementally -- one pass over the
    # matrix per iteration instead of the reference's five
    done1, el1, _ = _time_oracle(args, X, "onepass", args.cpu_seconds / 2, 3, 4, 200)
    blas, ncpu = blas_info()
    out = {
        "value": its_sample * scale, "unit": "iterations/s", "cores": int(host_threads()), "kind": "port",
        "sample": "oracle faithful mode (NumPy/OpenBLAS fp64, reference op sequence), %s, %s, first %d of %d rows, d=%d, "
                  "%d iterations in %.1f s = %.2f it/s on the sample, scaled linearly in N (x%.4f)"
                  % (args.alg, what, n_s, args.rows, X.shape[1], done, el, its_sample, scale),
        "onepass_value": done1 / el1 * scale, "blas": blas, "host_cpus": ncpu, "sample_rows": int(n_s),
        "sample_value_scaled": its_sample * scale, "full_n": None,
    }
    want_full = (args.cpu_full == "always" or (args.cpu_full == "auto" and world == 1)) and not explicit \
        and sample is None and n_s < args.rows
    if not want_full:
        if n_s < args.rows:
            out["full_n"] = {"ran": False, "why": "--cpu-rows given" if explicit else
                             ("multi-rank line: the scaled sample (--cpu-full always runs all rows)" if args.cpu_full == "auto"
                              else "--cpu-full never")}
        return out
    del X
    full = {"ran": False, "rows": int(args.rows), "wall_cap_s": args.cpu_wall_cap}
    out["full_n"] = full
    try:
        import psutil
        avail = float(psutil.virtual_memory().available)
    except ImportError:
        avail = 0.0
    need = 3.3 * args.rows * args.dim * 8.0       # rows + normalised copy + the constructor's transient square
    full.update({"host_free_GB": avail / 1e9, "host_need_GB": need / 1e9})
    if need > 0.85 * avail:
        full["why"] = "host memory: the oracle on all rows needs %.0f GB, %.0f GB are free" % (need / 1e9, avail / 1e9)
        return out
    t_it = 1.0 / max(its_sample * scale, 1e-12)                       # predicted seconds per faithful iteration at full N
    setup = (gen_s + ctor_s) / scale                                   # predicted generation + constructor
    spent = time.perf_counter() - t_leg
    n_it = int((args.cpu_wall_cap - setup) / t_it) - 1                 # one warm-up iteration
    full.update({"predicted_s_per_iteration": t_it, "predicted_setup_s": setup, "sample_leg_s": spent})
    if n_it < 2:
        full["why"] = "wall cap: set-up %.0f s + 3 iterations of %.1f s exceed %.0f s" % (setup, t_it, args.cpu_wall_cap)
        return out
    n_it = min(n_it, 12)
    try:
        t0 = time.perf_counter()
        X = host_rows(torch, args, args.rows)
        gen_full = time.perf_counter() - t0
        done, el, ctor_full = _time_oracle(args, X, "faithful", 1e30, 1, 1, n_it)
        del X
    except MemoryError as e:
        full["why"] = "MemoryError on the host (%s)" % e
        return out
    full.update({"ran": True, "iterations": done, "elapsed_s": el, "generation_s": gen_full, "constructor_s": ctor_full,
                 "iterations_per_s": done / el, "leg_wall_s": time.perf_counter() - t_leg})
    out["value"] = done / el
    out["sample_rows"] = int(args.rows)
    out["sample"] = ("oracle faithful mode (NumPy/OpenBLAS fp64, reference op sequence), %s, %s, ALL %d rows, d=%d: %d iterations "
                     "in %.1f s after 1 warm-up (constructor %.0f s, generation + copy %.0f s, %.0f GB free host memory before); "
                     "the %d-row sample scaled linearly gives %.4f it/s"
                     % (args.alg, what, args.rows, args.dim, done, el, ctor_full, gen_full, avail / 1e9, n_s, its_sample * scale))
    return out


def wait_for_rank0(dist, rank, key):
    """Ranks other than 0 block on the rendezvous store (a socket wait: off the CPU, unlike a spinning NCCL barrier that
    would take cores from rank 0's BLAS threads) until rank 0 has finished its host-only legs and set `key`."""
    from datetime import timedelta
    try:
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set(key, "1")
        else:
            store.wait([key], timedelta(hours=3))
    except Exception as e:      # (no store: the barrier that follows still orders the ranks, just not off the CPU)
        sys.stderr.write("bench.py: rank %d: store wait unavailable (%s)\n" % (rank, e))


# =====================================================================================================================
# greedy sparse-NNLS workloads (c2, c3, c4, ad hoc): step = one greedy iteration of HilbertCoreset's solver
# =====================================================================================================================
def load_synthetic(args, torch, solver):
    lo, hi = solver.row_begin, solver.row_end
    SUPER = 32 * GEN_BLOCK   # rows handed to the engine per ingest call (1 GiB of fp64 at d=512)
    buf = torch.empty(SUPER, args.dim, dtype=torch.float64, device="cuda")
    r = lo
    while r < hi:
        s0 = (r // GEN_BLOCK) * GEN_BLOCK                 # first generation block touching row r
        s1 = min(((hi + GEN_BLOCK - 1) // GEN_BLOCK) * GEN_BLOCK, s0 + SUPER,
                 ((args.rows + GEN_BLOCK - 1) // GEN_BLOCK) * GEN_BLOCK)
        for b0 in range(s0, s1, GEN_BLOCK):
            m = min(GEN_BLOCK, args.rows - b0)
            g = torch.Generator(device="cuda")
            g.manual_seed(args.seed * 1_000_003 + b0 // GEN_BLOCK)
            torch.randn(m, args.dim, dtype=torch.float64, device="cuda", generator=g, out=buf[b0 - s0:b0 - s0 + m])
        a, b = max(lo, s0), min(hi, s1, args.rows)
        solver.load_local(buf[a - s0:b - s0], a - lo)
        torch.cuda.synchronize()
        r = b
    del buf


def logistic_rows(args, torch, lo, hi):
    """Rows [lo, hi) of the simple_lr data set (examples/simple_lr/main.py:22-35: x ~ N(0, I_D), theta = 3 * 1,
    y ~ Bernoulli(sigmoid(x.theta)), z = y x), generated on the device per seeded 8192-row block."""
    D = args.features
    out = torch.empty(hi - lo, D, dtype=torch.float64, device="cuda")
    for b0 in range((lo // GEN_BLOCK) * GEN_BLOCK, hi, GEN_BLOCK):
        m = min(GEN_BLOCK, args.rows - b0)
        g = torch.Generator(device="cuda")
        g.manual_seed(args.seed * 1_000_003 + b0 // GEN_BLOCK)
        X = torch.randn(m, D, dtype=torch.float64, device="cuda", generator=g)
        u = torch.rand(m, dtype=torch.float64, device="cuda", generator=g)
        y = (u <= torch.sigmoid(3.0 * X.sum(dim=1))).double() * 2.0 - 1.0
        Zb = X * y[:, None]
        a, b = max(lo, b0), min(hi, b0 + m)
        out[a - lo:b - lo] = Zb[a - b0:b - b0]
    return out


def load_logistic(args, torch, dist, world, solver, info):
    """config 3: Laplace fit at the MAP over ALL rows (simple_lr/main.py:57-63; data sums all-reduced when sharded),
    S posterior samples, projection of the local rows on the device, ingest."""
    import model_lr
    import bayesiancoresets_amd as bc
    lo, hi = solver.row_begin, solver.row_end
    Z = logistic_rows(args, torch, lo, hi)

    def allreduce(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t
    t0 = time.perf_counter()
    mu, cov = model_lr.laplace_fit(Z, allreduce=allreduce if world > 1 else None)
    samples = np.random.RandomState(args.seed + 1).multivariate_normal(mu, cov, args.dim)   # simple_lr/main.py:74
    t1 = time.perf_counter()
    prj = bc.DeviceProjector("logistic", lambda n, w, p: samples[:n], args.dim, device=torch.cuda.current_device())
    prj.profile(True)
    # as HilbertCoreset does behind a device projector: raw log-likelihoods out of the projection kernel, the row means
    # (projector.py:21) subtracted by the solver's constructor pass -- N x S written once, read once
    vecs = prj.project_uncentred(Z)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    pms, pl, pfl = prj.profile_read()
    prj.profile(False)
    solver.load_local(vecs, center=True)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    vecs -= vecs.mean(dim=1, keepdim=True)          # (for the row-norm report and the CPU sample below; outside the timings)
    nrm = torch.linalg.vector_norm(vecs, dim=1)
    info.update({"laplace_fit_s": t1 - t0, "projection_s": t2 - t1, "ingest_s": t3 - t2, "features": args.features,
                 "projection_kernel_ms": pms, "projection_kernel_gelem_per_s": (hi - lo) * args.dim / max(pms, 1e-9) / 1e6,
                 "row_norm_min": float(nrm.min()), "row_norm_max": float(nrm.max())})
    sample = vecs[:min(args.cpu_rows or 1_000_000, hi - lo)].cpu().numpy() if lo == 0 else None
    del vecs, Z
    return sample


def run_snnls(args, torch, dist, nat, world, rank, local_rank):
    from bayesiancoresets_amd.sharded import ShardedSolver
    alg = {"giga": nat.ALG_GIGA, "fw": nat.ALG_FW, "omp": nat.ALG_OMP}[args.alg]
    store = {"float64": nat.F64, "float16": nat.F16}.get(args.dtype, nat.F32)
    elem = {nat.F64: 8, nat.F16: 2}.get(store, 4)
    info = {}
    solver = ShardedSolver(alg, args.rows, args.dim, device=local_rank, store_dtype=store,
                           keep_exact_rows=not args.no_exact_rows)
    cpu_sample = None
    if args.kind == "logistic":
        cpu_sample = load_logistic(args, torch, dist, world, solver, info)
    else:
        load_synthetic(args, torch, solver)
    rc = solver.finalize(None)
    if rc != nat.OK:
        raise SystemExit("finalize failed: %d" % rc)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def build(s, n):
        """solver.build with the safety net of the sharded path: if the device-side record exchange fails
        (it raises on every rank: timeout or trace mismatch), redo the work over the RCCL all-gather."""
        try:
            return s.build(n)
        except Exception as e:   # (EngineError in practice; anything else is treated the same way)
            if s.exchange != "mailbox":
                raise
            if rank == 0:
                print("bench: peer mailbox exchange failed (%s); falling back to the all-gather" % e, file=sys.stderr)
            s.fallback_to_collective()
            s.probe_info["reason"] = "mailbox failed mid-run (%s); fell back to the all-gather" % e
            s.engine.reset()
            return s.build(n)

    def timed(s, warmup, steps, solo=False):
        """(elapsed seconds over ranks, trace, scan ms total, scan launches, events-every) for exactly `steps` iterations
        (solo: `s` lives on this rank alone -- no barrier, no reduction over ranks)"""
        sync = (lambda: torch.cuda.synchronize()) if solo else sync_all
        if warmup > 0:
            build(s, warmup)
        if os.environ.get("BENCH_TEST_EXPIRE_MAILBOX") and s.exchange == "mailbox":
            s.engine.exchange_set_timeout(1e-7)      # tests: every later wait expires -> the fall-back above runs
        # an event pair costs 7-12 us of stream time: sample every 8th scan launch on long runs, time EVERY launch when
        # the timed region is short (< 64 steps: too few samples otherwise)
        every = 1 if steps < 64 else 8
        if not os.environ.get("BENCH_NO_EVENTS"):
            s.engine.profile(every)
        if s.exchange == "mailbox":
            s.engine.exchange_stats(reset=True)       # device-side wait stamps of the timed region only
        elif world > 1 and not solo and hasattr(s, "time_exchange"):
            s.time_exchange(True)                     # collective mode: events around every all-gather of the records
        sync()
        t0 = time.perf_counter()
        tr = build(s, steps)
        sync()
        el = time.perf_counter() - t0
        if world > 1 and not solo:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        ms, launches = s.engine.profile_read()
        s.engine.profile(False)
        return el, tr, ms, launches, every

    elapsed, tr, scan_ms, scan_launches, every = timed(solver, args.warmup, args.steps)
    sel, err, status = tr
    steps_done = len(sel)
    out = None
    # device-stamped exchange step of the timed region (mailbox mode): every rank's own view, worst rank reported
    xstat = None
    if world > 1 and solver.exchange == "collective":
        # no device-side stamps in this mode: the all-gather of the records on the stream's clock, worst / best rank
        us = solver.exchange_times_us()
        solver.time_exchange(False)
        mean, mx = (sum(us) / len(us), max(us)) if us else (0.0, 0.0)
        t = torch.tensor([mean, mx, mean, mx], dtype=torch.float64, device="cuda")
        lo_t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
        xstat = {"n": len(us), "max": [float(v) for v in t.tolist()], "min": [float(v) for v in lo_t.tolist()]}
    if world > 1 and solver.exchange == "mailbox":
        xs = solver.engine.exchange_stats()
        t = torch.tensor([xs["wait_us_mean"], xs["wait_us_max"], xs["total_us_mean"], xs["total_us_max"]],
                         dtype=torch.float64, device="cuda")
        lo_t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo_t, op=dist.ReduceOp.MIN)
        xstat = {"n": xs["exchanges"], "max": [float(v) for v in t.tolist()], "min": [float(v) for v in lo_t.tolist()]}
    # every rank's scan kernel (HIP events on its own stream): rows held and average launch duration
    per_rank = [[float(solver.n_local), scan_ms / max(scan_launches, 1), float(scan_launches)]]
    if world > 1:
        mine = torch.tensor(per_rank[0], dtype=torch.float64, device="cuda")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(v) for v in t.tolist()] for t in allr]
    if rank == 0:
        bytes_per_launch = float(solver.n_local) * args.dim * elem
        avg_ms = scan_ms / max(scan_launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # all shards together: the whole matrix's bytes over the SLOWEST shard's scan, against world x the HBM peak
        slowest_ms = max(r[1] for r in per_rank)
        agg = float(args.rows) * args.dim * elem / (slowest_ms * 1e-3) / 1e9 if slowest_ms > 0 else 0.0
        names = {"fw": "Frank-Wolfe", "giga": "GIGA", "omp": "OMP"}
        ran = ("M=%d greedy iterations" % steps_done if steps_done == args.steps else
               "%d of the requested M=%d greedy iterations (numeric limit reached, as in the reference)" % (steps_done, args.steps))
        if args.kind == "logistic":
            workload = ("Laplace-projected logistic-regression vectors (simple_lr pipeline, projection on the device) N=%d "
                        "features=%d S=d=%d, %s, %d row shard(s), %s"
                        % (args.rows, args.features, args.dim, names[args.alg], world, ran))
        else:
            workload = ("synthetic randn N=%d d=%d, %s, %d row shard(s), %s"
                        % (args.rows, args.dim, names[args.alg], world, ran))
        out = {
            "metric": "greedy coreset iterations/sec (N=%d, d=%d)" % (args.rows, args.dim),
            "value": steps_done / elapsed, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / max(steps_done, 1) * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None,
            "dtype": {nat.F32: "f32 scan + f64 state", nat.F16: "f16 rows, f32 scan + f64 state"}.get(store, "f64"),
            "data": "synthetic",
            "config": dict({
                "workload": workload, "name": "adhoc" if args.adhoc else args.config, "baseline_config": args.what,
                "alg": args.alg, "rows": args.rows, "dim": args.dim, "rows_per_gpu": solver.n_local,
                "exchange": solver.exchange if world > 1 else None,   # mailbox = device-side P2P stores; collective = all-gather
                "exchange_probe": solver.probe_info,
                # mailbox: wall_clock64 stamps inside the tail kernel (csrc/resolve.hip mailbox_exchange), timed region only;
                # collective: events around every all-gather of the records (wait == exchange there):
                # wait = from this shard's record being posted to the slowest peer's record arriving (load imbalance +
                # xGMI latency); exchange = wait + this shard's own G record stores.  Worst / best rank.
                "exchange_wait_us": xstat["max"][0] if xstat else None,
                "exchange_wait_us_max": xstat["max"][1] if xstat else None,
                "exchange_wait_us_best_rank": xstat["min"][0] if xstat else None,
                "exchange_us": xstat["max"][2] if xstat else None,
                "exchange_us_max": xstat["max"][3] if xstat else None,
                "exchanges_timed": xstat["n"] if xstat else None,
                "exact_rows_resident": not args.no_exact_rows,
                "iterations_run": int(steps_done), "reached_numeric_limit": bool(solver.reached_numeric_limit),
                "steps_accepted": int((status == 0).sum()), "final_error": float(err[-1]) if len(err) else None,
                "rescue": solver.engine.stats(),   # fp64 re-scored candidates / exact-scan fallbacks since construction
                # the same, flat (nested objects do not survive every consumer of this line)
                "rescue_exact_fallbacks": int(solver.engine.stats()["exact_fallbacks"]),
                "rescue_candidates_per_iteration": solver.engine.stats()["candidates"] / max(1.0, float(solver.engine.stats()["resolves"])),
            }, **info, **omp_fields(args, solver)),
            "roofline": {
                "bound": "hbm", "kernel": "scan_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args, solver.n_local),
                "avg_launch_ms": avg_ms, "launches": int(scan_launches), "timed_every": every,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                # achieved / peak / frac above: rank 0's GPU.  All `world` GPUs: N d sizeof over the slowest shard's scan
                "achieved_aggregate": agg, "peak_aggregate": HBM_PEAK_GBS * world, "frac_aggregate": agg / (HBM_PEAK_GBS * world),
                "per_gpu_rows": [int(r[0]) for r in per_rank], "per_gpu_avg_launch_ms": [r[1] for r in per_rank],
                "per_gpu_frac": [(r[0] * args.dim * elem / (r[1] * 1e-3) / 1e9 / HBM_PEAK_GBS) if r[1] > 0 else 0.0 for r in per_rank],
            },
        }
    # ---- row shards: the same iterations over the OTHER exchange mode (RCCL all-gather of the records per iteration, host
    # driven) right after the headline's timed region, so a multi-GPU line can be read whichever mode the probe chose ----
    if world > 1 and solver.exchange == "mailbox" and not args.no_side_legs:
        n_c = min(args.steps, 100)
        solver.fallback_to_collective()
        solver.engine.reset()
        el_c, tr_c, _, _, _ = timed(solver, min(args.warmup, 5), n_c)
        us_c = solver.exchange_times_us()
        solver.time_exchange(False)
        tc = torch.tensor([sum(us_c) / max(len(us_c), 1), max(us_c) if us_c else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(tc, op=dist.ReduceOp.MAX)
        if rank == 0:
            out["mailbox_ms_per_step"] = out["ms_per_step"]
            out["collective_ms_per_step"] = el_c / max(len(tr_c[0]), 1) * 1e3
            out["config"]["collective_leg"] = {
                "what": "same shards, records exchanged by one RCCL all-gather per iteration (host-driven scan / apply launches) "
                        "instead of the device-side peer mailbox; %d iterations after a reset" % len(tr_c[0]),
                "ms_per_step": out["collective_ms_per_step"], "iterations_per_s": len(tr_c[0]) / el_c,
                # the all-gather itself on the stream's clock (events around the call), worst rank: mean / max microseconds
                "exchange_us": float(tc[0].item()), "exchange_us_max": float(tc[1].item()), "exchanges_timed": len(us_c),
                "final_error": float(tr_c[1][-1]) if len(tr_c[1]) else None}
    # ---- row shards: the same workload as ONE shard on rank 0's GPU, in the same run (the other ranks wait off the CPU, see
    # the end of this function): the line then carries its own 1-GPU figure and the scaling efficiency against it ----------
    if world > 1 and rank == 0 and args.kind == "synthetic" and not args.no_side_legs and not args.no_one_rank_leg:
        try:
            keep = not args.no_exact_rows
            need = float(args.rows) * args.dim * (elem + (8 if keep else 0)) + float(32 * GEN_BLOCK) * args.dim * 8 + 4e9
            free = float(torch.cuda.mem_get_info()[0])
            if free < need:
                out["one_rank_error"] = "device memory: one shard of all rows needs %.0f GB, %.0f GB are free" % (need / 1e9, free / 1e9)
            else:
                one = ShardedSolver(alg, args.rows, args.dim, device=local_rank, store_dtype=store, keep_exact_rows=keep, solo=True)
                load_synthetic(args, torch, one)
                if one.finalize(None) != nat.OK:
                    raise RuntimeError("finalize failed")
                n1 = min(args.steps, 100)
                el1, tr1, ms1, l1, _ = timed(one, args.warmup, n1, solo=True)
                its1 = len(tr1[0]) / el1
                b1 = float(args.rows) * args.dim * elem
                out["one_rank_its"] = its1
                out["one_rank_ms_per_step"] = el1 / max(len(tr1[0]), 1) * 1e3
                out["one_rank_scan_frac"] = (b1 / (ms1 / max(l1, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms1 > 0 else None
                out["one_rank_same_selections"] = bool(all(int(x) == int(y) for x, y in zip(tr1[0], sel)))
                out["speedup_vs_one_rank"] = out["value"] / its1
                out["scaling_efficiency"] = out["value"] / its1 / world
                out["config"]["one_rank_leg"] = ("the same rows as one shard on rank 0's GPU, %d iterations after the same %d warm-up "
                                                 "iterations, while the other ranks wait off the CPU" % (len(tr1[0]), args.warmup))
                del one
                torch.cuda.empty_cache()
        except Exception as e:      # the headline line must survive a broken side leg
            out["one_rank_error"] = "%s: %s" % (type(e).__name__, e)
    # ---- the same workload with fp64-stored rows (the reference's own arithmetic end to end), c4 line only ----------
    if args.config == "c4" and not args.adhoc and store == nat.F32 and not args.no_exact_mode and world == 1:
        del solver
        torch.cuda.empty_cache()
        ex = ShardedSolver(alg, args.rows, args.dim, device=local_rank, store_dtype=nat.F64, keep_exact_rows=False)
        load_synthetic(args, torch, ex)
        if ex.finalize(None) != nat.OK:
            raise SystemExit("finalize (exact mode) failed")
        n_ex = min(args.steps, 60)
        el, tr2, ms2, l2, ev2 = timed(ex, min(args.warmup, 5), n_ex)
        if rank == 0:
            b64 = float(ex.n_local) * args.dim * 8
            a2 = b64 / (ms2 / max(l2, 1) * 1e-3) / 1e9 if ms2 > 0 else 0.0
            out["config"]["exact_mode"] = {
                "what": "rows stored in fp64 (dtype='float64'): every score is the reference's fp64 dot product, no candidate window",
                "iterations_per_s": len(tr2[0]) / el, "ms_per_step": el / max(len(tr2[0]), 1) * 1e3, "steps": n_ex,
                "scan_GBps": a2, "scan_frac_of_hbm_peak": a2 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b64,
            }
            # the same figures flat, at the top level and in config: the reference's own arithmetic end to end (fp64 rows)
            out["exact_mode_its"] = out["config"]["exact_mode_its"] = len(tr2[0]) / el
            out["exact_mode_frac"] = out["config"]["exact_mode_frac"] = a2 / HBM_PEAK_GBS
            out["config"]["exact_mode_ms_per_step"] = el / max(len(tr2[0]), 1) * 1e3
        del ex
        torch.cuda.empty_cache()
        # ---- and with the rows stored in fp16 (opt-in storage: half the bytes per iteration; the interval filter widens by
        # the storage term and the fp64 re-score on the resident raw rows decides, so the selections have to be THE SAME as
        # the fp32 run's -- checked here, iteration by iteration, on the same warm-up + steps) -----------------------------
        if not args.no_f16_leg:
            hs = ShardedSolver(alg, args.rows, args.dim, device=local_rank, store_dtype=nat.F16, keep_exact_rows=True)
            load_synthetic(args, torch, hs)
            if hs.finalize(None) != nat.OK:
                raise SystemExit("finalize (fp16 rows) failed")
            el3, tr3, ms3, l3, ev3 = timed(hs, args.warmup, args.steps)
            if rank == 0:
                b16 = float(hs.n_local) * args.dim * 2
                a3 = b16 / (ms3 / max(l3, 1) * 1e-3) / 1e9 if ms3 > 0 else 0.0
                same = len(tr3[0]) == len(sel) and all(int(x) == int(y) for x, y in zip(tr3[0], sel))
                st3 = hs.engine.stats()
                out["f16_rows_its"] = out["config"]["f16_rows_its"] = len(tr3[0]) / el3
                out["f16_rows_same_selections"] = out["config"]["f16_rows_same_selections"] = bool(same)
                out["config"]["f16_rows"] = {
                    "what": "rows stored in fp16 (dtype='float16'), fp32 accumulation, fp64 re-score of the candidates on the resident raw rows: "
                            "not the headline (north_star: fp32 rows), reported beside it",
                    "iterations_per_s": len(tr3[0]) / el3, "ms_per_step": el3 / max(len(tr3[0]), 1) * 1e3, "steps": len(tr3[0]),
                    "scan_GBps": a3, "scan_frac_of_hbm_peak": a3 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b16,
                    "same_selections_as_fp32_rows": bool(same), "max_rel_error_difference": float(
                        max((abs(float(x) - float(y)) / max(abs(float(y)), 1e-300) for x, y in zip(tr3[1], err)), default=0.0)),
                    "rescue": {"exact_fallbacks": int(st3.get("exact_fallbacks", 0)), "candidates": int(st3.get("candidates", 0)),
                               "resolves": int(st3.get("resolves", 0))},
                }
            del hs
    # ---- the reference's arithmetic on this box's host cores, on EVERY line (1, 2, 4, 8 ranks): rank 0 computes, the other
    # ranks sleep on the rendezvous store meanwhile ----------------------------------------------------------------------
    if rank == 0 and not args.no_cpu_baseline:
        if args.kind == "logistic":
            out["cpu_baseline"] = cpu_baseline_snnls(args, torch, world, "Laplace-projected logistic vectors", sample=cpu_sample)
        else:
            out["cpu_baseline"] = cpu_baseline_snnls(args, torch, world, "synthetic randn")
        out["cpu_baseline_onepass"] = out["cpu_baseline"]["onepass_value"]      # flat: it/s, scaled to N like `value`
        out["speedup_vs_cpu_baseline"] = out["value"] / max(out["cpu_baseline"]["value"], 1e-300)
    if world > 1:
        wait_for_rank0(dist, rank, "bench_host_legs_done_%s" % args.config)
    return out


def omp_fields(args, solver):
    """OMP step diagnostics of csrc/omp_lh.hip, flat: how many steps ran, columns that left the passive set, from-scratch
    re-solves (none expected: the inverse is kept in double-double)."""
    if args.alg != "omp" or not hasattr(solver.engine, "omp_stats"):
        return {}
    st = solver.engine.omp_stats()
    return {"omp_steps": int(st["steps"]), "omp_columns_left": int(st["columns_left"]), "omp_resolves": int(st["resolves"])}


def measured_traffic(args, n_local):
    """HBM bytes per scan launch from the committed PMC pass (profiles/scan_traffic.json) -- only if that pass was taken
    with THIS tree's scan kernel (csrc/scan.hip, its headers and the compiler flags: tools/stamp.py kernel_digest);
    otherwise null rather than a stale constant."""
    tpath = os.path.join(ROOT, "profiles", "scan_traffic.json")
    try:
        from tools.stamp import source_digest, kernel_digest
        tj = json.load(open(tpath))
        if tj.get("_stamp_scan", None) != kernel_digest("scan") and tj.get("_stamp") != source_digest():
            return None
        return tj.get("%s_n%d_d%d_%s" % (args.alg, n_local, args.dim, args.dtype))
    except Exception:
        return None


def projection_traffic(n_local, D, S):
    """HBM bytes per full-data projection launch (COLSUM, linear-regression family) from the committed FETCH_SIZE pass,
    under the same stamp rule as the scan's."""
    try:
        from tools.stamp import source_digest, kernel_digest
        tj = json.load(open(os.path.join(ROOT, "profiles", "scan_traffic.json")))
        if tj.get("_stamp_proj", None) != kernel_digest("proj") and tj.get("_stamp") != source_digest():
            return None
        return tj.get("proj_colsum_linreg_n%d_d%d_s%d" % (n_local, D, S))
    except Exception:
        return None


# =====================================================================================================================
# config 5: SparseVI on the RBF-basis regression; step = one greedy step = (1 + opt_itrs) full-data projections
# =====================================================================================================================
def run_sparsevi(args, torch, dist, nat, world, rank, local_rank):
    import bayesiancoresets_amd as bc
    import rbf_workload
    import model_linreg
    from bayesiancoresets_amd.sharded import shard_bounds
    N, S, nb = args.rows, args.dim, 50
    D = 6 * nb + 1
    lo, hi = shard_bounds(N, world)[0][rank]
    # observations: every rank draws the same small pilot (basis centres, prior statistics) and its own rows
    rs = np.random.RandomState(args.seed)
    pilot = rbf_workload.synthetic_observations(200_000, rs)
    scales, centres = rbf_workload.basis_layout(pilot, nb, rs)
    std, mean = pilot[:, 2].std(), pilot[:, 2].mean()
    mu0, Sig0, sigsq = mean * np.ones(D), (std ** 2 + mean ** 2) * np.eye(D), float(std ** 2)
    obs = torch.empty(hi - lo, 3, dtype=torch.float64, device="cuda")
    for b0 in range((lo // GEN_BLOCK) * GEN_BLOCK, hi, GEN_BLOCK):
        m = min(GEN_BLOCK, N - b0)
        g = torch.Generator(device="cuda")
        g.manual_seed(args.seed * 1_000_003 + b0 // GEN_BLOCK)
        loc = torch.rand(m, 2, dtype=torch.float64, device="cuda", generator=g)
        eps = torch.randn(m, dtype=torch.float64, device="cuda", generator=g)
        price = 5.3 + 0.35 * torch.sin(3.0 * loc[:, 0]) * torch.cos(2.0 * loc[:, 1]) + 0.25 * loc[:, 0] * loc[:, 1] + 0.15 * eps
        a, b = max(lo, b0), min(hi, b0 + m)
        obs[a - lo:b - lo, :2] = loc[a - b0:b - b0]
        obs[a - lo:b - lo, 2] = price[a - b0:b - b0]
    Z = rbf_workload.design_rows_device(torch, obs, scales, centres)       # (hi - lo) x 302, resident
    del obs
    group = dist.group.WORLD if world > 1 else None
    sampler = model_linreg.posterior_sampler(mu0, Sig0, sigsq, device="cuda", seed=args.seed + 7)
    np.random.seed(args.seed)
    prj = bc.DeviceProjector("linreg", sampler, S, sigsq=sigsq, device=local_rank, group=group, row_offset=lo, colsum=args.colsum)
    alg = bc.SparseVICoreset(Z, prj, opt_itrs=args.opt_itrs, row_offset=lo, group=group) if world > 1 else \
        bc.SparseVICoreset(Z, prj, opt_itrs=args.opt_itrs)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
    if args.warmup > 0:
        alg.build(args.warmup)
    prj.profile(True)
    sync()
    k_curve = [int(x) for x in str(args.k_curve).split(",")] if getattr(args, "k_curve", None) else None
    per_step = None
    t0 = time.perf_counter()
    if k_curve:
        # every greedy step on its own clock (the weight read-back at the end of a step synchronises anyway): wall seconds,
        # the coreset size the step's ADAM loop ran at, milliseconds of that step's select projection kernel
        per_step = []
        for _ in range(args.steps):
            km0 = prj.profile_read()[0]
            ts = time.perf_counter()
            alg.build(1)
            torch.cuda.synchronize()
            per_step.append((time.perf_counter() - ts, int(alg.wts.shape[0]), prj.profile_read()[0] - km0))
    else:
        alg.build(args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kms, launches, flops = prj.profile_read()
    prj.profile(False)
    host_leg = world > 1 and not args.no_cpu_baseline
    if rank != 0:
        if host_leg:
            wait_for_rank0(dist, rank, "bench_host_legs_done_c5_%s" % args.colsum)      # (rank 0 times the oracle meanwhile)
        return None
    # (the k-point core projections are launches too: their flops and time are in the totals, weight ~1e-5)
    per_launch_flops = 2.0 * (hi - lo) * D * S
    achieved = flops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
    out = {
        "metric": "SparseVI greedy steps/sec (N=%d, D=%d, S=%d, opt_itrs=%d)" % (N, D, S, args.opt_itrs),
        "value": args.steps / elapsed, "unit": "greedy steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": "SparseVICoreset on the synthetic RBF-basis linear regression (6 scales x 50 bases + 1 = %d columns), "
                        "N=%d, S=%d Monte-Carlo samples, opt_itrs=%d, %d row shard(s), %d greedy steps"
                        % (D, N, S, args.opt_itrs, world, args.steps),
            "name": "c5", "baseline_config": args.what, "rows": N, "features": D, "dim": S, "rows_per_gpu": hi - lo,
            "opt_itrs": args.opt_itrs, "projections_per_step": 1 + args.opt_itrs if args.colsum == "mfma" else 1,
            # mfma: every column sum is a fused fp64-MFMA projection of all rows (1 + opt_itrs per step, as the reference
            # does); moments: the column sums come in closed form from the one-time (D+1) x (D+1) moments of the data
            # (csrc/moments.hip), the select step's projection stays
            "colsum": args.colsum, "moments": prj.moments_info or None,
            "moments_setup_ms": (prj.moments_info or {}).get("setup_ms"),
            "sampler": "weighted conjugate posterior on the device (bc.LinregPosteriorSampler via examples/common/model_linreg.py: "
                       "rank-k correction of the prior's %d x %d factor through a k x k Cholesky, k = coreset size, from "
                       "weights that stay on the device)" % (D, D),
            "adam_loop": "enqueued: opt_itrs x (draws, column sums, coreset projection, ADAM step) without host synchronisation, "
                         "one read-back per greedy step (csrc/svi.hip)" if alg._enqueue_plan() is not None else "host loop (nn_opt)",
            "coreset_size": int(alg.size()), "coreset_points": int(alg.wts.shape[0]), "coreset_idcs": [int(i) for i in alg.idcs],
            "projection_ms_per_step_kernels": kms / args.steps,
        },
        "roofline": {
            "bound": "mfma", "kernel": "proj_kernel", "achieved": achieved, "peak": F64_MFMA_PEAK_TF, "unit": "TFLOP/s",
            "frac": achieved / F64_MFMA_PEAK_TF, "traffic": projection_traffic(hi - lo, D, S),
            "avg_launch_ms": kms / max(launches, 1), "launches": int(launches),
            "algorithmic_flops_per_full_launch": per_launch_flops,
            "attainable_note": "register-only loops on this chip sustain 72.8 TFLOP/s with v_mfma_f64_4x4x4_4b_f64 (the form "
                               "the kernel uses) and 47.6 with v_mfma_f64_16x16x4_f64 (tools/probe/mfma_f64_peak.hip, "
                               "profiles/r02_mfma_f64_probe.txt)",
        },
    }
    if per_step:
        # steps/s and microseconds per ADAM step (everything of the step that is not the select projection kernel, over
        # opt_itrs) at the asked coreset sizes: the mean over the steps whose loop ran at k - 2 .. k + 2 points
        curve = {}
        for kq in k_curve:
            win = [(t, km) for (t, kk, km) in per_step if abs(kk - kq) <= 2]
            if not win:
                continue
            tm = sum(t for t, _ in win) / len(win)
            km = sum(m for _, m in win) / len(win)
            curve[str(kq)] = {"steps_s": 1.0 / tm, "ms_per_step": tm * 1e3, "select_kernel_ms": km,
                              "adam_step_us": (tm * 1e3 - km) / max(args.opt_itrs, 1) * 1e3, "steps_in_window": len(win)}
        out["k_curve"] = curve
        out["per_step_seconds_first64"] = sum(t for t, _, _ in per_step[:64])
        out["steps_timed"] = len(per_step)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_sparsevi(args, Z, mu0, Sig0, sigsq, S)
        out["speedup_vs_cpu_baseline"] = out["value"] / max(out["cpu_baseline"]["value"], 1e-300)
    if host_leg:
        wait_for_rank0(dist, rank, "bench_host_legs_done_c5_%s" % args.colsum)
    return out


def cpu_baseline_sparsevi(args, Z, mu0, Sig0, sigsq, S):
    """oracle/sparsevi_oracle.py (NumPy restatement of sparsevi.py + projector.py + model_linreg.py, host sampler of
    linear_regression/main.py:141-147) timed on a bounded sample of rows: one greedy step after one warm-up step."""
    from oracle.sparsevi_oracle import SparseVIOracle, linreg_loglik
    import model_linreg
    n_s = min(20_000, Z.shape[0])
    Zs = Z[:n_s].cpu().numpy()
    np.random.seed(args.seed)
    itrs = min(args.opt_itrs, 25)      # keep the sample to tens of seconds; cost is linear in 1 + opt_itrs
    blas_threads_for_baseline()
    o = SparseVIOracle(Zs, model_linreg.posterior_sampler(mu0, Sig0, sigsq), lambda z, th: linreg_loglik(z, th, sigsq), S,
                       opt_itrs=itrs)
    o.step()
    t0 = time.perf_counter()
    o.step()
    el = time.perf_counter() - t0
    scale = (n_s / float(args.rows)) * ((1.0 + itrs) / (1.0 + args.opt_itrs))
    return {
        "value": scale / el, "unit": "greedy steps/s", "cores": int(host_threads()), "kind": "port",
        "sample": "oracle/sparsevi_oracle.py (NumPy/OpenBLAS fp64: sparsevi.py select + ADAM optimise, host sampler), first %d of "
                  "%d rows, opt_itrs=%d of %d: one greedy step in %.2f s, scaled linearly in N and in (1 + opt_itrs) (x%.6f)"
                  % (n_s, args.rows, itrs, args.opt_itrs, el, scale),
    }


def side_legs(args, out, torch, dist, nat):
    """The default (c4) line also carries compact legs of BASELINE.json's other GPU configs, each at its full size on
    this one GPU, as flat top-level keys (c2_* / c3_* / c5_*): the driver times only the default command, and these put
    configs[1], [2] and [4] on its clock.  No CPU baseline in the legs; a leg that fails leaves `<cfg>_error`."""
    import copy
    import gc

    def leg(name, **over):
        a = copy.copy(args)
        c = CONFIGS[name]
        a.config, a.kind, a.alg, a.rows, a.dim, a.what, a.adhoc = name, c["kind"], c["alg"], c["rows"], c["dim"], c["what"], False
        a.steps = {"sparsevi": 3, "logistic": 100}.get(a.kind, 300)
        a.warmup = {"sparsevi": 1, "logistic": 20}.get(a.kind, 30)
        a.no_cpu_baseline, a.no_exact_mode, a.dtype = True, True, "float32"
        for k, v in over.items():
            setattr(a, k, v)
        gc.collect()
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        try:
            r = (run_sparsevi if a.kind == "sparsevi" else run_snnls)(a, torch, dist, nat, 1, 0, torch.cuda.current_device())
        except Exception as e:      # the headline line must survive a broken side leg
            out["%s_error" % name] = "%s: %s" % (type(e).__name__, e)
            return None
        r["leg_wall_s"] = time.perf_counter() - t0
        return r

    for name in ("c2", "c3"):
        r = leg(name)
        if r is None:
            continue
        bytes_it = r["roofline"]["algorithmic_bytes_per_launch"]
        out["%s_its" % name] = r["value"]
        out["%s_ms_per_step" % name] = r["ms_per_step"]
        out["%s_scan_frac" % name] = r["roofline"]["frac"]
        # whole-iteration fraction of the HBM peak: the scan's algorithmic bytes over the full step time
        out["%s_iter_frac" % name] = bytes_it / (r["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["%s_steps" % name] = r["config"]["iterations_run"]
        out["%s_workload" % name] = r["config"]["workload"]
        out["%s_leg_wall_s" % name] = r["leg_wall_s"]
        if name == "c3":
            # everything of an OMP iteration that is not the scan: the fused resolve + Lawson-Hanson step kernel
            out["c3_omp_step_us"] = (r["ms_per_step"] - r["roofline"]["avg_launch_ms"]) * 1e3
            out["c3_final_error"] = r["config"]["final_error"]
    # configs[1] at its stated M = 1000 from a fresh solver (no warm-up): GIGA reaches its numeric limit on the way (the latch
    # of snnls.py:63-74) -- the whole run, latch included, on the driver's clock
    r = leg("c2", steps=1000, warmup=0)
    if r is not None:
        out["c2_m1000_its"] = r["value"]
        out["c2_m1000_iterations_run"] = r["config"]["iterations_run"]
        out["c2_m1000_reached_numeric_limit"] = r["config"]["reached_numeric_limit"]
        out["c2_m1000_steps_accepted"] = r["config"]["steps_accepted"]
        out["c2_m1000_final_error"] = r["config"]["final_error"]
        out["c2_m1000_leg_wall_s"] = r["leg_wall_s"]
    # configs[3] (the headline's workload) at ITS stated M = 1000 as well, from a fresh solver: Frank-Wolfe's whole error
    # trajectory on the driver's clock (the headline times 20 iterations; the rate per iteration is flat)
    r = leg("c4", steps=1000, warmup=0, no_f16_leg=True)
    if r is not None:
        out["c4_m1000_its"] = r["value"]
        out["c4_m1000_iterations_run"] = r["config"]["iterations_run"]
        out["c4_m1000_scan_frac"] = r["roofline"]["frac"]
        out["c4_m1000_final_error"] = r["config"]["final_error"]
        out["c4_m1000_leg_wall_s"] = r["leg_wall_s"]
    # c5, MFMA form (every column sum a projection of all rows, as the reference does): 3 greedy steps after 1
    r = leg("c5", colsum="mfma")
    if r is not None:
        out["c5_steps_s"] = r["value"]
        out["c5_ms_per_step"] = r["ms_per_step"]
        out["c5_leg_wall_s"] = r["leg_wall_s"]
        out["c5_mfma_frac"] = r["roofline"]["frac"]
        out["c5_mfma_tflops"] = r["roofline"]["achieved"]
        out["c5_workload"] = r["config"]["workload"]
        out["c5_coreset_idcs"] = r["config"]["coreset_idcs"]
    # c5, closed-form column sums: the coreset grown from empty to the reference experiment's 300 points
    # (examples/linear_regression/main.py:284 coreset_size_max), every greedy step timed; steps/s over the first 64 steps and
    # at k = 8 / 32 / 64 / 128 / 300, with the microseconds of one ADAM step there
    r = leg("c5", colsum="moments", steps=304, warmup=0, k_curve="8,32,64,128,300")
    if r is not None:
        n64 = min(64, r.get("steps_timed", 0))
        out["c5_steps_s_moments"] = n64 / r["per_step_seconds_first64"] if n64 else None
        out["c5_ms_per_step_moments"] = r["per_step_seconds_first64"] / n64 * 1e3 if n64 else None
        out["c5_moments_steps"] = r.get("steps_timed")
        out["c5_moments_coreset_points"] = r["config"]["coreset_points"]
        out["c5_leg_wall_s_moments"] = r["leg_wall_s"]
        out["c5_moments_mfma_frac"] = r["roofline"]["frac"]
        ref_idcs = out.get("c5_coreset_idcs")
        out["c5_moments_same_idcs"] = (r["config"]["coreset_idcs"][:len(ref_idcs)] == ref_idcs) if ref_idcs else None
        out["c5_moments_setup_ms"] = r["config"].get("moments_setup_ms")
        out["c5_adam_loop"] = r["config"].get("adam_loop")
        for kq, rec in r.get("k_curve", {}).items():
            out["c5_steps_s_at_k%s" % kq] = rec["steps_s"]
            out["c5_adam_step_us_at_k%s" % kq] = rec["adam_step_us"]
        out["c5_select_kernel_ms"] = (r.get("k_curve", {}).get("300") or {}).get("select_kernel_ms")
        # which kernels an ADAM step is made of at each size: profiles/r06_c5_adam_k*.txt (rocprofv3 traces of tools/c5_ksweep.py)
        out["c5_adam_kernels"] = ("k <= 4 + 2 ceil(D / 32) (= 24): lrs_apply_kernel (rank-k form, every workgroup repeats the k x k "
                                  "Cholesky) dominates from k ~ 8; beyond: lrp_chol_kernel (cooperative D x D Cholesky + inverse on one XCD, "
                                  "~56 % of the step), then proj_mid_quad_kernel (closed-form column sums and the coreset points' "
                                  "projection in one launch), lrp_form_kernel, svi_adam_a / b_kernel, lrp_draw_kernel")
    gram_leg(out, torch, nat)
    optimize_leg(out, torch, nat)


def gram_leg(out, torch, nat):
    """MFMA utilisation of the dense re-weight on the driver's clock: the Gram operator optimize() forms over its active rows
    (bcx_gram: csrc/moments.hip gram_tile_kernel + moments_reduce_kernel), timed by events on the stream it runs on; flops =
    the upper triangle's k (k + 1) d, peak = the fp64 MFMA peak.  Flat keys reweight_gram_*."""
    try:
        lib = nat.load()
        st = int(torch.cuda.current_stream().cuda_stream)
        for k, d in ((1497, 1024), (4096, 1024)):
            g = torch.Generator(device="cuda")
            g.manual_seed(99)
            V = torch.randn(k, d, dtype=torch.float64, device="cuda", generator=g)
            G = torch.empty(k, k, dtype=torch.float64, device="cuda")
            need = int(lib.bcx_gram_scratch_bytes(k, d))
            work = torch.empty((need + 7) // 8, dtype=torch.float64, device="cuda")
            call = lambda: lib.bcx_gram(st, V.data_ptr(), k, d, d, G.data_ptr(), k, work.data_ptr(), work.numel() * 8)
            for _ in range(3):
                if call() != 0:
                    raise RuntimeError(lib.bcx_project_last_error().decode())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 30
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / reps
            tf = float(k) * (k + 1) * d / us / 1e6
            ref = V[:64] @ V.T
            ok = bool(((G[:64] - ref).abs().max() / ref.abs().max()).item() < 1e-12)
            out["reweight_gram_k%d_us" % k] = us
            out["reweight_gram_k%d_tflops" % k] = tf
            out["reweight_gram_k%d_mfma_frac" % k] = tf / F64_MFMA_PEAK_TF
            out["reweight_gram_k%d_checked" % k] = ok
            del V, G, work
    except Exception as e:      # the headline line must survive a broken side leg
        out["reweight_gram_error"] = "%s: %s" % (type(e).__name__, e)


def optimize_leg(out, torch, nat):
    """optimize() (snnls.py:82-97) on the driver's clock: Frank-Wolfe supports of ~1500 points on synthetic randn rows, N = 1e6 --
    d = 1024 (k > d: dependent columns, hundreds of pivots after the warm start) and d = 2048 (k < d: the warm start is the
    answer).  Flat keys reweight_optimize_*: wall ms of the call, the warm start's passive set and the pivots after it, and
    the share of the call spent in the fp64-MFMA kernels that form the two Gram matrices (the support's and the inverse's),
    timed on their own on the same shapes."""
    import ctypes
    try:
        lib = nat.load()
        lib.bcx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        for d in (1024, 2048):
            N, its = 1_000_000, 1500
            eng = nat.Engine(nat.ALG_FW, N, d)
            g = torch.Generator(device="cuda")
            g.manual_seed(1)
            piece = 256 * 1024                     # (pieces start on the engine's 1024-row chunk boundaries)
            for r0 in range(0, N, piece):
                m = min(piece, N - r0)
                x = torch.randn(m, d, device="cuda", dtype=torch.float64, generator=g)
                eng.load_device_rows(x.data_ptr(), m, d, True, r0)
                torch.cuda.synchronize()
            del x
            if eng.finalize(None) != 0:
                raise RuntimeError("finalize failed")
            eng.run_build(its, 1e-12)
            k = len(eng.sparse_weights()[0])
            e0 = eng.error()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ok = eng.optimize(1e-12)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            st = (ctypes.c_longlong * 32)()
            lib.bcx_debug_stamps(eng.h, st)
            # the two Gram operators of the call, alone: k x d (the support) and kp x kp (Y Y^T = the inverse)
            kp = (k + 63) // 64 * 64
            gram_us = 0.0
            for (kk, dd) in ((k, d), (kp, kp)):
                V = torch.randn(kk, dd, dtype=torch.float64, device="cuda")
                G = torch.empty(kk, kk, dtype=torch.float64, device="cuda")
                work = torch.empty((int(lib.bcx_gram_scratch_bytes(kk, dd)) + 7) // 8, dtype=torch.float64, device="cuda")
                stream = int(torch.cuda.current_stream().cuda_stream)
                call = lambda: lib.bcx_gram(stream, V.data_ptr(), kk, dd, dd, G.data_ptr(), kk, work.data_ptr(), work.numel() * 8)
                call()
                e_0, e_1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e_0.record()
                for _ in range(10):
                    call()
                e_1.record()
                torch.cuda.synchronize()
                gram_us += e_0.elapsed_time(e_1) * 1e3 / 10
                del V, G, work
            key = "reweight_optimize_k%d_d%d" % (round(k, -2), d)
            out[key + "_ms"] = ms
            out[key + "_k"] = int(k)
            out[key + "_accepted"] = bool(ok)
            out[key + "_error_before"] = e0
            out[key + "_error_after"] = eng.error()
            out[key + "_warm"] = {"started_warm": int(st[20]), "passive_set": int(st[21]), "entered": int(st[23]), "left": int(st[24]),
                                  "closing_check_failed": int(st[26])}
            out[key + "_gram_us"] = gram_us
            out[key + "_mfma_share"] = gram_us / (ms * 1e3)
            if d == 1024:      # the share of the call the review asked for by name (k = 1497, d = 1024)
                out["reweight_mfma_share"] = gram_us / (ms * 1e3)
            eng.close()
            del eng
            torch.cuda.empty_cache()
    except Exception as e:      # the headline line must survive a broken side leg
        out["reweight_optimize_error"] = "%s: %s" % (type(e).__name__, e)


def mailbox_preflight_child():
    """`bench.py --mailbox-preflight` (one short-lived helper process per rank, started by mailbox_preflight below): map the
    peer mailboxes of a tiny row-sharded solver across the ranks' GPUs, run the probe and twenty greedy iterations through
    the device-side exchange, exit 0.  Anything else -- an exception, a GPU fault that kills the process, a hang past the
    parent's timeout -- tells the parent that this box cannot run the mailbox exchange."""
    import torch
    import torch.distributed as dist
    from bayesiancoresets_amd import _native as nat
    from bayesiancoresets_amd.sharded import ShardedSolver
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dev = 0 if os.environ.get("BENCH_SHARE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")                  # host-side rendezvous only (handles, agreement); the exchange itself is device-side
    if os.environ.get("BENCH_PREFLIGHT_CRASH") == str(rank):
        os.abort()                                   # tests: a helper that dies the way a GPU fault kills a process
    n, d = world * 4096, 64
    s = ShardedSolver(nat.ALG_FW, n, d, device=dev)
    g = torch.Generator(device="cuda")
    g.manual_seed(1234 + s.row_begin)
    s.load_local(torch.randn(s.n_local, d, dtype=torch.float64, device="cuda", generator=g))
    torch.cuda.synchronize()
    if s.finalize(None) != nat.OK:
        raise SystemExit("preflight: finalize failed")
    if s.exchange != "mailbox":
        raise SystemExit("preflight: mailbox unavailable (%s)" % s.probe_info.get("reason"))
    s.build(20)                                      # (ends with the cross-rank trace check)
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print("preflight ok: rank %d of %d, exchange %s" % (rank, world, s.exchange), flush=True)


def mailbox_preflight(world, rank, local_rank):
    """Before a multi-GPU run commits its own process to the device-side record exchange (hipIpc-mapped peer mailboxes,
    system-scope stores over xGMI -- never exercised across physical GPUs on the boxes this was developed on), every rank
    tries it in a throw-away helper process.  Returns (ok, why).  A helper that crashes or hangs costs the helper, not
    the benchmark: the run then uses the RCCL all-gather exchange and says so (config.exchange_probe)."""
    import subprocess
    env = dict(os.environ)
    env["MASTER_PORT"] = str((int(os.environ.get("MASTER_PORT", "29500")) + 23) % 65000 + 500)
    env["BCX_EXCHANGE_TIMEOUT"] = "5"
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")       # (the helpers' gloo rendezvous stays on the loopback: one node, and the
                                                     # container's host name need not resolve)
    env.pop("BCX_EXCHANGE", None)
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)                             # (the helpers rendezvous by themselves, not through the launcher's agent store)
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--mailbox-preflight"], env=env, capture_output=True, text=True,
                             timeout=float(os.environ.get("BENCH_PREFLIGHT_TIMEOUT", "180")))
        ok = out.returncode == 0 and "preflight ok" in out.stdout
        why = "" if ok else "helper exit code %d: %s" % (out.returncode, (out.stderr or out.stdout).strip().splitlines()[-1:] or "")
    except subprocess.TimeoutExpired:
        ok, why = False, "helper did not finish in time"
    return ok, why


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks here, one per GPU, under
    torch.distributed.run (RCCL) on 127.0.0.1 and pass rank 0's JSON line through.  Never prints a line for fewer ranks
    than --gpus asked for: too few devices is an error (exit 2) unless BENCH_SHARE_GPU=1 (testing: all ranks on cuda:0)."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and os.environ.get("BENCH_SHARE_GPU") != "1":
        sys.stderr.write("bench.py: --gpus %d but this box shows %d GPU(s); refusing to print a line for fewer ranks "
                         "(BENCH_SHARE_GPU=1 lets the ranks share cuda:0 for testing)\n" % (args.gpus, have))
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write("bench.py: no launcher in the environment, starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
    env = dict(os.environ)
    env["BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.call(cmd, env=env)


def main():
    if "--mailbox-preflight" in sys.argv[1:]:
        return mailbox_preflight_child()
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    import torch
    import torch.distributed as dist
    from bayesiancoresets_amd import _native as nat

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:      # never a line whose n_gpus differs from --gpus
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # BENCH_SHARE_GPU=1 (testing only): all ranks share cuda:0 and talk over gloo -- RCCL refuses two ranks
    # on one device; the driver's multi-GPU runs use one GPU per rank over RCCL ("nccl").
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d: LOCAL_RANK %d but %d visible GPU(s)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    backend = None
    preflight = None
    if world > 1:
        # one rank per GPU and the exchange mode not forced: try the device-side exchange in helper processes first
        want = os.environ.get("BENCH_PREFLIGHT", "0" if share else "1") == "1" and args.kind != "sparsevi" \
            and os.environ.get("BCX_EXCHANGE", "mailbox") == "mailbox"
        mine = mailbox_preflight(world, rank, local_rank) if want else None
        backend = "gloo" if share else "nccl"
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if want:
            flag = torch.tensor([0.0 if mine[0] else 1.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.SUM)
            failed = int(flag.item())
            preflight = {"ran": True, "ranks_failed": failed, "this_rank": "ok" if mine[0] else mine[1]}
            if failed:
                os.environ["BCX_EXCHANGE"] = "collective"       # every rank alike
                if rank == 0:
                    sys.stderr.write("bench.py: mailbox preflight failed on %d rank(s) (rank 0: %s); using the all-gather exchange\n"
                                     % (failed, "ok" if mine[0] else mine[1]))
    run = run_sparsevi if args.kind == "sparsevi" else run_snnls
    out = run(args, torch, dist, nat, world, rank, local_rank)
    if world == 1 and args.config == "c4" and not args.adhoc and not args.no_side_legs:
        side_legs(args, out, torch, dist, nat)
    if args.kind == "sparsevi" and args.colsum == "mfma" and not args.no_side_legs:
        # the same steps again with the column sums in closed form (every rank takes part), reported beside the headline
        import copy
        import gc
        a2 = copy.copy(args)
        a2.colsum, a2.no_cpu_baseline = "moments", True
        gc.collect()
        torch.cuda.empty_cache()
        r2 = run_sparsevi(a2, torch, dist, nat, world, rank, local_rank)
        if rank == 0:
            out["steps_s_moments"] = r2["value"]
            out["ms_per_step_moments"] = r2["ms_per_step"]
            out["moments_same_idcs"] = r2["config"]["coreset_idcs"] == out["config"]["coreset_idcs"]
            out["moments"] = r2["config"]["moments"]
            out["moments_select_tflops"] = r2["roofline"]["achieved"]
    if rank == 0:
        # the world the process group itself reports (RCCL ranks when backend == "nccl"), one rank per device
        out["rccl_ranks"] = dist.get_world_size() if (world > 1 and backend == "nccl") else (1 if world == 1 else 0)
        out["mailbox_preflight"] = preflight
        out["process_group"] = {"backend": backend, "world_size": dist.get_world_size() if world > 1 else 1,
                                "self_launched": os.environ.get("BENCH_SELF_LAUNCHED") == "1",
                                "devices_visible": torch.cuda.device_count(), "share_gpu": share}
        assert out["n_gpus"] == args.gpus
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
