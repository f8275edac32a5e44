#!/usr/bin/env python3
"""Headline benchmark: greedy coreset iterations/sec on synthetic N x d vectors.

    python bench.py [--gpus N --steps K --warmup W] [--alg fw|giga|omp] [--rows N] [--dim d]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one greedy iteration: one pass of the hot path (correlation scan over all N rows +
arg-max + reweight) -- BASELINE.json metric "greedy coreset iterations/sec (N=10M, d=512)".
Default workload = BASELINE.json configs[3] shape: N = 10,000,000 rows, d = 512, Frank-Wolfe,
row-sharded over --gpus ranks (total N fixed => "scaling": "strong").  Inputs are generated on
the device (seeded per 8192-row block, so the matrix is the same for every shard count) and are
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
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
GEN_BLOCK = 8192        # rows per seeded generation block (multiple of the 1024-row engine chunk)


PROFILE_EVERY = 8   # hipEvent pairs around every 8th scan launch of the timed region (a pair costs 7-12 us of stream time)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--alg", default="fw", choices=["fw", "giga", "omp"])
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64", "float16"],
                    help="storage of the normalised rows (state and re-score are always fp64)")
    ap.add_argument("--no-exact-rows", action="store_true", help="do not keep the raw fp64 rows resident")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=200_000, help="rows of the CPU-baseline sample")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def gen_block(torch, seed, block_id, rows, d, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1_000_003 + block_id)
    return torch.randn(rows, d, device=device, dtype=torch.float64, generator=g)


def cpu_baseline(args, alg, torch):
    """The oracle's faithful mode (reference op sequence: 5 passes of OpenBLAS dgemv/dgemm per
    iteration, fp64) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle.snnls_oracle import SnnlsOracle
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    n_s = min(args.cpu_rows, args.rows)
    n_s = (n_s // GEN_BLOCK) * GEN_BLOCK or n_s
    parts = []
    for blk in range((n_s + GEN_BLOCK - 1) // GEN_BLOCK):
        m = min(GEN_BLOCK, n_s - blk * GEN_BLOCK)
        parts.append(gen_block(torch, args.seed, blk, m, args.dim, "cuda").cpu().numpy())
    X = np.concatenate(parts, axis=0)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=alg, mode="faithful")
    o.build(3)  # warm-up
    done, t0 = 0, time.perf_counter()
    while True:
        o.build(2)
        done += 2
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or done >= 200:
            break
    its_sample = done / el
    scale = n_s / float(args.rows)   # cost per iteration is linear in N
    return {
        "value": its_sample * scale,
        "unit": "iterations/s",
        "cores": int(threads),
        "kind": "port",
        "sample": "oracle faithful mode (NumPy/OpenBLAS fp64, reference op sequence), %s, first %d of %d rows, "
                  "d=%d, %d iterations in %.1f s = %.2f it/s on the sample, scaled linearly in N (x%.4f)"
                  % (alg, n_s, args.rows, args.dim, done, el, its_sample, scale),
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from bayesiancoresets_amd import _native as nat
    from bayesiancoresets_amd.sharded import ShardedSolver

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # BENCH_SHARE_GPU=1 (testing only): all ranks share cuda:0 and talk over gloo -- RCCL refuses two ranks
    # on one device; the driver's multi-GPU runs use one GPU per rank over RCCL ("nccl").
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    alg = {"giga": nat.ALG_GIGA, "fw": nat.ALG_FW, "omp": nat.ALG_OMP}[args.alg]
    store = {"float64": nat.F64, "float16": nat.F16}.get(args.dtype, nat.F32)

    solver = ShardedSolver(alg, args.rows, args.dim, device=local_rank, store_dtype=store,
                           keep_exact_rows=not args.no_exact_rows)
    # ---- synthetic data, generated shard-locally, resident before timing ------------------
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
    rc = solver.finalize(None)
    if rc != nat.OK:
        raise SystemExit("finalize failed: %d" % rc)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def build(n):
        """solver.build with the safety net of the sharded path: if the device-side record exchange fails
        (it raises on every rank: timeout or trace mismatch), redo the work over the RCCL all-gather."""
        try:
            return solver.build(n)
        except Exception as e:   # (EngineError in practice; anything else is treated the same way)
            if solver.exchange != "mailbox":
                raise
            if rank == 0:
                print("bench: peer mailbox exchange failed (%s); falling back to the all-gather" % e, file=sys.stderr)
            solver.fallback_to_collective()
            solver.engine.reset()
            return solver.build(n)

    # ---- warm-up, then time exactly K greedy iterations --------------------------------------
    if args.warmup > 0:
        build(args.warmup)
    if os.environ.get("BENCH_TEST_EXPIRE_MAILBOX") and solver.exchange == "mailbox":
        solver.engine.exchange_set_timeout(1e-7)      # tests: every later wait expires -> the fall-back below runs
    if not os.environ.get("BENCH_NO_EVENTS"):      # dev: what do the per-launch hipEvents cost?
        solver.engine.profile(PROFILE_EVERY)
    sync()
    t0 = time.perf_counter()
    tr = build(args.steps)
    sync()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    scan_ms, scan_launches = solver.engine.profile_read()
    solver.engine.profile(False)
    sel, err, status = tr
    steps_done = len(sel)

    if rank == 0:
        bytes_per_launch = float(solver.n_local) * args.dim * {nat.F64: 8, nat.F16: 2}.get(store, 4)
        avg_ms = scan_ms / max(scan_launches, 1)
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "scan_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                key = "%s_n%d_d%d_%s" % (args.alg, solver.n_local, args.dim, args.dtype)
                traffic = tj.get(key)
            except Exception:
                traffic = None
        out = {
            "metric": "greedy coreset iterations/sec (N=%d, d=%d)" % (args.rows, args.dim),
            "value": steps_done / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(steps_done, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": {nat.F32: "f32 scan + f64 state", nat.F16: "f16 rows, f32 scan + f64 state"}.get(store, "f64"),
            "data": "synthetic",
            "config": {
                "workload": "synthetic randn N=%d d=%d, %s, %d row shard(s), M=%d greedy iterations"
                            % (args.rows, args.dim, {"fw": "Frank-Wolfe", "giga": "GIGA", "omp": "OMP"}[args.alg],
                               world, args.steps),
                "alg": args.alg, "rows": args.rows, "dim": args.dim, "rows_per_gpu": solver.n_local,
                "exchange": solver.exchange if world > 1 else None,   # mailbox = device-side P2P stores; collective = all-gather
                "exact_rows_resident": not args.no_exact_rows,
                "steps_accepted": int((status == 0).sum()), "final_error": float(err[-1]) if len(err) else None,
                "rescue": solver.engine.stats(),   # fp64 re-scored candidates / exact-scan fallbacks since construction
            },
            "roofline": {
                "bound": "hbm", "kernel": "scan_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "avg_launch_ms": avg_ms, "launches": int(scan_launches), "timed_every": PROFILE_EVERY,
                "algorithmic_bytes_per_launch": bytes_per_launch,
            },
        }
        if not args.no_cpu_baseline and world == 1:   # reported once, on the single-GPU run
            out["cpu_baseline"] = cpu_baseline(args, args.alg, torch)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
