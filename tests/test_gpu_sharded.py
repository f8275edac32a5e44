"""GPU: the row-sharded driver with the REAL engine, two ranks sharing cuda:0 over gloo (RCCL
refuses two ranks on one device; the transport is the only thing substituted).  World size 2 must
reproduce world size 1 bit-for-bit: same b (chunk sums in global order), same selections, same
weights (SURVEY.md section 8e determinism requirement)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _data(N, d):
    return np.random.RandomState(21).randn(N, d)


def _worker(rank, world, port, alg, itrs, N, d, out_dir, exchange="collective", tag="", explicit_b=False, expect=None):
    os.environ["BCX_EXCHANGE"] = exchange
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesiancoresets_amd.sharded import ShardedSolver
    X = _data(N, d)
    s = ShardedSolver(alg, N, d, device=0)
    s.load_local(torch.from_numpy(X[s.row_begin:s.row_end]).cuda())
    torch.cuda.synchronize()
    assert s.finalize(2.0 * X.sum(axis=0) if explicit_b else None) == 0
    tr = s.build(itrs)
    idx, w = s.sparse_weights()
    b = s.engine.vector(0)
    assert s.exchange == (expect or (exchange if world > 1 else "collective")), s.exchange
    np.savez(os.path.join(out_dir, "%sw%d_r%d.npz" % (tag, world, rank)), sel=tr[0], err=tr[1], status=tr[2], idx=idx, w=w, b=b)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("alg,name", ((0, "giga"), (1, "fw"), (2, "omp")))
def test_two_shards_on_one_gpu_match_one_shard(tmp_path, alg, name):
    import torch.multiprocessing as mp
    from oracle.snnls_oracle import SnnlsOracle
    N, d, itrs = 9000, 40, 30
    for world in (1, 2, 3):
        mp.spawn(_worker, args=(world, _free_port(), alg, itrs, N, d, str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "w1_r0.npz")
    for world in (2, 3):
        for rank in range(world):
            r = np.load(tmp_path / ("w%d_r%d.npz" % (world, rank)))
            for k in ("sel", "err", "status", "idx", "w", "b"):
                assert np.array_equal(ref[k], r[k]), (world, rank, k)
    # and the whole thing matches the CPU oracle
    X = _data(N, d)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg=name)
    o.build(itrs)
    assert np.array_equal(ref["sel"], np.array([t[0] for t in o.trace]))
    w = np.zeros(N)
    w[ref["idx"]] = ref["w"]
    ow = o.weights()
    assert np.array_equal(np.flatnonzero(w > 0), np.flatnonzero(ow > 0))
    np.testing.assert_allclose(w[w > 0], ow[ow > 0], rtol=1e-5)


@pytest.mark.parametrize("alg,name", ((0, "giga"), (1, "fw"), (2, "omp")))
def test_peer_mailbox_exchange_matches_one_shard(tmp_path, alg, name):
    """The device-side record exchange (hipIpc-mapped mailboxes, csrc/resolve.hip mailbox_exchange): ranks
    sharing cuda:0 map each other's mailbox exactly as ranks on different GPUs would; world sizes 2 and 3
    must reproduce the single-shard run bit for bit, with the whole build enqueued at once."""
    import torch.multiprocessing as mp
    N, d, itrs = 9000, 40, 60
    mp.spawn(_worker, args=(1, _free_port(), alg, itrs, N, d, str(tmp_path)), nprocs=1, join=True)
    ref = np.load(tmp_path / "w1_r0.npz")
    for world in (2, 3):
        mp.spawn(_worker, args=(world, _free_port(), alg, itrs, N, d, str(tmp_path), "mailbox", "mb_"),
                 nprocs=world, join=True)
        for rank in range(world):
            r = np.load(tmp_path / ("mb_w%d_r%d.npz" % (world, rank)))
            for k in ("sel", "err", "status", "idx", "w", "b"):
                assert np.array_equal(ref[k], r[k]), (world, rank, k)


def test_bench_mailbox_preflight_in_helper_processes():
    """Multi-GPU runs try the device-side exchange in throw-away helper processes first (bench.py mailbox_preflight): a
    helper that dies -- as a GPU fault would kill it -- costs the helper, and every rank of the benchmark proper then takes
    the all-gather exchange and says why; a clean preflight leaves the mailbox mode on.  (Ranks share the GPU here.)"""
    args = ["--rows", "300000", "--dim", "64", "--steps", "10", "--warmup", "5", "--no-cpu-baseline", "--no-side-legs"]
    one, _ = _run_bench({}, 1, args)
    ok, _ = _run_bench({"BENCH_SHARE_GPU": "1", "BENCH_PREFLIGHT": "1"}, 2, args)
    assert ok["mailbox_preflight"] == {"ran": True, "ranks_failed": 0, "this_rank": "ok"}
    assert ok["config"]["exchange"] == "mailbox" and ok["config"]["final_error"] == one["config"]["final_error"]
    bad, err = _run_bench({"BENCH_SHARE_GPU": "1", "BENCH_PREFLIGHT": "1", "BENCH_PREFLIGHT_CRASH": "1", "BENCH_PREFLIGHT_TIMEOUT": "60"}, 2, args)
    assert bad["mailbox_preflight"]["ran"] and bad["mailbox_preflight"]["ranks_failed"] >= 1
    assert bad["config"]["exchange"] == "collective" and "preflight failed" in err
    assert bad["config"]["exchange_probe"]["reason"] == "BCX_EXCHANGE=collective"
    assert bad["config"]["final_error"] == one["config"]["final_error"]


@pytest.mark.parametrize("alg", (0, 1, 2))
def test_rows_beyond_the_mailbox_record_use_the_all_gather(tmp_path, alg):
    """Rows of 20000 values: a record no longer fits the LDS staging of the exchange kernels, the mailbox set-up says so and
    every rank takes the all-gather exchange; two shards equal one shard bit for bit (long scan with the query's tail in
    global memory, in-place column sums of the constructor pass, multi-kernel OMP step)."""
    import torch.multiprocessing as mp
    N, d, itrs = 2500, 20000, 10
    mp.spawn(_worker, args=(1, _free_port(), alg, itrs, N, d, str(tmp_path), "collective", "lr_"), nprocs=1, join=True)
    ref = np.load(tmp_path / "lr_w1_r0.npz")
    mp.spawn(_worker, args=(2, _free_port(), alg, itrs, N, d, str(tmp_path), "mailbox", "lr_", False, "collective"), nprocs=2, join=True)
    for rank in range(2):
        r = np.load(tmp_path / ("lr_w2_r%d.npz" % rank))
        for k in ("sel", "err", "status", "idx", "w", "b"):
            assert np.array_equal(ref[k], r[k]), (rank, k)


@pytest.mark.parametrize("exchange", ("collective", "mailbox"))
def test_explicit_b_frank_wolfe_shards_share_sigma(tmp_path, exchange):
    """SparseNNLS(A, b) with a caller-supplied b on row shards: Frank-Wolfe's sigma = sum of ALL row norms
    (frankwolfe.py:22,25) must come from every shard's chunk sums -- two and three shards equal one shard bit for
    bit, and the weights equal the oracle's (b = 2 * column sums, so the trace differs from the b = None cases)."""
    import torch.multiprocessing as mp
    from oracle.snnls_oracle import SnnlsOracle
    N, d, itrs = 9000, 40, 25
    mp.spawn(_worker, args=(1, _free_port(), 1, itrs, N, d, str(tmp_path), "collective", "xb_", True), nprocs=1, join=True)
    ref = np.load(tmp_path / "xb_w1_r0.npz")
    for world in (2, 3):
        mp.spawn(_worker, args=(world, _free_port(), 1, itrs, N, d, str(tmp_path), exchange, "xb_", True), nprocs=world, join=True)
        for rank in range(world):
            r = np.load(tmp_path / ("xb_w%d_r%d.npz" % (world, rank)))
            for k in ("sel", "err", "status", "idx", "w", "b"):
                assert np.array_equal(ref[k], r[k]), (world, rank, k)
    X = _data(N, d)
    o = SnnlsOracle(X.T, 2.0 * X.sum(axis=0), alg="fw")
    o.build(itrs)
    assert np.array_equal(ref["sel"], np.array([t[0] for t in o.trace]))
    w = np.zeros(N)
    w[ref["idx"]] = ref["w"]
    ow = o.weights()
    np.testing.assert_allclose(w[ow > 0], ow[ow > 0], rtol=1e-5)


def _mailbox_storm_worker(rank, world, port, out_dir):
    """tie-heavy rows: the candidate window overflows on a shard, every rank takes the exact-scan fallback"""
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BCX_EXCHANGE"] = "mailbox" if world > 1 else "collective"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesiancoresets_amd.sharded import ShardedSolver
    rs = np.random.RandomState(5)
    base = rs.randn(8, 17)
    X = base[rs.randint(0, 8, size=40000)] * (1.0 + rs.randint(0, 3, size=40000)[:, None])   # heavily duplicated directions
    s = ShardedSolver(0, X.shape[0], X.shape[1], device=0)
    s.load_local(torch.from_numpy(X[s.row_begin:s.row_end]).cuda())
    torch.cuda.synchronize()
    assert s.finalize(None) == 0
    tr = s.build(12)
    idx, w = s.sparse_weights()
    ex = s.engine.stats()["exact_fallbacks"]
    np.savez(os.path.join(out_dir, "storm_w%d_r%d.npz" % (world, rank)), sel=tr[0], err=tr[1], status=tr[2], idx=idx, w=w, ex=ex)
    dist.barrier()
    dist.destroy_process_group()


def test_peer_mailbox_exact_fallback(tmp_path):
    import torch.multiprocessing as mp
    for world in (1, 2):
        mp.spawn(_mailbox_storm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "storm_w1_r0.npz")
    for rank in range(2):
        r = np.load(tmp_path / ("storm_w2_r%d.npz" % rank))
        for k in ("sel", "err", "status", "idx", "w"):
            assert np.array_equal(ref[k], r[k]), (rank, k)
    assert int(np.load(tmp_path / "storm_w2_r0.npz")["ex"]) > 0   # the fallback really ran


def _svi_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bayesiancoresets_amd as bc
    from models import make_linreg_data, linreg_sampler
    g = np.load(os.path.join(ROOT, "tests", "golden", "svi_golden.npz"))
    N, D, S, sigsq = int(g["N"]), int(g["D"]), int(g["S"]), float(g["sigsq"])
    Z = make_linreg_data(1, N, D)
    per = (N + world - 1) // world
    lo, hi = rank * per, min(N, (rank + 1) * per)
    np.random.seed(2)
    grp = dist.group.WORLD
    prj = bc.DeviceProjector("linreg", linreg_sampler(np.zeros(D), np.eye(D), sigsq), S, sigsq=sigsq, group=grp,
                             row_offset=lo)
    alg = bc.SparseVICoreset(Z[lo:hi], prj, opt_itrs=int(g["opt_itrs"]), row_offset=lo, group=grp)
    alg.build(3)
    np.savez(os.path.join(out_dir, "svi_r%d.npz" % rank), idcs=alg.idcs, wts=alg.wts, pts=alg.pts)
    dist.barrier()
    dist.destroy_process_group()


def test_sparsevi_two_shards_match_reference(tmp_path):
    """Row-sharded SparseVI (config-5 style): column sums all-reduced, arg-max over ranks; two ranks on
    one GPU over gloo reproduce the reference's first three greedy steps."""
    import torch.multiprocessing as mp
    g = np.load(os.path.join(ROOT, "tests", "golden", "svi_golden.npz"))
    mp.spawn(_svi_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "svi_r0.npz"), np.load(tmp_path / "svi_r1.npz")
    for k in ("idcs", "wts", "pts"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.array_equal(r0["idcs"], g["step2_idcs"])
    np.testing.assert_allclose(r0["wts"], g["step2_wts"], rtol=1e-5, atol=1e-8)


def _svi_enqueued_worker(rank, world, port, out_dir, colsum):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bayesiancoresets_amd as bc
    from models import make_linreg_data
    N, D, S, sigsq = 30000, 9, 48, 1.0
    Z = make_linreg_data(4, N, D)
    per = (N + world - 1) // world
    lo, hi = rank * per, min(N, (rank + 1) * per)
    smp = bc.LinregPosteriorSampler(np.zeros(D), 2.0 * np.eye(D), sigsq, seed=9)      # (replicated: the same seed on every rank)
    grp = dist.group.WORLD if world > 1 else None
    prj = bc.DeviceProjector("linreg", smp, S, sigsq=sigsq, group=grp, row_offset=lo, colsum=colsum)
    alg = bc.SparseVICoreset(Z[lo:hi], prj, opt_itrs=20, row_offset=lo, group=grp) if world > 1 else bc.SparseVICoreset(Z, prj, opt_itrs=20)
    alg.build(3)
    assert alg._enqueue_plan() is not None
    np.savez(os.path.join(out_dir, "svie_w%d_r%d.npz" % (world, rank)), idcs=alg.idcs, wts=alg.wts, pts=alg.pts)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("colsum", ("mfma", "moments"))
def test_sparsevi_enqueued_loop_on_two_shards(tmp_path, colsum):
    """The device-resident weight optimisation (csrc/svi.hip) with the rows sharded over two ranks: every rank enqueues the same
    loop (replicated sampler and weights), the column sums are all-reduced inside it; same coreset as one rank."""
    import torch.multiprocessing as mp
    for world in (1, 2):
        mp.spawn(_svi_enqueued_worker, args=(world, _free_port(), str(tmp_path), colsum), nprocs=world, join=True)
    one = np.load(tmp_path / "svie_w1_r0.npz")
    r0, r1 = np.load(tmp_path / "svie_w2_r0.npz"), np.load(tmp_path / "svie_w2_r1.npz")
    for k in ("idcs", "wts", "pts"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.array_equal(one["idcs"], r0["idcs"]) and one["idcs"].shape[0] >= 2
    np.testing.assert_allclose(r0["wts"], one["wts"], rtol=1e-8, atol=1e-12)


# ---- bench.py contract, single process and under torchrun (two ranks sharing the GPU) -------------------
def _run_bench(extra_env, nproc, args):
    import json
    import subprocess
    env = dict(os.environ)
    env.update(extra_env)
    bench = os.path.join(ROOT, "bench.py")
    if nproc == 1:
        cmd = [sys.executable, bench] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), bench, "--gpus", str(nproc)] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]          # exactly ONE JSON line, from rank 0
    return json.loads(lines[0]), out.stderr


def test_bench_contract_one_and_two_ranks():
    args = ["--rows", "300000", "--dim", "64", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"]
    one, _ = _run_bench({}, 1, args)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in one, key
    assert one["n_gpus"] == 1 and one["steps"] == 40 and one["roofline"]["bound"] == "hbm" and "workload" in one["config"]
    two, _ = _run_bench({"BENCH_SHARE_GPU": "1"}, 2, args)
    assert two["n_gpus"] == 2 and two["config"]["exchange"] == "mailbox"
    assert two["config"]["final_error"] == one["config"]["final_error"]          # same bits for any shard count
    assert two["config"]["steps_accepted"] == one["config"]["steps_accepted"]
    # a probe that fails at construction: every rank uses the all-gather from the start
    three, err = _run_bench({"BENCH_SHARE_GPU": "1", "BCX_TEST_FAIL_PROBE": "1"}, 2, args)
    assert three["config"]["exchange"] == "collective" and three["config"]["final_error"] == one["config"]["final_error"]
    # an exchange that stops delivering mid-run: the build raises on every rank, bench.py redoes it over the all-gather
    four, err = _run_bench({"BENCH_SHARE_GPU": "1", "BENCH_TEST_EXPIRE_MAILBOX": "1", "BCX_EXCHANGE_TIMEOUT": "5"}, 2, args)
    assert four["config"]["exchange"] == "collective" and "falling back" in err
    assert four["config"]["final_error"] is not None


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` with NO launcher around it starts N ranks itself (torch.distributed.run on 127.0.0.1) and
    prints one line with n_gpus == N; without enough devices (and without the share-one-GPU test switch) it exits non-zero
    instead of printing a line for fewer ranks; a WORLD_SIZE that contradicts --gpus is an error too."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    args = ["--rows", "300000", "--dim", "64", "--steps", "10", "--warmup", "5", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    one, _ = _run_bench({}, 1, args)
    assert one["rccl_ranks"] == 1 and one["process_group"]["world_size"] == 1
    out = subprocess.run([sys.executable, bench, "--gpus", "2"] + args, capture_output=True, text=True, timeout=900,
                         env=dict(env, BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2 and two["config"]["exchange"] in ("mailbox", "collective")
    assert two["process_group"]["world_size"] == 2 and two["process_group"]["self_launched"]
    assert two["config"]["final_error"] == one["config"]["final_error"]
    assert "starting 2 ranks" in out.stderr
    if two["config"]["exchange"] == "mailbox":      # the other exchange mode is timed beside it
        assert two["collective_ms_per_step"] > 0 and two["mailbox_ms_per_step"] == two["ms_per_step"]
        assert two["config"]["collective_leg"]["final_error"] is not None
    if _device_count() < 8:
        out = subprocess.run([sys.executable, bench, "--gpus", "8"] + args, capture_output=True, text=True, timeout=300,
                             env={k: v for k, v in env.items() if k != "BENCH_SHARE_GPU"}, cwd=ROOT)
        assert out.returncode != 0 and "refusing" in out.stderr
        assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run([sys.executable, bench, "--gpus", "4"] + args, capture_output=True, text=True, timeout=300,
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), cwd=ROOT)
    assert out.returncode != 0 and "does not match" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("exchange", ("collective", "mailbox"))
def test_empty_shard(tmp_path, exchange):
    """N = 1500 rows are two 1024-row chunks: with three ranks the last shard owns no rows at all and still
    has to take part in every exchange (and must never win one)."""
    import torch.multiprocessing as mp
    N, d, itrs = 1500, 8, 12
    mp.spawn(_worker, args=(1, _free_port(), 1, itrs, N, d, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(_worker, args=(3, _free_port(), 1, itrs, N, d, str(tmp_path), exchange, "e_"), nprocs=3, join=True)
    ref = np.load(tmp_path / "w1_r0.npz")
    for rank in range(3):
        r = np.load(tmp_path / ("e_w3_r%d.npz" % rank))
        for k in ("sel", "err", "status", "idx", "w", "b"):
            assert np.array_equal(ref[k], r[k]), (rank, k)


def _hilbert_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bayesiancoresets_amd as bc

    class IDProjector(bc.Projector):
        def update(self, wts, pts):
            pass

        def project(self, pts, grad=False):
            return pts

    N, d = 7000, 32
    data = _data(N, d)
    if world == 1:
        cs = bc.HilbertCoreset(data, IDProjector(), snnls=bc.snnls.GIGA)
    else:
        lo, hi = bc.ShardedHilbertCoreset.local_rows(N)
        cs = bc.ShardedHilbertCoreset(data[lo:hi], IDProjector(), N, snnls=bc.snnls.GIGA)
        assert cs.snnls.exchange == "mailbox"
    cs.build(10)
    cs.build(15)
    w1, p1, i1 = cs.get()
    e1 = cs.error()
    cs.optimize()
    w2, p2, i2 = cs.get()
    e2 = cs.error()
    cs.reset()
    assert cs.size() == 0
    cs.build(5)
    w3, p3, i3 = cs.get()
    np.savez(os.path.join(out_dir, "hc_w%d_r%d.npz" % (world, rank)), w1=w1, p1=p1, i1=i1, e1=e1, w2=w2, p2=p2, i2=i2, e2=e2,
             w3=w3, i3=i3)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hilbert_coreset_equals_hilbert_coreset(tmp_path):
    """bc.ShardedHilbertCoreset on two ranks (mailbox exchange) == bc.HilbertCoreset on the whole data: build in
    two calls, get(), error(), optimize(), reset() -- same values on both ranks."""
    import torch.multiprocessing as mp
    for world in (1, 2):
        mp.spawn(_hilbert_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "hc_w1_r0.npz")
    for rank in range(2):
        r = np.load(tmp_path / ("hc_w2_r%d.npz" % rank))
        for k in ref.files:
            assert np.array_equal(ref[k], r[k]), (rank, k)
    assert float(ref["e2"]) <= float(ref["e1"])


def test_two_level_column_sums_are_shard_count_independent(tmp_path):
    """N = 300k rows = 293 chunk sums: b is formed by the two-level reduction (groups of 128 chunks by GLOBAL
    chunk index, csrc/ingest.hip) and must still be bit-identical for 1, 2 and 3 shards."""
    import torch.multiprocessing as mp
    N, d, itrs = 300000, 8, 6
    for world in (1, 2, 3):
        mp.spawn(_worker, args=(world, _free_port(), 1, itrs, N, d, str(tmp_path), "mailbox" if world > 1 else "collective",
                                "b2_"), nprocs=world, join=True)
    ref = np.load(tmp_path / "b2_w1_r0.npz")
    np.testing.assert_allclose(ref["b"], _data(N, d).sum(axis=0), rtol=1e-12, atol=1e-9)
    for world in (2, 3):
        for rank in range(world):
            r = np.load(tmp_path / ("b2_w%d_r%d.npz" % (world, rank)))
            for k in ("sel", "err", "status", "idx", "w", "b"):
                assert np.array_equal(ref[k], r[k]), (world, rank, k)


def _devproj_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bayesiancoresets_amd as bc
    from lr_workload import make_data
    N, D, S = 9000, 10, 128
    Z = make_data(5, N, D)
    samples = np.random.RandomState(6).randn(S, D)
    prj = bc.DeviceProjector("logistic", lambda n, w, p: samples, S)
    if world == 1:
        cs = bc.HilbertCoreset(Z, prj, snnls=bc.snnls.OrthoPursuit)
        assert cs.snnls._center_rows                          # raw log-likelihoods, centred by the ingest pass
    else:
        lo, hi = bc.ShardedHilbertCoreset.local_rows(N)
        cs = bc.ShardedHilbertCoreset(Z[lo:hi], prj, N, snnls=bc.snnls.OrthoPursuit)
    cs.build(30)
    wts, pts, idcs = cs.get()
    np.savez(os.path.join(out_dir, "dp_w%d_r%d.npz" % (world, rank)), wts=wts, pts=pts, idcs=idcs, err=cs.error())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hilbert_behind_device_projector(tmp_path):
    """HilbertCoreset behind a DeviceProjector takes the raw log-likelihoods and lets the solver's constructor pass centre
    them (projector.py:21 folded into the ingest); row-sharded on two ranks the same must come out, bit for bit."""
    import torch.multiprocessing as mp
    for world in (1, 2):
        mp.spawn(_devproj_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "dp_w1_r0.npz")
    assert len(ref["idcs"]) >= 20
    for rank in range(2):
        r = np.load(tmp_path / ("dp_w2_r%d.npz" % rank))
        for k in ("idcs", "wts", "pts", "err"):
            assert np.array_equal(ref[k], r[k]), (rank, k)


# ---- "identical for G in {1, 2, 4, 8}" -- world sizes 4 and 8, ranks sharing cuda:0 ---------------------------
@pytest.mark.parametrize("exchange", ("collective", "mailbox"))
@pytest.mark.parametrize("alg", (0, 1, 2))
def test_four_and_eight_shards_match_one_shard(tmp_path, alg, exchange):
    """Every rank of a 4- and an 8-way split (mailbox flags [2][G], G spinning lanes, 8 records per exchange; or the
    all-gather) records the trace, weights and b of the single-shard run, bit for bit.  N = 40000 = 40 chunks: every one of
    the 8 shards owns rows (the last one a ragged 4160)."""
    import torch.multiprocessing as mp
    N, d, itrs = 40000, 48, 50
    mp.spawn(_worker, args=(1, _free_port(), alg, itrs, N, d, str(tmp_path)), nprocs=1, join=True)
    ref = np.load(tmp_path / "w1_r0.npz")
    assert len(ref["sel"]) == itrs
    for world in (4, 8):
        mp.spawn(_worker, args=(world, _free_port(), alg, itrs, N, d, str(tmp_path), exchange, "g_"), nprocs=world, join=True)
        for rank in range(world):
            r = np.load(tmp_path / ("g_w%d_r%d.npz" % (world, rank)))
            for k in ("sel", "err", "status", "idx", "w", "b"):
                assert np.array_equal(ref[k], r[k]), (world, rank, k)


def test_peer_mailbox_exact_fallback_four_shards(tmp_path):
    """the exact-scan fallback (candidate window overflow on tie-heavy rows) taken by FOUR ranks together"""
    import torch.multiprocessing as mp
    for world in (1, 4):
        mp.spawn(_mailbox_storm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "storm_w1_r0.npz")
    for rank in range(4):
        r = np.load(tmp_path / ("storm_w4_r%d.npz" % rank))
        for k in ("sel", "err", "status", "idx", "w"):
            assert np.array_equal(ref[k], r[k]), (rank, k)
    assert int(np.load(tmp_path / "storm_w4_r0.npz")["ex"]) > 0


def _c4_worker(rank, world, port, itrs, out_dir):
    """configs[3] geometry (N = 10M, d = 512, Frank-Wolfe) with bench.py's own generator (seeded per 8192-row block, so
    the matrix does not depend on the shard count); `world` ranks share cuda:0."""
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import argparse
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BCX_EXCHANGE"] = "mailbox"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from bayesiancoresets_amd.sharded import ShardedSolver
    args = argparse.Namespace(rows=10_000_000, dim=512, seed=1)
    s = ShardedSolver(1, args.rows, args.dim, device=0)
    bench.load_synthetic(args, torch, s)
    assert s.finalize(None) == 0
    tr = s.build(itrs)
    idx, w = s.sparse_weights()
    xs = s.engine.exchange_stats() if world > 1 else {}
    assert world == 1 or (s.exchange == "mailbox" and xs["exchanges"] >= itrs)
    np.savez(os.path.join(out_dir, "c4_w%d_r%d.npz" % (world, rank)), sel=tr[0], err=tr[1], status=tr[2], idx=idx, w=w,
             b=s.engine.vector(0), n_local=s.n_local)
    dist.barrier()
    dist.destroy_process_group()


def test_config4_geometry_as_eight_shards(tmp_path):
    """The FULL configs[3] problem as the 8 row shards of the 8-GPU run (1,250,304 rows x 512 each, ~9 GB per rank, all on
    this one GPU): trace, weights and b equal the one-shard run of the same matrix bit for bit (the one-shard selects are
    checked against a torch-fp64 restatement over all rows in test_gpu_fullsize.py::test_config4_fw_10m_x_512)."""
    import torch.multiprocessing as mp
    itrs = 48
    mp.spawn(_c4_worker, args=(1, _free_port(), itrs, str(tmp_path)), nprocs=1, join=True)
    ref = np.load(tmp_path / "c4_w1_r0.npz")
    assert len(ref["sel"]) == itrs and (ref["status"] == 0).all()
    mp.spawn(_c4_worker, args=(8, _free_port(), itrs, str(tmp_path)), nprocs=8, join=True)
    rows = 0
    for rank in range(8):
        r = np.load(tmp_path / ("c4_w8_r%d.npz" % rank))
        rows += int(r["n_local"])
        for k in ("sel", "err", "status", "idx", "w", "b"):
            assert np.array_equal(ref[k], r[k]), (rank, k)
    assert rows == 10_000_000 and int(np.load(tmp_path / "c4_w8_r0.npz")["n_local"]) == 1_250_304
    # the selections come from all over the matrix, i.e. from every shard
    assert len(set(int(i) // 1_250_304 for i in ref["sel"])) == 8


def test_bench_contract_eight_ranks():
    """bench.py --gpus 8 under torchrun (ranks sharing the GPU): the SCALE contract end to end -- n_gpus, rows_per_gpu,
    exchange mode + probe, the device-stamped exchange wait -- and the same bits as one rank."""
    args = ["--rows", "400000", "--dim", "64", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"]
    one, _ = _run_bench({}, 1, args)
    eight, _ = _run_bench({"BENCH_SHARE_GPU": "1"}, 8, args)
    assert eight["n_gpus"] == 8 and eight["config"]["exchange"] == "mailbox"
    assert eight["config"]["rows_per_gpu"] == 50176                      # 49 chunks of 1024 rows on rank 0
    assert eight["config"]["final_error"] == one["config"]["final_error"]
    assert eight["config"]["steps_accepted"] == one["config"]["steps_accepted"]
    assert eight["config"]["exchanges_timed"] == 40
    assert 0.0 < eight["config"]["exchange_wait_us"] <= eight["config"]["exchange_wait_us_max"]
    assert eight["config"]["exchange_us"] >= eight["config"]["exchange_wait_us_best_rank"]
    assert eight["config"]["exchange_probe"]["result"] == 1
    # the same line carries the OTHER exchange mode on the same shards (one all-gather of the records per iteration) with its
    # exchange step timed, so that the first 8-GPU box yields mailbox and collective numbers in one run ...
    leg = eight["config"]["collective_leg"]
    assert leg["exchanges_timed"] >= 40 and 0.0 < leg["exchange_us"] <= leg["exchange_us_max"]
    assert leg["final_error"] == one["config"]["final_error"]
    # ... and a run that is FORCED onto the collective (what a failed mailbox probe does) reports the same fields
    forced, _ = _run_bench({"BENCH_SHARE_GPU": "1", "BCX_EXCHANGE": "collective"}, 8, args)
    assert forced["config"]["exchange"] == "collective" and forced["config"]["exchanges_timed"] == 40
    assert 0.0 < forced["config"]["exchange_us"] <= forced["config"]["exchange_us_max"]
    assert forced["config"]["final_error"] == one["config"]["final_error"]


def test_multi_rank_line_carries_cpu_baseline_and_both_rooflines():
    """`BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 10` with NO launcher: the line of a multi-rank run carries the CPU
    baseline (rank 0 times the oracle while the other rank waits on the rendezvous store), the per-GPU and the aggregate
    roofline, and the same workload as one shard with the scaling efficiency against it."""
    import json
    import subprocess
    bench = os.path.join(ROOT, "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "10", "--warmup", "5", "--rows", "300000", "--dim", "64",
                          "--cpu-seconds", "2"], capture_output=True, text=True, timeout=900, env=dict(env, BENCH_SHARE_GPU="1"), cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    two = json.loads(lines[0])
    assert two["n_gpus"] == 2
    cb = two["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["onepass_value"] > 0
    assert cb["sample_rows"] == 300000        # the whole (small) workload fits the sample: nothing scaled
    rf = two["roofline"]
    assert 0.0 < rf["frac"] and 0.0 < rf["frac_aggregate"] <= 1.0 and rf["peak_aggregate"] == 2 * rf["peak"]
    assert len(rf["per_gpu_frac"]) == 2 and sum(rf["per_gpu_rows"]) == 300000
    assert two["one_rank_its"] > 0 and two["one_rank_same_selections"] is True
    assert abs(two["scaling_efficiency"] - two["value"] / two["one_rank_its"] / 2) < 1e-12
    assert two["speedup_vs_cpu_baseline"] > 1.0


# ---- one rank per DEVICE over RCCL: runs only where the box has >= 2 GPUs (the 1-GPU test boxes skip it) ------------
def _multi_device_worker(rank, world, port, alg, itrs, N, d, out_dir, exchange):
    os.environ["BCX_EXCHANGE"] = exchange
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    from bayesiancoresets_amd.sharded import ShardedSolver
    X = _data(N, d)
    s = ShardedSolver(alg, N, d, device=rank)
    s.load_local(torch.from_numpy(X[s.row_begin:s.row_end]).cuda(rank))
    torch.cuda.synchronize()
    assert s.finalize(None) == 0
    tr = s.build(itrs)
    idx, w = s.sparse_weights()
    np.savez(os.path.join(out_dir, "md_%s_w%d_r%d.npz" % (exchange, world, rank)), sel=tr[0], err=tr[1], status=tr[2], idx=idx, w=w,
             b=s.engine.vector(0), exchange=np.array(s.exchange), probe=np.array(str(s.probe_info)))
    dist.barrier()
    dist.destroy_process_group()


def _device_count():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:
        return 0


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: one rank per device over RCCL / xGMI")
@pytest.mark.parametrize("exchange", ("collective", "mailbox"))
@pytest.mark.parametrize("alg", (0, 1, 2))
def test_one_rank_per_device_matches_one_shard(tmp_path, alg, exchange):
    """SURVEY.md section 8e on real peers: ranks on DIFFERENT GPUs (RCCL all-gather, or the hipIpc peer mailbox with
    system-scope stores over xGMI) reproduce the single-shard run bit for bit: (score desc, global index asc) winner,
    chunk sums reduced in global order.  The mailbox run also reports which mode the construction-time probe chose."""
    import torch.multiprocessing as mp
    N, d, itrs = 40000, 64, 80
    world = min(_device_count(), 4)
    mp.spawn(_worker, args=(1, _free_port(), alg, itrs, N, d, str(tmp_path)), nprocs=1, join=True)
    ref = np.load(tmp_path / "w1_r0.npz")
    mp.spawn(_multi_device_worker, args=(world, _free_port(), alg, itrs, N, d, str(tmp_path), exchange), nprocs=world, join=True)
    for rank in range(world):
        r = np.load(tmp_path / ("md_%s_w%d_r%d.npz" % (exchange, world, rank)))
        for k in ("sel", "err", "status", "idx", "w", "b"):
            assert np.array_equal(ref[k], r[k]), (rank, k)
        if exchange == "collective":
            assert str(r["exchange"]) == "collective"
        else:   # the probe decides; either outcome must give the same bits, and it must say why
            assert str(r["exchange"]) in ("mailbox", "collective") and "reason" in str(r["probe"])


# ---- subsampling on row shards (hilbert.py:13-22, sparsevi.py:32-35) with the real engine, ranks sharing cuda:0 ----------
def _subsample_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bayesiancoresets_amd as bc

    class Cubed(bc.Projector):                      # any row-wise map; zero rows stay zero
        def update(self, wts, pts):
            pass

        def project(self, pts, grad=False):
            return pts ** 3

    N, d, n_sub = 9000, 24, 5000
    data = _data(N, d)
    data[3::11] = 0.0
    np.random.seed(77)                              # the same draw on every rank and in the single-process run
    if world == 1:
        cs = bc.HilbertCoreset(data, Cubed(), n_subsample=n_sub, snnls=bc.snnls.FrankWolfe)
    else:
        lo, hi = bc.ShardedHilbertCoreset.local_rows(N)
        cs = bc.ShardedHilbertCoreset(data[lo:hi], Cubed(), N, n_subsample=n_sub, snnls=bc.snnls.FrankWolfe)
    cs.build(25)
    wts, pts, idcs = cs.get()
    np.savez(os.path.join(out_dir, "sub_w%d_r%d.npz" % (world, rank)), wts=wts, pts=pts, idcs=idcs, err=cs.error(), sub=cs.sub_idcs)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hilbert_subsample_equals_hilbert_subsample(tmp_path):
    import torch.multiprocessing as mp
    for world in (1, 2):
        mp.spawn(_subsample_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "sub_w1_r0.npz")
    assert len(ref["sub"]) > 2048 and len(ref["idcs"]) >= 20
    for rank in range(2):
        r = np.load(tmp_path / ("sub_w2_r%d.npz" % rank))
        for k in ("sub", "idcs", "wts", "pts", "err"):
            assert np.array_equal(ref[k], r[k]), (rank, k)


def _svi_subsample_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bayesiancoresets_amd as bc
    from models import make_linreg_data, linreg_sampler
    N, D, S, sigsq = 30000, 12, 48, 1.0
    Z = make_linreg_data(3, N, D)
    per = (N + world - 1) // world
    lo, hi = rank * per, min(N, (rank + 1) * per)
    np.random.seed(9)
    grp = dist.group.WORLD if world > 1 else None
    prj = bc.DeviceProjector("linreg", linreg_sampler(np.zeros(D), np.eye(D), sigsq), S, sigsq=sigsq, group=grp, row_offset=lo)
    if world > 1:
        alg = bc.SparseVICoreset(Z[lo:hi], prj, n_subsample_select=8000, n_subsample_opt=5000, opt_itrs=8, row_offset=lo, group=grp)
    else:
        alg = bc.SparseVICoreset(Z, prj, n_subsample_select=8000, n_subsample_opt=5000, opt_itrs=8)
    alg.build(3)
    np.savez(os.path.join(out_dir, "svis_w%d_r%d.npz" % (world, rank)), idcs=alg.idcs, wts=alg.wts, pts=alg.pts)
    dist.barrier()
    dist.destroy_process_group()


def test_sparsevi_subsampling_on_row_shards(tmp_path):
    """n_subsample_select / n_subsample_opt (sparsevi.py:32-35) with rows sharded over two ranks: the replicated draw
    picks the same points as the single-process run; weights agree to the rounding of the all-reduced column sums."""
    import torch.multiprocessing as mp
    for world in (1, 2):
        mp.spawn(_svi_subsample_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = np.load(tmp_path / "svis_w1_r0.npz")
    assert len(ref["idcs"]) >= 2
    r0, r1 = np.load(tmp_path / "svis_w2_r0.npz"), np.load(tmp_path / "svis_w2_r1.npz")
    for k in ("idcs", "wts", "pts"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.array_equal(ref["idcs"], r0["idcs"]) and np.array_equal(ref["pts"], r0["pts"])
    np.testing.assert_allclose(r0["wts"], ref["wts"], rtol=1e-6, atol=1e-10)
