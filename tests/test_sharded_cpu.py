"""world_size-2 gloo test (CPU, no GPU): the row-sharded orchestration in
bayesiancoresets_amd/sharded.py -- shard bounds on chunk boundaries, chunk-sum all-gather in global
chunk order, one all-gather of (d+4)-double records per greedy iteration, replicated apply --
reproduces the single-process oracle.  The GPU engine is replaced by tests/fake_engine.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, alg, itrs, N, d, out_dir, explicit_b=False):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_engine import FakeEngine
    from bayesiancoresets_amd.sharded import ShardedSolver
    X = np.random.RandomState(11).randn(N, d)
    FakeEngine.FULL = X
    s = ShardedSolver(alg, N, d, engine_factory=FakeEngine)
    s.load_local(X[s.row_begin:s.row_end])
    rc = s.finalize(X.sum(axis=0) if explicit_b else None)
    assert rc == 0
    assert s.engine.saw_gathered        # every shard's chunk sums reach every rank, with or without a caller-supplied b
    tr = s.build(itrs)
    idx, w = s.sparse_weights()
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), sel=tr[0], err=tr[1], idx=idx, w=w, b=s.engine.b,
             bounds=np.array(s.bounds))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("alg,name,explicit_b", ((0, "giga", False), (1, "fw", False), (2, "omp", False), (1, "fw", True)))
def test_two_shards_match_single_process(tmp_path, alg, name, explicit_b):
    from oracle.snnls_oracle import SnnlsOracle
    N, d, itrs, world = 5000, 24, 15, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, alg, itrs, N, d, str(tmp_path), explicit_b), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    # replicated state is identical on both ranks
    for k in ("sel", "err", "idx", "w", "b"):
        assert np.array_equal(r0[k], r1[k]), k
    # shard boundaries are multiples of the 1024-row chunk and cover [0, N)
    bounds = r0["bounds"]
    assert bounds[0][0] == 0 and bounds[-1][1] == N and all(b[0] % 1024 == 0 for b in bounds)
    X = np.random.RandomState(11).randn(N, d)
    np.testing.assert_allclose(r0["b"], X.sum(axis=0), rtol=1e-12, atol=1e-12)
    o = SnnlsOracle(X.T, r0["b"], alg=name, mode="onepass")
    o.build(itrs)
    assert np.array_equal(r0["sel"], np.array([t[0] for t in o.trace]))
    ow = o.weights()
    assert np.array_equal(np.sort(r0["idx"]), np.flatnonzero(ow != 0))
    w = np.zeros(N)
    w[r0["idx"]] = r0["w"]
    np.testing.assert_allclose(w, ow, rtol=1e-12, atol=0)


@pytest.mark.parametrize("alg", (0, 1, 2))
def test_four_shards_match_two_shards(tmp_path, alg):
    """world size 4 under gloo: the replicated state is the same on all four ranks and equals the 2-rank run bit for bit"""
    N, d, itrs = 9000, 16, 12
    mp.spawn(_worker, args=(2, _free_port(), alg, itrs, N, d, str(tmp_path), False), nprocs=2, join=True)
    two = {k: np.load(tmp_path / "r0.npz")[k] for k in ("sel", "err", "idx", "w", "b")}
    mp.spawn(_worker, args=(4, _free_port(), alg, itrs, N, d, str(tmp_path), False), nprocs=4, join=True)
    for rank in range(4):
        r = np.load(tmp_path / ("r%d.npz" % rank))
        for k in ("sel", "err", "idx", "w", "b"):
            assert np.array_equal(two[k], r[k]), (rank, k)


def test_shard_bounds_properties():
    sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
    from bayesiancoresets_amd.sharded import shard_bounds
    for n in (0, 1, 1023, 1024, 1025, 10_000_000, 5_000_001):
        for world in (1, 2, 3, 4, 8):
            bounds, per = shard_bounds(n, world)
            assert len(bounds) == world
            assert bounds[0][0] == 0 and bounds[-1][1] == n
            for (lo, hi), (lo2, hi2) in zip(bounds, bounds[1:]):
                assert hi == lo2 and lo <= hi
            assert all(lo % 1024 == 0 for lo, hi in bounds if hi > lo)
            # every rank but the last non-empty one holds exactly `per` chunks -> gathered chunk sums are
            # already in global order with padding only at the end
            sizes = [hi - lo for lo, hi in bounds]
            nonempty = [s for s in sizes if s > 0]
            assert all(s == per * 1024 for s in nonempty[:-1])


def _hilbert_worker(rank, world, port, N, d, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_engine import FakeEngine
    import bayesiancoresets_amd as bc

    class Scaled(bc.Projector):           # any row-wise map will do: vecs = 2 * data
        def update(self, wts, pts):
            pass

        def project(self, pts, grad=False):
            return 2.0 * pts

    data = np.random.RandomState(17).randn(N, d)
    FakeEngine.FULL = 2.0 * data
    lo, hi = bc.ShardedHilbertCoreset.local_rows(N)
    cs = bc.ShardedHilbertCoreset(data[lo:hi], Scaled(), N, snnls=bc.snnls.FrankWolfe, engine_factory=FakeEngine)
    assert cs.get()[0].shape == (0,)                       # nothing built yet: empty triple (coreset.py:25-28)
    cs.build(7)
    cs.build(8)                                            # incremental builds continue (coreset.py:33-38)
    wts, pts, idcs = cs.get()
    err = cs.error()
    with pytest.raises(ValueError):
        bc.ShardedHilbertCoreset(data[lo:hi - 1] if hi > lo else data[:1], Scaled(), N, snnls=bc.snnls.FrankWolfe,
                                 engine_factory=FakeEngine)
    np.savez(os.path.join(out_dir, "h%d.npz" % rank), wts=wts, pts=pts, idcs=idcs, err=err, size=cs.size())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hilbert_coreset_matches_single_process(tmp_path):
    """ShardedHilbertCoreset on two ranks == the reference HilbertCoreset semantics on the whole data
    (index-sorted wts / idcs, pts = data rows gathered from their owners)."""
    from oracle.snnls_oracle import SnnlsOracle, hilbert_readout
    N, d, world = 3000, 12, 2
    mp.spawn(_hilbert_worker, args=(world, _free_port(), N, d, str(tmp_path)), nprocs=world, join=True)
    h0, h1 = np.load(tmp_path / "h0.npz"), np.load(tmp_path / "h1.npz")
    for k in ("wts", "pts", "idcs", "err", "size"):
        assert np.array_equal(h0[k], h1[k]), k
    data = np.random.RandomState(17).randn(N, d)
    vecs = 2.0 * data
    o = SnnlsOracle(vecs.T, vecs.sum(axis=0), alg="fw", mode="onepass")
    o.build(15)
    wts, idcs = hilbert_readout(o.weights())
    assert np.array_equal(h0["idcs"], idcs)
    np.testing.assert_allclose(h0["wts"], wts, rtol=1e-12)
    assert np.array_equal(h0["pts"], data[idcs])
    np.testing.assert_allclose(float(h0["err"]), o.error(), rtol=1e-12)
    assert int(h0["size"]) == len(idcs)


def _subsample_worker(rank, world, port, N, d, n_sub, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_engine import FakeEngine
    import bayesiancoresets_amd as bc

    class Scaled(bc.Projector):
        def update(self, wts, pts):
            pass

        def project(self, pts, grad=False):
            return 2.0 * pts

    data = np.random.RandomState(29).randn(N, d)
    data[5::7] = 0.0                                       # zero vectors: dropped in the subsample branch (hilbert.py:19-22)
    np.random.seed(41)
    drawn = np.unique(np.random.randint(N, size=n_sub))
    kept = drawn[np.abs(data[drawn]).sum(axis=1) > 0]
    FakeEngine.FULL = 2.0 * data[kept]                     # what the replicated reweight of the stand-in engine reads
    np.random.seed(41)                                     # every rank seeds alike: the same draw everywhere
    lo, hi = bc.ShardedHilbertCoreset.local_rows(N)
    cs = bc.ShardedHilbertCoreset(data[lo:hi], Scaled(), N, n_subsample=n_sub, snnls=bc.snnls.GIGA, engine_factory=FakeEngine)
    assert np.array_equal(cs.sub_idcs, kept)
    cs.build(12)
    wts, pts, idcs = cs.get()
    np.savez(os.path.join(out_dir, "ss%d.npz" % rank), wts=wts, pts=pts, idcs=idcs, err=cs.error(), kept=kept)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_hilbert_subsample_matches_single_process(tmp_path):
    """n_subsample on row shards (hilbert.py:13-22): same draw on every rank, zero vectors dropped, surviving
    vectors dealt to aligned solver shards; result == the oracle on the subsampled matrix, idcs are DATA rows."""
    from oracle.snnls_oracle import SnnlsOracle, hilbert_readout
    N, d, n_sub, world = 6000, 10, 2500, 2
    mp.spawn(_subsample_worker, args=(world, _free_port(), N, d, n_sub, str(tmp_path)), nprocs=world, join=True)
    h0, h1 = np.load(tmp_path / "ss0.npz"), np.load(tmp_path / "ss1.npz")
    for k in ("wts", "pts", "idcs", "err"):
        assert np.array_equal(h0[k], h1[k]), k
    data = np.random.RandomState(29).randn(N, d)
    data[5::7] = 0.0
    kept = h0["kept"]
    assert len(kept) > 1024                                 # the subsample really spans two solver shards
    vecs = 2.0 * data[kept]
    o = SnnlsOracle(vecs.T, vecs.sum(axis=0), alg="giga", mode="onepass")
    o.build(12)
    wts, sub_rows = hilbert_readout(o.weights())
    assert np.array_equal(h0["idcs"], kept[sub_rows])
    np.testing.assert_allclose(h0["wts"], wts, rtol=1e-12)
    assert np.array_equal(h0["pts"], data[kept[sub_rows]])


def _mailbox_worker(rank, world, port, fail_probe, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.pop("BCX_EXCHANGE", None)
    if fail_probe and rank == 1:
        os.environ["BCX_TEST_FAIL_PROBE"] = "1"      # ONE rank sees a bad probe: every rank must fall back
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_engine import FakeEngine, FakeMailboxEngine
    from bayesiancoresets_amd.sharded import ShardedSolver
    N, d = 4000, 16
    X = np.random.RandomState(23).randn(N, d)
    FakeEngine.FULL = X
    s = ShardedSolver(1, N, d, engine_factory=FakeMailboxEngine)
    assert s.exchange == ("collective" if fail_probe else "mailbox"), s.exchange
    assert getattr(s.engine, "attached", False) == (not fail_probe)
    s.load_local(X[s.row_begin:s.row_end])
    assert s.finalize(None) == 0
    tr = s.build(12)
    np.savez(os.path.join(out_dir, "mb%d_r%d.npz" % (int(fail_probe), rank)), sel=tr[0], err=tr[1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_probe", (False, True))
def test_mailbox_setup_and_fallback_host_logic(tmp_path, fail_probe):
    """ShardedSolver._setup_mailbox: handles gathered in rank order, probe, all-or-nothing agreement; a probe that
    fails on one rank drops every rank to the all-gather exchange.  Either way the trace is the oracle's."""
    from oracle.snnls_oracle import SnnlsOracle
    world = 2
    mp.spawn(_mailbox_worker, args=(world, _free_port(), fail_probe, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / ("mb%d_r0.npz" % int(fail_probe)))
    r1 = np.load(tmp_path / ("mb%d_r1.npz" % int(fail_probe)))
    assert np.array_equal(r0["sel"], r1["sel"]) and np.array_equal(r0["err"], r1["err"])
    X = np.random.RandomState(23).randn(4000, 16)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg="fw", mode="onepass")
    o.build(12)
    assert np.array_equal(r0["sel"], np.array([t[0] for t in o.trace]))


def _flaky_worker(rank, world, port, where, out_dir):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["BCX_EXCHANGE"] = "collective"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fake_engine import FakeEngine
    from bayesiancoresets_amd import _native as nat
    from bayesiancoresets_amd.sharded import ShardedSolver

    class FlakyEngine(FakeEngine):
        """fails ONCE, on rank 1 only, in the 6th iteration of the first build"""
        armed = True

        def _maybe_fail(self, phase):
            if self.rank == 1 and self.armed and self.done == 5 and phase == where:
                FlakyEngine.armed = False
                raise nat.EngineError(nat.ERR_HIP, "injected failure in %s" % phase)

        def step_scan_tensor(self, send, exact=False):
            self._maybe_fail("scan")
            return FakeEngine.step_scan_tensor(self, send, exact)

        def step_apply_tensor(self, recv):
            self._maybe_fail("apply")
            return FakeEngine.step_apply_tensor(self, recv)

        def poll(self):
            self._maybe_fail("poll")
            return FakeEngine.poll(self)

    N, d = 4000, 16
    X = np.random.RandomState(23).randn(N, d)
    FakeEngine.FULL = X
    s = ShardedSolver(1, N, d, engine_factory=FlakyEngine)
    s.load_local(X[s.row_begin:s.row_end])
    assert s.finalize(None) == 0
    import time
    t0 = time.time()
    msg = None
    try:
        s.build(5 if where == "poll" else 12)
    except nat.EngineError as e:
        msg = str(e)
    took = time.time() - t0
    # every rank raised -- the failing one its own error, its peer one that names it -- and nobody hung in a collective
    assert msg is not None and took < 60.0, (msg, took)
    assert ("injected failure" in msg) if rank == 1 else ("rank 1" in msg and "injected failure" in msg), msg
    # the solver survives: after a reset on every rank the next build is the oracle's again
    s.engine.reset()
    tr = s.build(12)
    np.savez(os.path.join(out_dir, "fl_%s_r%d.npz" % (where, rank)), sel=tr[0], err=tr[1])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("where", ("scan", "apply", "poll"))
def test_engine_error_on_one_rank_stops_every_rank_and_the_solver_survives(tmp_path, where):
    """An EngineError in the middle of a build on ONE rank (in the scan, in the apply, at the poll): the rank keeps its peers'
    all-gathers matched, every rank settles the failure together (ShardedSolver._settle) and raises; a reset + rebuild
    then reproduces the oracle on both ranks."""
    from oracle.snnls_oracle import SnnlsOracle
    world = 2
    mp.spawn(_flaky_worker, args=(world, _free_port(), where, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / ("fl_%s_r0.npz" % where)), np.load(tmp_path / ("fl_%s_r1.npz" % where))
    assert np.array_equal(r0["sel"], r1["sel"]) and np.array_equal(r0["err"], r1["err"])
    X = np.random.RandomState(23).randn(4000, 16)
    o = SnnlsOracle(X.T, X.sum(axis=0), alg="fw", mode="onepass")
    o.build(12)
    assert np.array_equal(r0["sel"], np.array([t[0] for t in o.trace]))
