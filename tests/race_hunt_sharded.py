"""test infrastructure (by hand): N ranks sharing cuda:0, peer-mailbox exchange, the same build repeated many
times -- every repetition must give the same trace (and all ranks the same one: ShardedSolver checks that)."""
import os, sys, socket, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, alg, N, d, itrs, reps):
    for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd")):
        sys.path.insert(0, p)
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ["BCX_EXCHANGE"] = "mailbox"
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bayesiancoresets_amd.sharded import ShardedSolver
    X = np.random.RandomState(7).randn(N, d)
    s = ShardedSolver(alg, N, d, device=0)
    assert s.exchange == "mailbox"
    s.load_local(torch.from_numpy(X[s.row_begin:s.row_end]).cuda()); torch.cuda.synchronize()
    assert s.finalize(None) == 0
    seen = {}
    for r in range(reps):
        s.engine.reset()
        tr = s.build(itrs)
        idx, w = s.sparse_weights()
        h = hashlib.md5(tr[0].tobytes() + tr[1].tobytes() + w.tobytes()).hexdigest()
        seen.setdefault(h, []).append(r)
    if rank == 0:
        print("alg %d world %d N=%d d=%d itrs=%d x%d: %d distinct outcome(s) %s" % (alg, world, N, d, itrs, reps, len(seen),
              "" if len(seen) == 1 else [v[:5] for v in seen.values()]), flush=True)
    dist.barrier(); dist.destroy_process_group()


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


if __name__ == "__main__":
    import torch.multiprocessing as mp
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for alg, world in ((1, 2), (0, 3), (2, 2)):
        mp.spawn(worker, args=(world, free_port(), alg, 60000, 64, 60, reps), nprocs=world, join=True)
