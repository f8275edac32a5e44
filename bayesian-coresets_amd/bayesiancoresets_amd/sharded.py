"""Row-sharded solver: one process per GPU, contiguous row blocks, one small all-gather per
greedy iteration.

Every rank scans only its own rows and produces a record {exact score, global index, norm,
flags, raw row} (d + 4 doubles).  The records are all-gathered (RCCL over xGMI through
``torch.distributed``; payload world_size * (d + 4) * 8 bytes, i.e. latency-bound) and every
rank applies the *same* fp64 reweight to its replicated O(d) state, so xw, weights and the
trace stay bit-identical on all ranks; no residual all-reduce is needed.  Contiguous shards
plus the (score desc, global index asc) winner rule keep NumPy's lowest-index tie-break.

The driver is written against a small engine protocol (``bayesiancoresets_amd._native.Engine``
implements it on the GPU); tests inject a CPU stand-in to exercise this orchestration under
``gloo`` without a GPU.
"""
import os

import numpy as np

from . import _native as nat

CHUNK_ROWS = nat.CHUNK_ROWS


def shard_bounds(n_global, world_size, align=CHUNK_ROWS):
    """Contiguous row ranges, every boundary a multiple of ``align`` rows (the engine's column-sum
    chunk) so that the chunk partition -- hence b and sum(norms) -- is identical for any world size."""
    n_chunks = (n_global + align - 1) // align
    per = (n_chunks + world_size - 1) // world_size
    bounds = []
    for r in range(world_size):
        lo = min(n_global, r * per * align)
        hi = min(n_global, (r + 1) * per * align)
        bounds.append((lo, hi))
    return bounds, per


class ShardedSolver(object):
    """Collective object: construct / call on every rank of ``group`` with the same arguments."""

    def __init__(self, alg, n_global, d, group=None, device=None, engine_factory=None, **engine_kw):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.alg, self.n_global, self.d = alg, int(n_global), int(d)
        bounds, self.chunks_per_rank = shard_bounds(self.n_global, self.world)
        self.row_begin, self.row_end = bounds[self.rank]
        self.n_local = self.row_end - self.row_begin
        self.bounds = bounds
        factory = engine_factory or nat.Engine
        if device is None:
            device = torch.cuda.current_device() if torch.cuda.is_available() else 0
        self.engine = factory(alg, self.n_local, self.d, n_global=self.n_global,
                              row_offset=self.row_begin if self.n_local > 0 else 0,
                              rank=self.rank, world_size=self.world, device=device, **engine_kw)
        self.tdev = self.engine.tensor_device() if hasattr(self.engine, "tensor_device") else torch.device("cuda", device)
        rec = self.d + nat.REC_HDR
        self.send = torch.zeros(rec, dtype=torch.float64, device=self.tdev)
        self.recv = torch.zeros(self.world * rec, dtype=torch.float64, device=self.tdev)
        self.reached_numeric_limit = False
        self.last_trace = None
        self._flat_gather = True

    def _all_gather(self, recv, send):
        """all_gather_into_tensor (one RCCL call); backends without it (gloo on device tensors) get the
        list form over views of the same receive buffer."""
        if self._flat_gather:
            try:
                self.dist.all_gather_into_tensor(recv, send, group=self.group)
                return
            except (RuntimeError, NotImplementedError):
                self._flat_gather = False
        views = list(recv.view(self.world, -1).unbind(0))
        self.dist.all_gather(views, send, group=self.group)

    # ---- construction ------------------------------------------------------
    def load_local(self, rows, local_row_begin=0):
        """rows: torch tensor (device or cpu) or ndarray holding local rows
        [local_row_begin, local_row_begin + len(rows))."""
        self.engine.load_rows_any(rows, local_row_begin)

    def finalize(self, b=None):
        torch, dist = self.torch, self.dist
        gathered_ptr, n_gathered, keep = None, 0, None
        if b is None and self.world > 1:
            per = self.chunks_per_rank
            mine = torch.zeros(per * (self.d + 1), dtype=torch.float64, device=self.tdev)
            self.engine.export_chunk_sums_tensor(mine, per)
            allc = torch.zeros(self.world * per * (self.d + 1), dtype=torch.float64, device=self.tdev)
            self._all_gather(allc, mine)
            keep = allc
            gathered_ptr = allc
            n_gathered = (self.n_global + CHUNK_ROWS - 1) // CHUNK_ROWS
        rc = self.engine.finalize_any(b, gathered_ptr, n_gathered)
        if self.world > 1:
            # a zero row / zero b on any shard fails the constructor everywhere
            flag = torch.tensor([float(-rc)], dtype=torch.float64, device=self.tdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            worst = -int(flag.item())
            if rc == nat.OK and worst != nat.OK:
                rc = worst
        del keep
        return rc

    # ---- the hot loop --------------------------------------------------------
    def _one_iteration(self, exact=False):
        self.engine.step_scan_tensor(self.send, exact)
        if self.world > 1:
            self._all_gather(self.recv, self.send)
            self.engine.step_apply_tensor(self.recv)
        else:
            self.engine.step_apply_tensor(self.send)

    def build(self, itrs, tol=1e-12):
        itrs = int(itrs)
        if self.engine.build_begin(itrs, tol):
            return None
        remaining = itrs
        while True:
            if self.world == 1 and hasattr(self.engine, "enqueue") and not os.environ.get("BCX_SHARDED_GENERIC"):
                self.engine.enqueue(remaining)       # single shard: scan + merged resolve/apply launches
            else:
                for _ in range(remaining):
                    self._one_iteration()
            done, need_exact, limit = self.engine.poll()   # replicated state: same answer on every rank
            if need_exact:
                self._one_iteration(exact=True)
                done, need_exact, limit = self.engine.poll()
                if need_exact:
                    raise nat.EngineError(nat.ERR_STATE, "exact scan did not resolve the iteration")
            if limit or done >= itrs:
                break
            remaining = itrs - done
        self.reached_numeric_limit = bool(limit)
        self.last_trace = self.engine.trace(itrs)
        return self.last_trace

    # ---- read-out --------------------------------------------------------------
    def sparse_weights(self):
        return self.engine.sparse_weights()

    def error(self):
        return self.engine.error()

    def size(self):
        idx, w = self.engine.sparse_weights()
        return int((w > 0).sum())
