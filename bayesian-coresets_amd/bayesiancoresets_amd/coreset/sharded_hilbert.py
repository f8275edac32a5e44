"""Hilbert coreset over row shards: one process per GPU, every rank holds a contiguous block of the data.

Same construction and the same methods as ``HilbertCoreset`` (reference: bayesiancoresets/coreset/
hilbert.py:7-48) -- project, b = column sums, greedy sparse NNLS, ``wts / idcs / pts`` in index order --
but the N x d vectors never exist in one place: each rank projects and ingests only its own rows and the
solver state is replicated (``bayesiancoresets_amd/sharded.py``), so ``build / get / error / optimize /
size / reset`` return the same values on every rank, equal to what the single-process class gives on the
whole data.  COLLECTIVE: construct and call on every rank of ``group``.

    lo, hi = ShardedHilbertCoreset.local_rows(n_global)            # this rank's rows (1024-row aligned)
    cs = ShardedHilbertCoreset(data[lo:hi], projector, n_global, snnls=bc.snnls.GIGA)
    cs.build(1000); wts, pts, idcs = cs.get()                       # identical on all ranks

``idcs`` are global row numbers; ``pts`` are the data rows of the selected points, supplied by the ranks
that own them.

``n_subsample`` (hilbert.py:13-22): every rank draws the SAME ``unique(randint(n_global, size=n_subsample))`` -- seed
NumPy identically on all ranks, as for any replicated random decision -- projects the drawn rows it owns, drops
zero vectors, and the surviving vectors are dealt to 1024-row aligned solver shards (one padded all-gather of
n_subsample x d values; a subsample is small by intent).  ``sub_idcs`` is replicated.
"""
import numpy as np

from .. import util
from .. import _native as nat
from ..sharded import ShardedSolver, shard_bounds
from ..snnls.giga import GIGA
from ..snnls.snnls import DeviceSparseNNLS as _DeviceSolver, warn_failed_steps
from ..util.errors import NumericalPrecisionError
from .coreset import Coreset


class ShardedHilbertCoreset(Coreset):
    @staticmethod
    def local_rows(n_global, group=None):
        """(first, one-past-last) global row of the calling rank's shard."""
        import torch.distributed as dist
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        return shard_bounds(int(n_global), world)[0][rank]

    def __init__(self, local_data, ll_projector, n_global, n_subsample=None, snnls=GIGA, group=None,
                 engine_factory=None, **kw):
        if not (isinstance(snnls, type) and issubclass(snnls, _DeviceSolver) and snnls._ALG is not None):
            raise ValueError("ShardedHilbertCoreset needs a device solver class (GIGA, FrankWolfe, OrthoPursuit)")
        import torch.distributed as dist
        n_global = int(n_global)
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(n_global, world)[0][rank]
        if int(local_data.shape[0]) != hi - lo:
            raise ValueError("this rank must hold global rows [%d, %d) (ShardedHilbertCoreset.local_rows); got %d rows"
                             % (lo, hi, int(local_data.shape[0])))
        self.data, self.group = local_data, group
        self.row_begin, self.row_end = lo, hi
        # (an engine_factory written for the two-argument load_rows_any gets the centred vectors of project())
        from .hilbert import HilbertCoreset as _H
        load = getattr(engine_factory, "load_rows_any", None) if engine_factory is not None else None
        fold = (n_subsample is None and hasattr(ll_projector, "project_uncentred")
                and (load is None or _H._accepts(load, "center")))
        if n_subsample is None:
            # a device projector: raw log-likelihoods, centred by the solver's constructor pass (one pass over N x S less)
            vecs = ll_projector.project_uncentred(local_data) if fold else ll_projector.project(local_data)
            n_rows, self.sub_idcs = n_global, None
        else:
            drawn = np.unique(np.random.randint(n_global, size=n_subsample))      # hilbert.py:16, same on every rank
            mine = drawn[(drawn >= lo) & (drawn < hi)]
            vecs, self.sub_idcs = self._deal_subsample(ll_projector.project(local_data[mine - lo]), mine, world, rank)
            n_rows = int(self.sub_idcs.shape[0])
        d = int(vecs.shape[1])
        if n_rows == 0:
            raise ValueError("ShardedHilbertCoreset.__init__(): every drawn vector is zero -- nothing to build a coreset from")
        self.snnls = ShardedSolver(snnls._ALG, n_rows, d, group=group, engine_factory=engine_factory)
        if int(vecs.shape[0]) != self.snnls.n_local:
            raise ValueError("ShardedHilbertCoreset.__init__(): this rank holds %d projected rows, its solver shard %d"
                             % (int(vecs.shape[0]), self.snnls.n_local))
        if self.snnls.n_local:
            # an engine built by a plain factory FUNCTION cannot be inspected before it exists: one written for the
            # two-argument load_rows_any gets what project() would have given it (row means subtracted here)
            if fold and not _H._accepts(self.snnls.engine.load_rows_any, "center"):
                vecs = vecs - vecs.mean(axis=1)[:, None] if isinstance(vecs, np.ndarray) else vecs - vecs.mean(dim=1, keepdim=True)
                fold = False
            self.snnls.load_local(vecs, center=fold)
        rc = self.snnls.finalize(None)                     # b = column sums over all shards (hilbert.py:24)
        if rc == nat.ERR_ZERO_ROW:
            raise ValueError("ShardedHilbertCoreset.__init__(): A must not have any 0 columns")   # giga.py:11-12
        if rc == nat.ERR_ZERO_B:
            raise NumericalPrecisionError("norm of b must be > 0")                                 # giga.py:16-17
        if rc != nat.OK:
            raise nat.EngineError(rc, "finalize failed")
        super().__init__(**kw)

    def _deal_subsample(self, vecs, mine, world, rank):
        """Drop zero vectors (hilbert.py:19-22), then move the surviving projected rows from the ranks that OWN the
        data rows to the ranks that hold the matching 1024-row aligned solver shards.  Returns (this rank's solver
        rows, the global data indices of all surviving rows in solver-row order)."""
        import torch
        import torch.distributed as dist
        is_t = isinstance(vecs, torch.Tensor)
        v = vecs if is_t else torch.from_numpy(np.ascontiguousarray(vecs, dtype=np.float64))
        if world > 1 and v.device.type == "cpu" and dist.get_backend(self.group) == "nccl":
            v = v.to(torch.device("cuda", torch.cuda.current_device()))     # RCCL moves device tensors only
        if v.is_cuda:
            keep = nat.device_row_sumsq(v) > 0.0            # (one row pass of csrc/proj.hip)
        else:
            keep = (v.numpy() ** 2).sum(axis=1) > 0.0
        v, mine = v[torch.as_tensor(keep, device=v.device)], mine[keep]
        if world == 1:
            return (v if is_t else v.numpy()), mine
        everyone = [None] * world
        dist.all_gather_object(everyone, (mine, int(v.shape[1])), group=self.group)
        d = max(dd for _, dd in everyone)
        counts = [len(m) for m, _ in everyone]
        sub_idcs = np.concatenate([m for m, _ in everyone]).astype(np.int64)       # owners are in row order already
        cap = max(max(counts), 1)
        block = torch.zeros((cap, d), dtype=torch.float64, device=v.device)
        if len(mine):
            block[:len(mine)] = v.to(torch.float64)
        gathered = [torch.zeros_like(block) for _ in range(world)]
        dist.all_gather(gathered, block, group=self.group)
        rows = torch.cat([g[:c] for g, c in zip(gathered, counts)], dim=0)
        slo, shi = shard_bounds(int(sub_idcs.shape[0]), world)[0][rank]
        out = rows[slo:shi].contiguous()
        return (out if is_t else out.cpu().numpy()), sub_idcs

    # ---- Coreset interface ----------------------------------------------------------------
    def reset(self):
        self.snnls.engine.reset()
        self.snnls.reached_numeric_limit = False
        super().reset()

    def error(self):
        return self.snnls.error()

    def _owned_points(self, idcs):
        """Data rows of the selected points: every rank contributes the ones it owns."""
        mine = [(int(i), np.asarray(self.data[int(i) - self.row_begin])) for i in idcs
                if self.row_begin <= int(i) < self.row_end]
        if self.snnls.world == 1:
            everyone = [mine]
        else:
            everyone = [None] * self.snnls.world
            self.snnls.dist.all_gather_object(everyone, mine, group=self.group)
        rows = {i: r for part in everyone for i, r in part}
        return np.array([rows[int(i)] for i in idcs]) if len(idcs) else np.array([])

    def _read_solver(self):
        idx, w = self.snnls.sparse_weights()
        keep = w > 0
        idx, w = idx[keep], w[keep]
        order = np.argsort(idx, kind="stable")              # the reference reports in index order (w > 0 mask)
        self.idcs, self.wts = idx[order], w[order]
        if self.sub_idcs is not None:
            self.idcs = self.sub_idcs[self.idcs]            # solver rows -> data rows (hilbert.py:37)
        self.pts = self._owned_points(self.idcs)

    def _build(self, itrs):
        tr = self.snnls.build(itrs, float(util.TOL))
        if tr is not None:
            warn_failed_steps(self.log, tr[2], self.snnls.reached_numeric_limit)
        # (as in the reference, the solver's latch is not copied to the coreset: hilbert.py:40-42, coreset.py:34)
        self._read_solver()

    def _optimize(self):
        if not self.snnls.engine.optimize(float(util.TOL)):      # replicated solve: same outcome on every rank
            self.log.warning("self.optimize() returned a solution with increasing error. "
                             "Numeric limit possibly reached.")
            self.snnls.reached_numeric_limit = True
        self._read_solver()
