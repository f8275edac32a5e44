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
that own them.  The subsampling branch of the reference (``n_subsample``) is not offered here.
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

    def __init__(self, local_data, ll_projector, n_global, snnls=GIGA, group=None, engine_factory=None, **kw):
        if not (isinstance(snnls, type) and issubclass(snnls, _DeviceSolver) and snnls._ALG is not None):
            raise ValueError("ShardedHilbertCoreset needs a device solver class (GIGA, FrankWolfe, OrthoPursuit)")
        vecs = ll_projector.project(local_data)
        d = int(vecs.shape[1])
        self.snnls = ShardedSolver(snnls._ALG, int(n_global), d, group=group, engine_factory=engine_factory)
        if int(vecs.shape[0]) != self.snnls.n_local:
            raise ValueError("this rank must hold global rows [%d, %d) (ShardedHilbertCoreset.local_rows); got %d rows"
                             % (self.snnls.row_begin, self.snnls.row_end, int(vecs.shape[0])))
        if self.snnls.n_local:
            self.snnls.load_local(vecs)
        rc = self.snnls.finalize(None)                     # b = column sums over all shards (hilbert.py:24)
        if rc == nat.ERR_ZERO_ROW:
            raise ValueError("ShardedHilbertCoreset.__init__(): A must not have any 0 columns")   # giga.py:11-12
        if rc == nat.ERR_ZERO_B:
            raise NumericalPrecisionError("norm of b must be > 0")                                 # giga.py:16-17
        if rc != nat.OK:
            raise nat.EngineError(rc, "finalize failed")
        self.data = local_data
        self.group = group
        self.row_begin, self.row_end = self.snnls.row_begin, self.snnls.row_end
        super().__init__(**kw)

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
        self.pts = self._owned_points(self.idcs)

    def _build(self, itrs):
        tr = self.snnls.build(itrs, float(util.TOL))
        if tr is not None:
            warn_failed_steps(self.log, tr[2])
        # (as in the reference, the solver's latch is not copied to the coreset: hilbert.py:40-42, coreset.py:34)
        self._read_solver()

    def _optimize(self):
        if not self.snnls.engine.optimize(float(util.TOL)):      # replicated solve: same outcome on every rank
            self.log.warning("self.optimize() returned a solution with increasing error. "
                             "Numeric limit possibly reached.")
            self.snnls.reached_numeric_limit = True
        self._read_solver()
