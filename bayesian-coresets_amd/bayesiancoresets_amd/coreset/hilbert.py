"""Hilbert coreset: project once, then greedy sparse NNLS on the projected vectors
(reference: bayesiancoresets/coreset/hilbert.py:7-48).  This is the drop-in boundary: same
constructor, the solver class is a keyword (``snnls=GIGA``), and solver weights are translated
to ``wts / idcs / pts`` in index order."""
import numpy as np

from ..snnls.giga import GIGA
from ..snnls.snnls import DeviceSparseNNLS as _DeviceSolver
from .coreset import Coreset


def _is_torch(x):
    try:
        import torch
        return isinstance(x, torch.Tensor)
    except ImportError:
        return False


class HilbertCoreset(Coreset):
    def __init__(self, data, ll_projector, n_subsample=None, snnls=GIGA, **kw):
        rows = None if n_subsample is None else self._draw_subsample(data.shape[0], n_subsample)
        # a device projector feeding a device solver on the full data: take the raw log-likelihoods and let the solver's
        # constructor pass subtract the row means (projector.py:21) -- one pass over N x S less (csrc/ingest.hip)
        fold = (rows is None and hasattr(ll_projector, "project_uncentred")
                and isinstance(snnls, type) and issubclass(snnls, _DeviceSolver) and self._accepts(snnls, "center_rows"))
        if fold:
            vecs = ll_projector.project_uncentred(data)
        else:
            vecs = ll_projector.project(data if rows is None else data[rows])
        if rows is None:
            rows = np.arange(data.shape[0])
        else:
            # subsample branch only: zero vectors cannot change the coreset and would make the solver
            # constructor raise (hilbert.py:19-22)
            keep = self._nonzero_rows(vecs)
            if not keep.all():
                vecs = self._take_rows(vecs, keep)
            rows = rows[keep]
        self.snnls = self._make_solver(snnls, vecs, center_rows=fold)
        self.sub_idcs = rows
        self.data = data
        super().__init__(**kw)

    # ---- construction helpers ------------------------------------------------------------
    @staticmethod
    def _accepts(fn, keyword):
        """Does ``fn`` (a class: its __init__) take ``keyword``?  A user subclass of a device solver written against the
        reference's (A, b) signature does not know ``center_rows``: it then gets the centred vectors of project() instead."""
        import inspect
        try:
            params = inspect.signature(fn).parameters
        except (TypeError, ValueError):
            return False
        return keyword in params or any(q.kind == q.VAR_KEYWORD for q in params.values())

    @staticmethod
    def _draw_subsample(n, n_subsample):
        # randint then unique: cheap for huge n, duplicates removed (hilbert.py:16)
        return np.unique(np.random.randint(n, size=n_subsample))

    @staticmethod
    def _nonzero_rows(vecs):
        if _is_torch(vecs) and vecs.is_cuda:
            from .. import _native as nat
            return nat.device_row_sumsq(vecs) > 0.0         # (a device projector's output stays on the device: one row pass)
        if _is_torch(vecs):
            vecs = vecs.detach().numpy()
        return np.sqrt((vecs ** 2).sum(axis=1)) > 0.0

    @staticmethod
    def _take_rows(vecs, keep):
        if _is_torch(vecs):
            import torch
            return vecs[torch.as_tensor(keep, device=vecs.device)]
        return vecs[keep, :]

    @staticmethod
    def _make_solver(snnls, vecs, center_rows=False):
        """Solver on A = vecs^T with b = the column sums of vecs (hilbert.py:24).  The device solvers form b
        themselves during ingest (fp64 chunked column sums, csrc/ingest.hip) when handed ``b=None``; host-only
        solver classes (the sampling baselines) get the NumPy sum as in the reference."""
        on_device = isinstance(snnls, type) and issubclass(snnls, _DeviceSolver)
        if on_device:
            if center_rows:
                return snnls(vecs.t(), None, center_rows=True)
            return snnls(vecs.t() if _is_torch(vecs) else vecs.T, None)
        if _is_torch(vecs):
            vecs = vecs.detach().cpu().numpy()      # a device projector's output, for a host-side solver class
        return snnls(vecs.T, vecs.sum(axis=0))

    # ---- Coreset interface ----------------------------------------------------------------
    def reset(self):
        self.snnls.reset()
        super().reset()

    def _read_solver(self):
        sup = getattr(self.snnls, "support", None)
        if sup is not None:
            # a device solver: its sparse list (k entries), in index order -- the same triple as the dense form below
            idx, self.wts = sup()
            self.idcs = self.sub_idcs[idx]
        else:
            w = self.snnls.weights()                        # hilbert.py:36-38
            support = w > 0
            self.wts, self.idcs = w[support], self.sub_idcs[support]
        self.pts = self.data[self.idcs]

    def _build(self, itrs):
        self.snnls.build(itrs)
        self._read_solver()

    def _optimize(self):
        self.snnls.optimize()
        self._read_solver()

    def error(self):
        return self.snnls.error()
