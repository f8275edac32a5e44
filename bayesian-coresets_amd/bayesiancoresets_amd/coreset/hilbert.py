"""Hilbert coreset: project once, then greedy sparse NNLS on the projected vectors
(reference: bayesiancoresets/coreset/hilbert.py:7-48).  This is the drop-in boundary: same
constructor, the solver class is a keyword (``snnls=GIGA``), and solver weights are translated
to ``wts / idcs / pts`` in index order."""
import numpy as np

from ..snnls.giga import GIGA
from .coreset import Coreset


def _is_torch(x):
    try:
        import torch
        return isinstance(x, torch.Tensor)
    except ImportError:
        return False


class HilbertCoreset(Coreset):
    def __init__(self, data, ll_projector, n_subsample=None, snnls=GIGA, **kw):
        if n_subsample is None:
            sub_idcs = np.arange(data.shape[0])
            vecs = ll_projector.project(data)
        else:
            # hilbert.py:16-22: random subsample without duplicates, zero vectors dropped
            sub_idcs = np.unique(np.random.randint(data.shape[0], size=n_subsample))
            vecs = ll_projector.project(data[sub_idcs])
            if _is_torch(vecs):
                nonzero = ((vecs ** 2).sum(dim=1).sqrt() > 0.0).cpu().numpy()
                if not nonzero.all():
                    import torch
                    vecs = vecs[torch.as_tensor(nonzero, device=vecs.device)]
            else:
                nonzero = np.sqrt((vecs ** 2).sum(axis=1)) > 0.0
                vecs = vecs[nonzero, :]
            sub_idcs = sub_idcs[nonzero]
        # b = vecs.sum(axis=0) on the host when the vectors are host arrays (bit-identical to
        # hilbert.py:24); a device-resident projection lets the engine form the column sums.
        if _is_torch(vecs):
            b = None if vecs.device.type == "cuda" else vecs.sum(dim=0).numpy()
            self.snnls = snnls(vecs.t(), b)
        else:
            self.snnls = snnls(vecs.T, vecs.sum(axis=0))
        self.sub_idcs = sub_idcs
        self.data = data
        super().__init__(**kw)

    def reset(self):
        self.snnls.reset()
        super().reset()

    def _read_solver(self):
        w = self.snnls.weights()
        keep = w > 0
        self.wts = w[keep]
        self.idcs = self.sub_idcs[keep]
        self.pts = self.data[self.idcs]

    def _build(self, itrs):
        self.snnls.build(itrs)
        self._read_solver()

    def _optimize(self):
        self.snnls.optimize()
        self._read_solver()

    def error(self):
        return self.snnls.error()
