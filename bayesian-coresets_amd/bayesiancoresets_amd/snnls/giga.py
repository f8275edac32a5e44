"""GIGA: greedy iterative geodesic ascent (reference: bayesiancoresets/snnls/giga.py)."""
import numpy as np

from .snnls import DeviceSparseNNLS
from .. import _native as nat


class GIGA(DeviceSparseNNLS):
    _ALG = nat.ALG_GIGA

    def __init__(self, A, b, **kw):
        super().__init__(A, b, **kw)
        bb = self._eng.vector(0)
        self.bnorm = float(np.sqrt((bb ** 2).sum()))
        self.bn = bb / self.bnorm
