"""Non-negative orthogonal matching pursuit (reference: bayesiancoresets/snnls/orthopursuit.py)."""
from .snnls import SparseNNLS
from .. import _native as nat


class OrthoPursuit(SparseNNLS):
    _ALG = nat.ALG_OMP
