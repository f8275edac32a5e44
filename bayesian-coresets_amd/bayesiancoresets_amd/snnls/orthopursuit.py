"""Non-negative orthogonal matching pursuit (reference: bayesiancoresets/snnls/orthopursuit.py)."""
from .snnls import DeviceSparseNNLS
from .. import _native as nat


class OrthoPursuit(DeviceSparseNNLS):
    _ALG = nat.ALG_OMP
