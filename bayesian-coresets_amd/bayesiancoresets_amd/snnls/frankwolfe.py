"""Frank-Wolfe on the norm-weighted simplex (reference: bayesiancoresets/snnls/frankwolfe.py)."""
from .snnls import DeviceSparseNNLS
from .. import _native as nat


class FrankWolfe(DeviceSparseNNLS):
    _ALG = nat.ALG_FW
