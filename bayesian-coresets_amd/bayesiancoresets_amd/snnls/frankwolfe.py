"""Frank-Wolfe on the norm-weighted simplex (reference: bayesiancoresets/snnls/frankwolfe.py)."""
from .snnls import SparseNNLS
from .. import _native as nat


class FrankWolfe(SparseNNLS):
    _ALG = nat.ALG_FW
