"""Abstract coreset shell (reference: bayesiancoresets/coreset/coreset.py:7-70).

A coreset is the triple ``wts / idcs / pts``.  ``build`` is a no-op once the numeric-limit latch is
set or when asked for ``itrs <= 0``; ``get`` hands out only the strictly positive weights; ``optimize``
keeps its result unless the error grew by more than a factor (1 + TOL), in which case the previous
triple is restored and the latch is set.  Subclasses supply ``_build``, ``_optimize`` and ``error``."""
import numpy as np

from .. import util
from ..util.errors import NumericalPrecisionError
from ..util.log import object_logger

_STATE = ("wts", "idcs", "pts")


class Coreset(object):
    def __init__(self):
        self.alg_name, self.log = object_logger(self)
        self._clear()

    def _clear(self):
        self.reached_numeric_limit = False
        self.wts = np.array([])
        self.idcs = np.array([], dtype=np.int64)
        self.pts = np.array([])

    def _snapshot(self):
        return {k: getattr(self, k).copy() for k in _STATE}

    def _restore(self, snap):
        for k in _STATE:
            setattr(self, k, snap[k])

    # ---- public surface ---------------------------------------------------------
    def reset(self):
        self._clear()

    def size(self):
        return (self.wts > 0).sum()

    def get(self):
        if self.wts.shape[0] == 0:                       # nothing built yet: three empty arrays
            return np.array([]), np.array([]), np.array([])
        positive = self.wts > 0
        return self.wts[positive], self.pts[positive, :], self.idcs[positive]

    def error(self):
        raise NotImplementedError()

    def build(self, itrs):
        if itrs <= 0 or self.reached_numeric_limit:
            return
        self._build(itrs)
        if self.reached_numeric_limit:
            self.log.warning("the numeric limit has been reached. No more points will be added. size = %s, error = %s"
                             % (self.size(), self.error()))

    def optimize(self):
        snap = self._snapshot()
        try:
            cost_before = self.error()
            self._optimize()
            cost_after = self.error()
            if cost_after > cost_before * (1.0 + util.TOL):
                raise NumericalPrecisionError(
                    "self.optimize() returned a solution with increasing error. Numeric limit possibly reached: "
                    "preverr = %s err = %s.\n If the two errors are very close, try running "
                    "bc.util.set_tolerance(tol) with tol > current tol = %s" % (cost_before, cost_after, util.TOL))
        except NumericalPrecisionError as e:
            self.log.warning(e)
            self._restore(snap)
            self.reached_numeric_limit = True

    # ---- to be provided by subclasses ------------------------------------------------
    def _build(self, itrs):
        raise NotImplementedError

    def _optimize(self):
        raise NotImplementedError
