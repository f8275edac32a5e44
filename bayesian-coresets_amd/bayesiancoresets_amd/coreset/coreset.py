"""Abstract coreset shell (reference: bayesiancoresets/coreset/coreset.py:7-70).

State is the triple ``wts / idcs / pts``; ``build`` is guarded by the numeric-limit latch and
``itrs <= 0``; ``get`` returns the strictly positive part; ``optimize`` keeps a result only when
the error did not grow by more than a factor (1 + TOL)."""
import numpy as np

from .. import util
from ..util.errors import NumericalPrecisionError
from ..util.log import object_logger


def _empty_state():
    return np.array([]), np.array([], dtype=np.int64), np.array([])


class Coreset(object):
    def __init__(self):
        self.alg_name, self.log = object_logger(self)
        self.reached_numeric_limit = False
        self.wts, self.idcs, self.pts = _empty_state()

    def reset(self):
        self.wts, self.idcs, self.pts = _empty_state()
        self.reached_numeric_limit = False

    def size(self):
        return (self.wts > 0).sum()

    def get(self):
        if self.wts.shape[0] == 0:
            return np.array([]), np.array([]), np.array([])
        keep = self.wts > 0
        return self.wts[keep], self.pts[keep, :], self.idcs[keep]

    def error(self):
        raise NotImplementedError()

    def build(self, itrs):
        if self.reached_numeric_limit or itrs <= 0:
            return
        self._build(itrs)
        if self.reached_numeric_limit:
            self.log.warning("the numeric limit has been reached. No more points will be added. size = "
                             + str(self.size()) + ", error = " + str(self.error()))

    def optimize(self):
        before = (self.wts.copy(), self.idcs.copy(), self.pts.copy())
        try:
            prev_cost = self.error()
            self._optimize()
            new_cost = self.error()
            if new_cost > prev_cost * (1.0 + util.TOL):
                raise NumericalPrecisionError(
                    "self.optimize() returned a solution with increasing error. Numeric limit possibly reached: "
                    "preverr = " + str(prev_cost) + " err = " + str(new_cost) + ".\n If the two errors are very "
                    "close, try running bc.util.set_tolerance(tol) with tol > current tol = " + str(util.TOL))
        except NumericalPrecisionError as e:
            self.log.warning(e)
            self.wts, self.idcs, self.pts = before
            self.reached_numeric_limit = True

    def _optimize(self):
        raise NotImplementedError

    def _build(self, itrs):
        raise NotImplementedError
