"""Sampling baselines (reference: bayesiancoresets/snnls/sampling.py:6-37).  O(1) work per
iteration, so they stay host NumPy; kept for API parity --
examples/synthetic_vectors/main.py:49 uses UniformSampling."""
import numpy as np

from ..util.log import object_logger


class ImportanceSampling(object):
    def __init__(self, A, b):
        self.alg_name, self.log = object_logger(self)
        self.A, self.b = A, b
        self.reached_numeric_limit = False
        self.check_error_monotone = False
        n = A.shape[1]
        self.w = np.zeros(n)
        self.cts = np.zeros(n)
        self.ps = self._probabilities()

    def _probabilities(self):
        ps = np.sqrt((self.A ** 2).sum(axis=0))
        if np.any(ps > 0):
            return ps / ps.sum()
        return np.ones(self.A.shape[1]) / float(self.A.shape[1])

    def reset(self):
        self.w = np.zeros(self.A.shape[1])
        self.cts = np.zeros(self.A.shape[1])
        self.reached_numeric_limit = False

    def size(self):
        return (self.w > 0).sum()

    def weights(self):
        return self.w.copy()

    def error(self):
        return np.sqrt(((self.A.dot(self.w) - self.b) ** 2).sum())

    def build(self, itrs):
        if self.A.size == 0:
            self.log.warning("there are no data, returning.")
            return
        for _ in range(itrs):
            f = np.random.choice(self.ps.shape[0], p=self.ps)
            self.cts[f] += 1
            self.w = (self.cts / self.cts.sum()) / self.ps

    def optimize(self):
        from scipy.optimize import nnls
        prev_cost, prev_w = self.error(), self.w.copy()
        nz = self.w > 0
        self.w[nz] = nnls(self.A[:, nz], self.b, maxiter=100 * self.A.shape[1])[0]
        from .. import util
        if self.error() > prev_cost * (1.0 + util.TOL):
            self.w = prev_w
            self.reached_numeric_limit = True


class UniformSampling(ImportanceSampling):
    def _probabilities(self):
        return np.ones(self.A.shape[1]) / float(self.A.shape[1])
