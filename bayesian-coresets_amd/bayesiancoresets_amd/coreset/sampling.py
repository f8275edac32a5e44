"""Uniform-sampling coreset baseline (reference: bayesiancoresets/coreset/sampling.py:5-27);
host-side bookkeeping, kept for API parity."""
import numpy as np

from .coreset import Coreset


class UniformSamplingCoreset(Coreset):
    def __init__(self, data, **kw):
        super().__init__(**kw)
        self.data = data
        self._counts = {}

    def reset(self):
        self._counts = {}
        super().reset()

    def _build(self, itrs):
        n = self.data.shape[0]
        for _ in range(itrs):
            f = int(np.random.randint(n))
            self._counts[f] = self._counts.get(f, 0) + 1
        idcs = np.array(list(self._counts.keys()))      # insertion order == first-draw order
        cts = np.array(list(self._counts.values()))
        self.wts = n * cts / cts.sum()
        self.idcs = idcs
        self.pts = self.data[self.idcs]
