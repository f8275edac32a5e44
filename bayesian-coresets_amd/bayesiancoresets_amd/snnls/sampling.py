"""Sampling baselines (reference: bayesiancoresets/snnls/sampling.py:6-37): subclasses of the package's
``SparseNNLS`` surface that supply ``_select`` / ``_reweight`` on the host (O(1) work per iteration, SURVEY.md
section 8f #4), driven by the base class's build loop with the monotone check switched off (sampling.py:16) --
so the latch early-out, the "no data" early-out and the log lines are the reference's.
examples/synthetic_vectors/main.py:49 uses UniformSampling."""
import numpy as np

from .snnls import SparseNNLS


def _host(x):
    """ndarray view of a projector's output (a device tensor is brought to the host: these solvers are host-side)."""
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


class ImportanceSampling(SparseNNLS):
    def __init__(self, A, b):
        A = _host(A)
        super().__init__(A, A.sum(axis=1) if b is None else _host(b), check_error_monotone=False)
        n = self.w.shape[0]
        self.cts = np.zeros(n)
        self.ps = self._column_norm_probabilities()

    def _column_norm_probabilities(self):
        # proportional to the column norms; uniform when every column is zero (sampling.py:10-14)
        norms = np.sqrt((self.A ** 2).sum(axis=0))
        n = self.w.shape[0]
        return norms / norms.sum() if np.any(norms > 0) else np.ones(n) / float(n)

    def reset(self):
        super().reset()
        self.cts = np.zeros(self.w.shape[0])

    def _select(self):
        return np.random.choice(self.ps.shape[0], p=self.ps)

    def _reweight(self, f):
        self.cts[f] += 1
        self.w = (self.cts / self.cts.sum()) / self.ps


class UniformSampling(ImportanceSampling):
    def __init__(self, A, b):
        super().__init__(A, b)
        n = self.w.shape[0]
        self.ps = np.ones(n) / float(n)
