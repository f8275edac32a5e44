"""Projector interface (reference: bayesiancoresets/projector.py:4-32), kept verbatim in
behaviour: ``project(pts, grad=False)`` returns an N x S array of log-likelihood vectors,
``update(wts, pts)`` refreshes the Monte-Carlo samples.  A projector may also return a
``torch.Tensor`` that already lives on the GPU; the solvers ingest it without a host copy."""
import numpy as np


class Projector(object):
    def project(self, pts, grad=False):
        raise NotImplementedError

    def update(self, wts, pts):
        raise NotImplementedError


class BlackBoxProjector(Projector):
    """Monte-Carlo projection: rows are ``loglikelihood(pts, samples)`` minus their row mean
    (projector.py:19-21; note: no 1/sqrt(S) scaling)."""

    def __init__(self, sampler, projection_dimension, loglikelihood, grad_loglikelihood=None):
        self.projection_dimension = projection_dimension
        self.sampler = sampler
        self.loglikelihood = loglikelihood
        self.grad_loglikelihood = grad_loglikelihood
        # samplers must accept empty wts/pts (projector.py:17)
        self.update(np.array([]), np.array([]))

    def update(self, wts, pts):
        self.samples = self.sampler(self.projection_dimension, wts, pts)

    def project(self, pts, grad=False):
        lls = self.loglikelihood(pts, self.samples)
        lls -= lls.mean(axis=1)[:, np.newaxis]
        if not grad:
            return lls
        if self.grad_loglikelihood is None:
            raise ValueError("grad_loglikelihood was requested but not initialized in BlackBoxProjector.project")
        glls = self.grad_loglikelihood(pts, self.samples)
        glls -= glls.mean(axis=2)[:, :, np.newaxis]
        return lls, glls
