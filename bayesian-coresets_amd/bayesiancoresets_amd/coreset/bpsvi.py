"""Batch pseudocoreset by sparse variational inference (reference: bayesiancoresets/coreset/bpsvi.py:6-64).

``build(sz)`` draws ``sz`` data points as the initial pseudo-points with weights N / sz and then runs ``opt_itrs`` projected-ADAM
steps on weights AND points; every step needs the column sums of a fresh projection of the (sub-sampled) data and the
projection of the pseudo-points with its gradient in the points.  The host loop is the reference's.  With a
``DeviceProjector`` the N-sized part -- the column sums -- is the fused projection kernel (or, for the linear-regression
family, the closed form from the data's second moments): no N x S matrix is formed; the gradients of the handful of
pseudo-points come from the projector's host callbacks.  Any other projector is used as the reference uses it."""
import numpy as np

from .coreset import Coreset
from ..util.opt import nn_opt
from ..projector import DeviceProjector


class BatchPSVICoreset(Coreset):
    def __init__(self, data, ll_projector, opt_itrs, n_subsample_opt=None, step_sched=lambda i: 1.0 / (1.0 + i), **kw):
        self.data = data
        self.ll_projector = ll_projector
        self.opt_itrs = opt_itrs
        self.n_subsample_opt = None if n_subsample_opt is None else min(data.shape[0], n_subsample_opt)
        self.step_sched = step_sched
        super().__init__(**kw)

    def _build(self, sz):
        # the pseudo-points start as a uniform subsample of the data (bpsvi.py:16-21)
        init_idcs = np.random.choice(self.data.shape[0], size=sz, replace=False)
        self.pts = np.asarray(self.data[init_idcs], dtype=np.float64)
        self.wts = self.data.shape[0] / sz * np.ones(sz)
        self.idcs = -1 * np.ones(sz)
        self._optimize()

    def _get_projection(self, n_subsample, w, p):
        """(column sums of the data projection, scaling, corevecs, pgrads) after updating the projector at (w, p)."""
        self.ll_projector.update(w, p)                                              # bpsvi.py:27
        if n_subsample is None:
            sub, scaling = self.data, 1.0
        else:
            sub = self.data[np.random.randint(self.data.shape[0], size=n_subsample)]   # bpsvi.py:34
            scaling = self.data.shape[0] / n_subsample
        if isinstance(self.ll_projector, DeviceProjector):
            colsum = self.ll_projector.project_colsum(sub)
        else:
            vecs = self.ll_projector.project(sub)
            colsum = np.asarray(vecs.cpu().numpy() if hasattr(vecs, "cpu") else vecs).sum(axis=0)
        if p.size > 0:
            corevecs, pgrads = self.ll_projector.project(p, grad=True)
        else:
            corevecs, pgrads = np.zeros((0, colsum.shape[0])), np.zeros((0, colsum.shape[0], p.shape[1]))
        return colsum, scaling, corevecs, pgrads

    def _optimize(self):
        sz, d = self.wts.shape[0], self.pts.shape[1]

        def grd(x):
            w, p = x[:sz], x[sz:].reshape((sz, d))
            colsum, scaling, corevecs, pgrads = self._get_projection(self.n_subsample_opt, w, p)
            resid = scaling * colsum - w.dot(corevecs)                              # bpsvi.py:52
            wgrad = -corevecs.dot(resid) / corevecs.shape[1]
            ugrad = -(w[:, np.newaxis, np.newaxis] * pgrads * resid[np.newaxis, :, np.newaxis]).sum(axis=1) / corevecs.shape[1]
            return np.hstack((wgrad, ugrad.reshape(sz * d)))

        x0 = np.hstack((self.wts, self.pts.reshape(sz * d)))
        xf = nn_opt(x0, grd, nn_idcs=np.arange(sz), opt_itrs=self.opt_itrs, step_sched=self.step_sched)
        self.wts = xf[:sz]
        self.pts = xf[sz:].reshape((sz, d))

    def error(self):
        return 0.0   # as in the reference (bpsvi.py:62-63: KL estimate not implemented)
