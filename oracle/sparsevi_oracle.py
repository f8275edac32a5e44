"""CPU oracle for the SparseVI path (BASELINE.json configs[4]).

TEST INFRASTRUCTURE ONLY (see oracle/snnls_oracle.py): imported by ``tests/`` and by the ``cpu_baseline`` leg of
``bench.py``; the product never imports it.

NumPy restatement, written from SURVEY.md section 8a, of the reference's sparse variational-inference coreset
(trevorcampbell/bayesian-coresets @ v0.9.1):

* greedy step: select then optimise ............... bayesiancoresets/coreset/sparsevi.py:16-21
* tangent-space refresh + (sub)projection ......... bayesiancoresets/coreset/sparsevi.py:23-42
* residual, correlations, arg-max, append ......... bayesiancoresets/coreset/sparsevi.py:44-67
* weight optimisation (projected ADAM) ............ bayesiancoresets/coreset/sparsevi.py:69-76, util/opt.py:4-28
* Monte-Carlo projection with row centring ........ bayesiancoresets/projector.py:19-24
* Gaussian linear-regression likelihood ........... examples/common/model_linreg.py:4-10

organised as one explicit state (weights / indices / points / samples) advanced by ``step()`` instead of the
reference's class hierarchy.  Pinned by tests/test_oracle_golden.py against tests/golden/svi_golden.npz (F6) and
rbf_golden.npz (F6b), both produced by the reference itself.
"""
import numpy as np


def linreg_loglik(z, th, sigsq):
    """model_linreg.py:4-10: log N(y | x.theta, sigsq) for rows z = [x, y] and parameter samples th (S x D)."""
    z, th = np.atleast_2d(z), np.atleast_2d(th)
    pred = z[:, :-1].dot(th.T)
    y = z[:, -1][:, None]
    return -0.5 * np.log(2.0 * np.pi * sigsq) - (y ** 2 - 2.0 * pred * y + pred ** 2) / (2.0 * sigsq)


class SparseVIOracle(object):
    def __init__(self, data, sampler, loglik, S, opt_itrs=100, step_sched=lambda i: 1.0 / (1.0 + i),
                 n_subsample_select=None, n_subsample_opt=None):
        self.data, self.sampler, self.loglik, self.S = data, sampler, loglik, S
        self.opt_itrs, self.step_sched = opt_itrs, step_sched
        n = data.shape[0]
        self.n_sel = None if n_subsample_select is None else min(n, n_subsample_select)       # sparsevi.py:12
        self.n_opt = None if n_subsample_opt is None else min(n, n_subsample_opt)             # sparsevi.py:13
        self.wts = np.zeros(0)
        self.idcs = np.zeros(0, dtype=np.int64)
        self.pts = np.zeros((0, data.shape[1]))
        self.samples = sampler(S, np.array([]), np.array([]))                                 # projector.py:17

    def _project(self, pts):
        ll = self.loglik(pts, self.samples)                                                    # projector.py:20
        return ll - ll.mean(axis=1)[:, None]                                                   # projector.py:21

    def _tangent(self, n_sub, w):
        """sparsevi.py:23-42: refresh the samples at (w, pts); project the data (or a random subsample) and the core."""
        self.samples = self.sampler(self.S, w, self.pts)                                       # sparsevi.py:25 / projector.py:24
        if n_sub is None:
            sub, vecs, scaling = None, self._project(self.data), 1.0
        else:
            sub = np.random.randint(self.data.shape[0], size=n_sub)
            vecs, scaling = self._project(self.data[sub]), self.data.shape[0] / n_sub
        core = self._project(self.pts) if self.pts.size > 0 else np.zeros((0, vecs.shape[1]))
        return vecs, scaling, sub, core

    def select(self):
        vecs, scaling, sub, core = self._tangent(self.n_sel, self.wts)
        resid = scaling * vecs.sum(axis=0) - self.wts.dot(core)                                # sparsevi.py:47
        corrs = vecs.dot(resid) / np.sqrt((vecs ** 2).sum(axis=1)) / vecs.shape[1]             # sparsevi.py:50
        corecorrs = np.fabs(core.dot(resid) / np.sqrt((core ** 2).sum(axis=1))) / core.shape[1]   # sparsevi.py:52
        if corecorrs.size == 0 or corrs.max() > corecorrs.max():                               # sparsevi.py:55
            f = int(sub[np.argmax(corrs)]) if sub is not None else int(np.argmax(corrs))
            if f not in self.idcs:                                                             # sparsevi.py:59
                self.wts = np.append(self.wts, 0.0)
                self.idcs = np.append(self.idcs, f)
                self.pts = np.vstack((self.pts, self.data[f][None, :]))

    def optimize(self):
        def grad(w):                                                                           # sparsevi.py:70-74
            vecs, scaling, sub, core = self._tangent(self.n_opt, w)
            resid = scaling * vecs.sum(axis=0) - w.dot(core)
            return -core.dot(resid) / core.shape[1]
        # util/opt.py:4-28 (all coordinates constrained non-negative)
        x = self.wts.copy()
        m1, m2 = np.zeros(x.shape[0]), np.zeros(x.shape[0])
        b1, b2, eps = 0.9, 0.999, 1e-8
        for i in range(self.opt_itrs):
            g = grad(x)
            m1 = b1 * m1 + (1.0 - b1) * g
            m2 = b2 * m2 + (1.0 - b2) * g ** 2
            x = x - self.step_sched(i) * m1 / (1.0 - b1 ** (i + 1)) / (eps + np.sqrt(m2 / (1.0 - b2 ** (i + 1))))
            x = np.maximum(x, 0.0)
        self.wts = x

    def step(self):
        self.select()
        self.optimize()
