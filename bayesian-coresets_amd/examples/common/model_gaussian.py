"""Multivariate normal with unknown mean: the model of the reference's Gaussian experiment (examples/common/model_gaussian.py
there; rows are the observations x, the parameter is the mean theta, the covariance Sig is known, the prior N(mu0, Sig0)).

    log p(x | theta) = -D/2 log(2 pi) - 1/2 log det Sig - 1/2 (x - theta)^T Sig^-1 (x - theta)            (model_gaussian.py:4-10)
    posterior of weighted data: Sigma_w^-1 = Sig0^-1 + (sum w) Sig^-1,  mu_w = Sigma_w (Sig0^-1 mu0 + Sig^-1 sum_n w_n x_n)   (:24-31)
"""
import numpy as np
import scipy.linalg as sl


def log_likelihood(x, th, Siginv, logdetSig):
    """N x S matrix of log-likelihoods (NumPy): the callback form of a host ``bc.BlackBoxProjector``."""
    x, th = np.atleast_2d(x), np.atleast_2d(th)
    xs = x.dot(Siginv)
    quad_x = (x * xs).sum(axis=1)
    quad_t = (th * th.dot(Siginv)).sum(axis=1)
    return -0.5 * x.shape[1] * np.log(2.0 * np.pi) - 0.5 * logdetSig - 0.5 * (quad_x[:, None] + quad_t[None, :] - 2.0 * xs.dot(th.T))


def weighted_posterior(mu0, Sig0inv, Siginv, x, w):
    """(mu, U) with Sigma = U U^T."""
    w = np.asarray(w, dtype=np.float64)
    L = np.linalg.cholesky(Sig0inv + w.sum() * Siginv)
    U = sl.solve_triangular(L, np.eye(L.shape[0]), lower=True, check_finite=False).T
    rhs = Sig0inv.dot(mu0)
    if w.shape[0] > 0:
        rhs = rhs + Siginv.dot((w[:, None] * np.atleast_2d(x)).sum(axis=0))
    return U.dot(U.T.dot(rhs)), U


def gaussian_kl(mu0, Sig0, mu1, Sig1inv):
    """KL(N(mu0, Sig0) || N(mu1, Sig1)) (model_gaussian.py:17-21)."""
    diff = mu1 - mu0
    return 0.5 * (np.trace(Sig1inv.dot(Sig0)) + diff.dot(Sig1inv).dot(diff)
                  - np.linalg.slogdet(Sig1inv)[1] - np.linalg.slogdet(Sig0)[1] - mu0.shape[0])


def tangent_space_projector(bc, mu0, Sig0inv, Siginv):
    """The EXACT tangent-space projection of this model (examples/gaussian/main.py:117-138, `GaussianProjector`) as a
    ``bc.Projector``: with theta = mu_w + U_w eps the centred log-likelihood of x is linear in eps up to a term that is
    the same for every x, so D + 1 numbers per point carry all inner products --

        [ (x - mu_w)^T L Psi_L ,  sqrt(tr(Psi^T Psi) / 2) ] * sqrt(D + 1),     Sig^-1 = L L^T,  Psi_L = L^T U_w,  Psi = Psi_L Psi_L^T.

    ``update(wts, pts)`` moves the tangent point to the weighted posterior (an empty coreset: the prior, main.py:133-136)."""
    mu0 = np.asarray(mu0, dtype=np.float64)
    LSigInv = np.linalg.cholesky(Siginv)

    class GaussianTangentProjector(bc.Projector):
        def __init__(self):
            self.update(None, None)

        def update(self, wts=None, pts=None):
            if wts is None or pts is None or np.asarray(pts).shape[0] == 0:
                wts, pts = np.zeros(1), np.zeros((1, mu0.shape[0]))
            self.muw, self.USigw = weighted_posterior(mu0, Sig0inv, Siginv, np.atleast_2d(pts), np.asarray(wts, dtype=np.float64))

        def project(self, pts, grad=False):
            if grad:
                raise NotImplementedError("the exact projector has no usable gradient form (the reference's names an undefined variable)")
            pts = np.atleast_2d(np.asarray(pts, dtype=np.float64))
            PsiL = LSigInv.T.dot(self.USigw)
            Psi = PsiL.dot(PsiL.T)
            nu = (pts - self.muw).dot(LSigInv).dot(PsiL)
            last = np.sqrt(0.5 * np.trace(Psi.T.dot(Psi))) * np.ones((pts.shape[0], 1))
            out = np.hstack((nu, last))
            return out * np.sqrt(out.shape[1])

    return GaussianTangentProjector()
