"""Test-side restatements of the reference's example likelihoods and samplers (inputs of the
projection tests; the product never imports this):
  logistic   examples/common/model_lr.py:25-32      (also in tests/lr_workload.py)
  poisson    examples/common/model_poiss.py:25-38
  linreg     examples/common/model_linreg.py:4-10, weighted posterior :24-37"""
import numpy as np
import scipy.linalg as sl
from scipy.special import gammaln

from lr_workload import log_likelihood as logistic_log_likelihood  # noqa: F401


def poisson_log_likelihood(z, th):
    z, th = np.atleast_2d(z), np.atleast_2d(th)
    x, y = z[:, :-1], z[:, -1][:, None]
    s = x.dot(th.T)
    big = s > -100
    s[big] = np.log(np.maximum(s[big], 0) + np.log1p(np.exp(-np.fabs(s[big]))))
    return y * s - gammaln(y + 1) - np.exp(s)


def linreg_log_likelihood(z, th, sigsq):
    z, th = np.atleast_2d(z), np.atleast_2d(th)
    x, y = z[:, :-1], z[:, -1][:, None]
    xst = x.dot(th.T)
    return -0.5 * np.log(2.0 * np.pi * sigsq) - 1.0 / (2.0 * sigsq) * (y ** 2 - 2 * xst * y + xst ** 2)


def linreg_grad_z_log_likelihood(z, th, sigsq):
    """N x S x (D + 1): the gradient the reference's example hands to its projectors (model_linreg.py:12-17: residual / sigsq
    times [theta, 1] -- as written there, including the sign of the response component)."""
    z, th = np.atleast_2d(z), np.atleast_2d(th)
    x, y = z[:, :-1], z[:, -1][:, None]
    r = (y - x.dot(th.T)) / sigsq
    return r[:, :, None] * np.hstack((th, np.ones((th.shape[0], 1))))[None, :, :]


def linreg_weighted_post(th0, Sig0inv, sigsq, z, w):
    """Gaussian posterior of the weighted linear regression: mean and U with Sigma = U U^T."""
    if w.shape[0] > 0:
        z = np.atleast_2d(z)
        X, Y = z[:, :-1], z[:, -1]
        L = np.linalg.cholesky(Sig0inv + (w[:, None] * X).T.dot(X) / sigsq)
        U = sl.solve_triangular(L, np.eye(L.shape[0]), lower=True, check_finite=False).T
        mu = U.dot(U.T).dot(Sig0inv.dot(th0) + (w[:, None] * Y[:, None] * X).sum(axis=0) / sigsq)
    else:
        mu = th0
        L = np.linalg.cholesky(Sig0inv)
        U = sl.solve_triangular(L, np.eye(L.shape[0]), lower=True, check_finite=False).T
    return mu, U


def make_linreg_data(seed, N, D, sigma=1.0):
    rs = np.random.RandomState(seed)
    X = rs.randn(N, D)
    th = rs.randn(D)
    y = X.dot(th) + sigma * rs.randn(N)
    return np.hstack((X, y[:, None]))


def make_poisson_data(seed, N, D):
    rs = np.random.RandomState(seed)
    X = np.hstack((rs.randn(N, D - 1), np.ones((N, 1))))
    th = 0.5 * rs.randn(D)
    y = rs.poisson(np.log1p(np.exp(X.dot(th))))
    return np.hstack((X, y[:, None].astype(float)))


def linreg_sampler(mu0, Sig0, sigsq):
    """examples/linear_regression/main.py:141-147 (sampler_w): draws from the weighted posterior."""
    Sig0inv = np.linalg.inv(Sig0)

    def sampler(n, wts, pts):
        if wts is None or pts is None or np.asarray(pts).shape[0] == 0:
            mu, U = mu0, np.linalg.cholesky(Sig0)
        else:
            mu, U = linreg_weighted_post(mu0, Sig0inv, sigsq, pts, wts)
        return mu + np.random.randn(n, mu.shape[0]).dot(U.T)
    return sampler
