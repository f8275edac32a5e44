"""Test-side restatement of the config-3 workload: synthetic logistic
regression data and the logistic log-likelihood of examples/common/model_lr.py:25-32
(-log1p(exp(-z.theta)), linear branch for the argument >= 100)."""
import numpy as np


def make_data(seed, N, D):
    rs = np.random.RandomState(seed)
    X = rs.randn(N, D)
    th = 3.0 * np.ones(D)
    ps = 1.0 / (1.0 + np.exp(-(X * th).sum(axis=1)))
    y = (rs.rand(N) <= ps).astype(int)
    y[y == 0] = -1
    return y[:, np.newaxis] * X


def log_likelihood(z, th):
    m = -np.atleast_2d(z).dot(np.atleast_2d(th).T)
    small = m < 100
    out = np.empty_like(m)
    out[small] = -np.log1p(np.exp(m[small]))
    out[~small] = -m[~small]
    return out
