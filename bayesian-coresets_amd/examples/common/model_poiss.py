"""Poisson regression with a softplus rate and a standard-normal prior: the second model of the reference's
logistic / Poisson regression experiment (examples/common/model_poiss.py there; rows are z = [x, y], y a count).

    rate(s) = log(1 + e^s),  s = x.theta;    log p(z | theta) = y log rate - rate - log(y!)      (model_poiss.py:25-38)
    log p(theta) = -D/2 log(2 pi) - |theta|^2 / 2                                               (model_poiss.py:40-42)

``log_likelihood`` is the callback form a host ``bc.BlackBoxProjector`` takes; on the device the same function is the
"poisson" family of ``bc.DeviceProjector`` (csrc/proj.hip).  ``laplace_fit`` is the tangent-space location the experiment
uses for its projectors (logistic_poisson_regression/main.py:15-41 `get_laplace`): the MAP of the weighted log joint and the
inverse negative Hessian there -- Newton's method with step halving on the concave objective instead of SciPy's BFGS
(same maximiser)."""
import numpy as np
from scipy.special import gammaln


def _rate_and_log(s):
    """(rate, log rate) of softplus, with log rate = s where the rate is e^s to every bit (s <= -100: model_poiss.py:27-31)."""
    rate = np.maximum(s, 0.0) + np.log1p(np.exp(-np.fabs(s)))
    with np.errstate(divide="ignore"):
        lr = np.where(s > -100.0, np.log(np.where(rate > 0.0, rate, 1.0)), s)
    return rate, lr


def log_likelihood(z, th):
    """N x S matrix of log-likelihoods (NumPy)."""
    z, th = np.atleast_2d(z), np.atleast_2d(th)
    s = z[:, :-1].dot(th.T)
    y = z[:, -1][:, None]
    rate, lr = _rate_and_log(s)
    return y * lr - gammaln(y + 1.0) - np.exp(lr)


def _derivs(s, y):
    """d/ds and d^2/ds^2 of y log rate(s) - rate(s)."""
    rate, _ = _rate_and_log(s)
    e = np.exp(-np.fabs(s))
    sig = np.where(s >= 0.0, 1.0, e) / (1.0 + e)              # rate'(s) = sigmoid(s), relative accuracy in both tails
    dsig = e / ((1.0 + e) * (1.0 + e))                        # rate''(s)
    safe = np.where(rate > 0.0, rate, 1.0)
    r1 = np.where(rate > 0.0, sig / safe, 1.0)                # rate' / rate  (-> 1 as s -> -inf)
    g = y * r1 - sig
    # (rate'' rate - rate'^2) / rate^2  (-> 0 as s -> -inf, where log rate = s)
    r2 = np.where(rate > 0.0, (dsig * safe - sig * sig) / (safe * safe), 0.0)
    h = y * r2 - dsig
    return g, h


def laplace_fit(Z, wts=None, mu0=None, tol=1e-10, max_iter=200):
    """(mu, cov) of the Laplace approximation to the (weighted) posterior."""
    Z = np.atleast_2d(np.asarray(Z, dtype=np.float64))
    X, y = Z[:, :-1], Z[:, -1]
    w = np.ones(Z.shape[0]) if wts is None else np.asarray(wts, dtype=np.float64)
    D = X.shape[1]
    th = np.zeros(D) if mu0 is None else np.asarray(mu0, dtype=np.float64).copy()

    def objective(t):
        rate, lr = _rate_and_log(X.dot(t))
        return float((w * (y * lr - rate)).sum() - 0.5 * t.dot(t))

    f = objective(th)
    for _ in range(max_iter):
        g, h = _derivs(X.dot(th), y)
        grad = (w * g).dot(X) - th
        negH = np.eye(D) - (X * (w * h)[:, None]).T.dot(X)
        # the log-likelihood is not concave in s everywhere (y log rate is, -rate is): keep the Newton matrix positive definite
        ev = np.linalg.eigvalsh(negH)
        if ev[0] < 1e-8 * max(ev[-1], 1.0):
            negH = negH + (1e-8 * max(ev[-1], 1.0) - ev[0]) * np.eye(D)
        step = np.linalg.solve(negH, grad)
        t = 1.0
        while True:
            cand = th + t * step
            fc = objective(cand)
            if fc >= f or t < 1e-10:
                break
            t *= 0.5
        th, f = cand, fc
        if np.abs(step).max() * t < tol:
            break
    _, h = _derivs(X.dot(th), y)
    negH = np.eye(D) - (X * (w * h)[:, None]).T.dot(X)
    return th, np.linalg.inv(negH)


def synthetic_rows(n, d, rs):
    """Counts y ~ Poisson(softplus(x.theta)) on x = [N(0, I_{d-1}), 1] (the shape of data/synth_poiss.npz and of
    model_poiss.py:21-24, any width): rows [x, y]."""
    X = np.hstack((rs.randn(n, d - 1), np.ones((n, 1))))
    theta = np.hstack((np.ones(d - 1) / np.sqrt(max(d - 1, 1)), 0.0))
    y = rs.poisson(np.log1p(np.exp(X.dot(theta)))).astype(np.float64)
    return np.hstack((X, y[:, None]))


def standardized(X):
    """Covariates whitened, last column (the intercept) untouched: the preprocessing of load_data (model_poiss.py:4-19,
    model_lr.py:3-12) for data sets given as files."""
    X = np.array(X, dtype=np.float64)
    m = X[:, :-1].mean(axis=0)
    V = np.atleast_2d(np.cov(X[:, :-1], rowvar=False)) + 1e-12 * np.eye(X.shape[1] - 1)
    X[:, :-1] = np.linalg.solve(np.linalg.cholesky(V), (X[:, :-1] - m).T).T
    return X
