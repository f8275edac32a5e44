"""Logistic regression with a standard-normal prior: the model of the reference's simple_lr / logistic regression
experiments (examples/common/model_lr.py:25-79 there; rows are z = y * x with y in {-1, +1}).

    log p(z | theta) = -log(1 + exp(-z.theta))            (linear tail -(-z.theta) once -z.theta >= 100, model_lr.py:29-31)
    log p(theta)     = -D/2 log(2 pi) - |theta|^2 / 2      (model_lr.py:33-35)

``laplace_fit`` is the tangent-space location of simple_lr/main.py:57-63: the MAP of the log joint and the inverse
negative Hessian there.  The reference finds the MAP with SciPy's BFGS from ``Z.mean(axis=0)``; here it is Newton's
method on the same concave objective (unique maximiser, so the same point), which needs only the D x D Hessian
sums and therefore runs on data that already lives on the GPU (``device=``: torch is plumbing for those sums).
"""
import numpy as np


def log_likelihood(z, th):
    """N x S matrix of log-likelihoods (NumPy): the callback form a host ``BlackBoxProjector`` takes."""
    arg = -np.atleast_2d(z).dot(np.atleast_2d(th).T)
    out = -arg                                                  # linear tail
    small = arg < 100
    out[small] = -np.log1p(np.exp(arg[small]))
    return out


def _matvec(xp, Z, th):
    # row-wise dot products as an elementwise product + sum: D is ~10, and this keeps the device path free of any
    # BLAS call (a 1M x 10 GEMV / GEMM through a vendor library costs more in set-up than the arithmetic is worth)
    return (Z * th).sum(1)


def _sums(xp, Z, th, wts):
    """sum_n w_n s_n z_n and sum_n w_n s_n (1 - s_n) z_n z_n^T with s = sigmoid(-z.theta) (0 curvature on the linear tail)."""
    arg = -_matvec(xp, Z, th)
    e = xp.exp(xp.clip(arg, None, 100.0)) if xp is np else xp.exp(xp.clamp(arg, max=100.0))
    s = xp.where(arg < 100, e / (1.0 + e), xp.ones_like(arg))
    c = xp.where(arg < 100, e / (1.0 + e) ** 2, xp.zeros_like(arg))
    if wts is not None:
        s, c = s * wts, c * wts
    D = Z.shape[1]
    H = xp.zeros((D, D), dtype=Z.dtype) if xp is np else xp.zeros((D, D), dtype=Z.dtype, device=Z.device)
    cz = Z * c[:, None]
    for j in range(D):                                          # D column sums of N-vectors
        H[j] = (cz * Z[:, j:j + 1]).sum(0)
    return (s[:, None] * Z).sum(0), H


def laplace_fit(Z, wts=None, device=None, tol=1e-10, max_iter=100, allreduce=None):
    """(mu, cov): MAP of the (weighted) log joint and the covariance of the Laplace approximation there.
    ``Z`` may be an ndarray or, with ``device`` / as a CUDA tensor, device-resident rows.  ``allreduce`` (row-sharded
    data, tensors only): a function that sums a tensor over all ranks in place and returns it -- every data sum goes
    through it, so every rank walks the same Newton path."""
    ar = allreduce if allreduce is not None else (lambda t: t)
    xp, eye, th = np, None, None
    is_tensor = hasattr(Z, "is_cuda")
    if device is not None or is_tensor:
        import torch
        xp = torch
        Z = Z if is_tensor else torch.as_tensor(np.asarray(Z, dtype=np.float64), device=device)
        wts = None if wts is None else torch.as_tensor(np.asarray(wts, dtype=np.float64), device=Z.device)
        eye = torch.eye(Z.shape[1], dtype=torch.float64, device=Z.device)
        cnt = ar(torch.tensor([float(Z.shape[0])], dtype=torch.float64, device=Z.device)) if wts is None else ar(wts.sum()[None])
        th = ar(Z.sum(0) if wts is None else (Z * wts[:, None]).sum(0)) / cnt
    else:
        Z = np.asarray(Z, dtype=np.float64)
        eye = np.eye(Z.shape[1])
        th = Z.mean(axis=0)                                     # simple_lr/main.py:59 starts there too
    if xp is np:
        solve = np.linalg.solve
    else:   # the D x D Newton system is solved on the host (D ~ 10): only the N-sized sums run on the device
        solve = lambda A, b: xp.as_tensor(np.linalg.solve(A.cpu().numpy(), b.cpu().numpy()), device=Z.device)

    def objective(t):
        arg = -_matvec(xp, Z, t)
        ll = xp.where(arg < 100, -xp.log1p(xp.exp(xp.clip(arg, None, 100.0) if xp is np else xp.clamp(arg, max=100.0))), -arg)
        if wts is not None:
            ll = ll * wts
        tot = ll.sum()
        if allreduce is not None:
            tot = ar(tot[None])[0]
        return float(tot - 0.5 * (t * t).sum())

    f = objective(th)
    for _ in range(max_iter):
        g1, H1 = _sums(xp, Z, th, wts)
        g1, H1 = ar(g1), ar(H1)
        grad, negH = g1 - th, H1 + eye                          # model_lr.py:57-62,73-79
        step = solve(negH, grad)
        t = 1.0
        while True:                                             # damped Newton: the objective must not decrease
            cand = th + t * step
            fc = objective(cand)
            if fc >= f or t < 1e-8:
                break
            t *= 0.5
        th, f = cand, fc
        if float(abs(step).max()) * t < tol:
            break
    _, H1 = _sums(xp, Z, th, wts)
    H1 = ar(H1)
    if xp is not np:
        return th.cpu().numpy(), np.linalg.inv((H1 + eye).cpu().numpy())
    return th, np.linalg.inv(H1 + eye)


def synthetic_rows(n, d, rs):
    """simple_lr/main.py:22-35: x ~ N(0, I_d), theta = 3 * 1, y ~ Bernoulli(sigmoid(x.theta)) in {-1, +1}; rows y * x."""
    X = rs.randn(n, d)
    ps = 1.0 / (1.0 + np.exp(-(X * 3.0).sum(axis=1)))
    y = (rs.rand(n) <= ps).astype(int)
    y[y == 0] = -1
    return y[:, None] * X
