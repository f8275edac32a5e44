"""Synthetic stand-in for the linear-regression experiment's data (BASELINE.json configs[4]).

The reference experiment (examples/linear_regression/main.py:56-108) regresses log10 house prices on radial basis
functions of (lat, lon); its data file (prices2018.npy) is not distributed with the repository and there is no
network here, so the *observations* are synthetic: locations uniform on the unit square, a smooth price surface
plus Gaussian noise.  Everything after the data load follows the reference: log-price statistics, the basis
scales .2, .4, .8, 1.2, 1.6, 2. (``n_bases_per_scale`` each) and 100 (one, effectively a constant), centres
drawn from the data locations, the design matrix X[n, i] = exp(-|x_n - c_i|^2 / (2 s_i^2)), Z = [X, Y], and the
prior mu0 = mean * 1, Sig0 = (std^2 + mean^2) I, likelihood variance std^2.  With locations on the unit square
the wide bases (s >= 1.2) are nearly constant and nearly parallel -- the strongly collinear design matrix the
config is about.
"""
import numpy as np

BASIS_SCALES = (0.2, 0.4, 0.8, 1.2, 1.6, 2.0, 100.0)      # linear_regression/main.py:80


def synthetic_observations(n, rs):
    """n rows [x0, x1, log10 price]: the role of prices2018.npy after the log transform (main.py:65-73)."""
    loc = rs.rand(n, 2)
    surface = 5.3 + 0.35 * np.sin(3.0 * loc[:, 0]) * np.cos(2.0 * loc[:, 1]) + 0.25 * loc[:, 0] * loc[:, 1]
    return np.column_stack((loc, surface + 0.15 * rs.randn(n)))


def basis_layout(obs, n_bases_per_scale, rs):
    """(scales, centres) as in main.py:80-99: n_bases_per_scale centres per narrow scale + one wide basis,
    centres picked without replacement among the observed locations."""
    counts = [n_bases_per_scale] * (len(BASIS_SCALES) - 1) + [1]
    scales = np.concatenate([s * np.ones(c) for s, c in zip(BASIS_SCALES, counts)])
    centres = np.vstack([obs[rs.choice(obs.shape[0], size=c, replace=False), :2] for c in counts])
    return scales, centres


def design_rows(obs, scales, centres, out=None):
    """Z = [X, Y] with X[n, i] = exp(-|loc_n - centre_i|^2 / (2 scale_i^2)) (main.py:101-106)."""
    n, nb = obs.shape[0], scales.shape[0]
    Z = np.empty((n, nb + 1)) if out is None else out
    for i in range(nb):
        Z[:, i] = np.exp(-((obs[:, :2] - centres[i]) ** 2).sum(axis=1) / (2.0 * scales[i] ** 2))
    Z[:, nb] = obs[:, 2]
    return Z


def make_rbf_regression(n, n_bases_per_scale=50, seed=1):
    """Returns dict(Z, mu0, Sig0, sigsq, scales, centres, obs) for an n-point synthetic RBF regression."""
    rs = np.random.RandomState(seed)
    obs = synthetic_observations(n, rs)
    std, mean = obs[:, 2].std(), obs[:, 2].mean()
    scales, centres = basis_layout(obs, n_bases_per_scale, rs)
    d = scales.shape[0]
    return {"Z": design_rows(obs, scales, centres), "mu0": mean * np.ones(d), "Sig0": (std ** 2 + mean ** 2) * np.eye(d),
            "sigsq": float(std ** 2), "scales": scales, "centres": centres, "obs": obs}


def design_rows_device(torch, obs_dev, scales, centres):
    """The same design matrix built on the GPU from device-resident observations (N x 3 fp64): used at sizes where
    the host loop above would dominate a benchmark's set-up.  Same formula, fp64."""
    sc = torch.as_tensor(scales, dtype=torch.float64, device=obs_dev.device)
    ce = torch.as_tensor(centres, dtype=torch.float64, device=obs_dev.device)
    n, nb = obs_dev.shape[0], sc.shape[0]
    Z = torch.empty((n, nb + 1), dtype=torch.float64, device=obs_dev.device)
    step = 1 << 18
    for r in range(0, n, step):
        loc = obs_dev[r:r + step, :2]
        d2 = ((loc[:, None, :] - ce[None, :, :]) ** 2).sum(dim=2)
        Z[r:r + step, :nb] = torch.exp(-d2 / (2.0 * sc[None, :] ** 2))
    Z[:, nb] = obs_dev[:, 2]
    return Z
