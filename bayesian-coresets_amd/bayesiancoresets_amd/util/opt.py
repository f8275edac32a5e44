"""Projected ADAM on the non-negative orthant (reference: bayesiancoresets/util/opt.py:4-28).
O(k) host vector math used by the variational coresets; not on the device hot path."""
import sys

import numpy as np


def nn_opt(x0, grd, nn_idcs=None, opt_itrs=1000, step_sched=lambda i: 1.0 / (i + 1), b1=0.9, b2=0.999,
           eps=1e-8, verbose=False):
    x = x0.copy()
    mom1 = np.zeros(x.shape[0])
    mom2 = np.zeros(x.shape[0])
    clamp_all = nn_idcs is None
    for i in range(opt_itrs):
        g = grd(x)
        if verbose:
            at_bound = np.intersect1d(nn_idcs, np.where(x == 0)[0])
            free = np.setdiff1d(np.arange(x.shape[0]), at_bound)
            sys.stdout.write("itr %d/%d: ||inactive constraint grads|| = %s                \r"
                             % (i + 1, opt_itrs, np.sqrt((g[free] ** 2).sum())))
            sys.stdout.flush()
        mom1 = b1 * mom1 + (1.0 - b1) * g
        mom2 = b2 * mom2 + (1.0 - b2) * g ** 2
        step = step_sched(i) * mom1 / (1.0 - b1 ** (i + 1)) / (eps + np.sqrt(mom2 / (1.0 - b2 ** (i + 1))))
        x -= step
        if clamp_all:
            x = np.maximum(x, 0.0)
        else:
            x[nn_idcs] = np.maximum(x[nn_idcs], 0.0)
    if verbose:
        sys.stdout.write("\n")
        sys.stdout.flush()
    return x
