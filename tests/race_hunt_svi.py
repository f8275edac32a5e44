"""Race hunt (dev, GPU): SparseVI's enqueued ADAM loop repeated on the same inputs and the same normal numbers must give the
same weights bit for bit (every kernel of the step sums in a fixed order).  python tests/race_hunt_svi.py [repeats]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bayesian-coresets_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import bayesiancoresets_amd as bc
from models import make_linreg_data
from test_gpu_svi import _ReplaySampler

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for D, N, S, T, steps, colsum in ((301, 200000, 256, 100, 4, "moments"), (24, 50000, 64, 40, 5, "mfma"), (57, 80000, 200, 60, 6, "moments")):
    Z = torch.from_numpy(make_linreg_data(3, N, D)).cuda()
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    noise = torch.randn(steps * (T + 1) + 2, S, D + D % 2, dtype=torch.float64, device="cuda", generator=g)
    seen = {}
    for r in range(reps):
        smp = _ReplaySampler(bc.LinregPosteriorSampler(np.zeros(D), 3.0 * np.eye(D), 1.0), noise)
        alg = bc.SparseVICoreset(Z, bc.DeviceProjector("linreg", smp, S, sigsq=1.0, colsum=colsum), opt_itrs=T)
        alg.build(steps)
        h = hashlib.sha256(alg.wts.tobytes() + alg.idcs.tobytes()).hexdigest()[:12]
        seen[h] = seen.get(h, 0) + 1
    print("D=%d N=%d S=%d opt_itrs=%d steps=%d colsum=%s x%d: %d distinct outcome(s) %s" % (D, N, S, T, steps, colsum, reps, len(seen), seen))
