"""test infrastructure (by hand): the cooperative D x D factorisation (csrc/lrpost.hip: chain / assistants / helpers handing
tiles to each other through write-through stores and relaxed flags) run many times on the same inputs, with a competing
kernel keeping the other CUs busy every other run -- any number of distinct outcomes other than 1 is a data race (a stale
line, a missed flag).     python tests/race_hunt_lrpost.py [reps] [D] [k]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
from bayesiancoresets_amd import _native

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 301
k = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lib = _native.load()
rs = np.random.RandomState(4)
ld, ldk = D + D % 2, (k + 31) // 32 * 32
XT = np.zeros((D, ldk))
XT[:, :k] = rs.rand(D, k)
d = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()
need = int(lib.bcx_linreg_posterior_factor_scratch_bytes(D))
work = torch.zeros(need // 8, dtype=torch.float64, device="cuda")
U, u = torch.zeros(D, ld, dtype=torch.float64, device="cuda"), torch.zeros(D, dtype=torch.float64, device="cuda")
X_d, y_d, S_d, r_d = d(XT), d(rs.randn(k)), d(np.eye(D) * 0.03), d(np.ones(D))
ws = [d(np.abs(rs.randn(k)) * 50) for _ in range(4)]
busy = torch.randn(4096, 4096, device="cuda")
st = int(torch.cuda.current_stream().cuda_stream)
side = torch.cuda.Stream()
seen = [{} for _ in ws]
for r in range(reps):
    i = r % len(ws)
    if r & 4:                                   # half of the runs beside a GEMM on another stream
        with torch.cuda.stream(side):
            busy2 = busy @ busy
    rc = lib.bcx_linreg_posterior_factor(st, k, D, ldk, ws[i].data_ptr(), X_d.data_ptr(), y_d.data_ptr(), S_d.data_ptr(), D, r_d.data_ptr(),
                                         0.02, work.data_ptr(), work.numel() * 8, U.data_ptr(), ld, u.data_ptr(), None)
    assert rc == 0
    assert lib.bcx_linreg_posterior_factor_status(st, D, work.data_ptr()) == 0, lib.bcx_project_last_error()
    h = hashlib.md5(U.cpu().numpy().tobytes() + u.cpu().numpy().tobytes()).hexdigest()
    seen[i].setdefault(h, []).append(r)
torch.cuda.synchronize()
print("lrpost D=%d k=%d: %d runs over %d weight vectors: distinct outcomes per weight vector %s%s"
      % (D, k, reps, len(ws), [len(s) for s in seen], "" if all(len(s) == 1 for s in seen) else "  <-- RACE " + str([[v[:4] for v in s.values()] for s in seen])))
