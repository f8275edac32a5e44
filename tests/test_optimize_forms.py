"""GPU: optimize() (snnls.py:82-97) through the incremental Lawson-Hanson kernel (csrc/omp_lh.hip: optimize_lh_kernel) must give
the same weights whatever form its grid barriers take -- the default (no release / acquire fences, the last arriver publishes
the barrier's index), with the fences back (BCX_GRID_FENCE=1), with every workgroup polling the arrival counter (BCX_GRID_FLAT=1)
-- and whatever the workgroup count: the fences order nothing that is read (every cross-workgroup datum is a drained
write-through store read by sc1 loads), so a difference would be a hand-off that depended on them.  The library reads these
switches once per process: every form runs in its own interpreter.  Supports with k > d (hundreds of columns leave and enter
after the warm start) and k > 1024 (the 128-workgroup shape)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, os, hashlib
import numpy as np
sys.path.insert(0, os.path.join(sys.argv[1], "bayesian-coresets_amd"))
import bayesiancoresets_amd as bc
N, d, itrs = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
X = np.random.RandomState(N + d).randn(N, d)
s = bc.snnls.FrankWolfe(X.T, X.sum(axis=0), dtype="float32")
s.build(itrs)
e0 = s.error()
s.optimize()
w = s.weights()
print("RESULT", hashlib.md5(w.tobytes()).hexdigest(), int((w > 0).sum()), repr(float(e0)), repr(float(s.error())))
"""


def _child(env, N, d, itrs):
    e = dict(os.environ)
    e.update({"BCX_DEV": "1"})
    e.update(env)
    out = subprocess.run([sys.executable, "-c", _CHILD, ROOT, str(N), str(d), str(itrs)], capture_output=True, text=True, timeout=600,
                         env=e, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    return line[1], int(line[2]), float(line[3]), float(line[4])


@pytest.mark.parametrize("shape", [(20000, 1024, 1300), (8000, 256, 700)])
def test_barrier_forms_give_one_result(shape):
    N, d, itrs = shape
    base = _child({}, N, d, itrs)
    assert base[3] <= base[2] * (1.0 + 1e-12)                 # optimize() never raises the error (snnls.py:91-97)
    if itrs > d:
        assert base[1] <= d                                   # a vertex: at most d columns carry weight
    for env in ({"BCX_GRID_FENCE": "1"}, {"BCX_GRID_FLAT": "1"}, {"BCX_GRID_FENCE": "1", "BCX_GRID_FLAT": "1"}):
        other = _child(env, N, d, itrs)
        assert other[0] == base[0], (env, base, other)        # the same bits
    # another workgroup count changes the order of no sum (rows of H are whole-wave dot products, block sums are per workgroup)
    other = _child({"BCX_OPT_WGS": "64"}, N, d, itrs)
    assert other[0] == base[0], ("BCX_OPT_WGS=64", base, other)
