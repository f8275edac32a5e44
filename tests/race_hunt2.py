"""by hand, library built with EXTRA=-DBCX_DEBUG_GRID: do the 16 workgroups of optimize_grid_kernel agree?"""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd")); sys.path.insert(0, ROOT)
import bayesiancoresets_amd as bc
N, d, itrs = 30000, 100, 90
X = np.random.RandomState(N + d).randn(N, d); b = X.sum(axis=0)
ref = None
import hashlib
href = None
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 1000):
    s = bc.snnls.GIGA(X.T, b); s.build(itrs)
    k = len(s._eng.sparse_weights()[0])
    s.optimize()
    buf = (C.c_double * 256)()
    lib = s._eng.lib; lib.bcx_debug_wbak.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    assert lib.bcx_debug_wbak(s._eng.h, buf, 256) == 0
    dig = np.array(list(buf))[k + 8:k + 8 + 64].reshape(16, 4)
    agree = all(np.array_equal(dig[0], dig[i]) for i in range(16))
    key = (dig[0][0], dig[0][1], dig[0][2], dig[0][3])
    w = s.weights(); e = s.error()
    h = hashlib.md5(w.tobytes() + np.float64(e).tobytes()).hexdigest()
    if ref is None: ref = key; href = h; wref = w.copy(); eref = e
    if h != href:
        dw = np.abs(w - wref); print('run', r, 'RESULT differs: err', repr(e), 'vs', repr(eref), 'max |dw|', dw.max(), 'n differing', (dw > 0).sum(), 'digest equal', key == ref, 'agree', agree, 'limit', s.reached_numeric_limit)
    if not agree or key != ref:
        print("run", r, "agree", agree, "wg0", dig[0], "differs from first run:", key != ref)
        if not agree:
            for i in range(16): print("   wg", i, dig[i])
print("done; reference digest", ref, "k", k)
