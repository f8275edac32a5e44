"""dev: time optimize() (fp64-MFMA Gram + cold-start NNLS) after a greedy build."""
import sys, os, time
os.environ.setdefault("BCX_DEV", "1")   # BCX_OPT_COLD=1 (never start warm) is a dev switch (csrc/dev_util.h)
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
from bayesiancoresets_amd import _native as nat

def run(alg, N, d, its):
    eng = nat.Engine(alg, N, d)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.randn(N, d, device="cuda", dtype=torch.float64, generator=g)
    eng.load_device_rows(x.data_ptr(), N, d, True); torch.cuda.synchronize()
    assert eng.finalize(None) == 0
    eng.run_build(its, 1e-12)
    idx, w = eng.sparse_weights()
    e0 = eng.error()
    t0 = time.perf_counter()
    ok = eng.optimize(1e-12)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    k = len(idx)
    import ctypes
    st = (ctypes.c_longlong * 32)()
    eng.lib.bcx_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    eng.lib.bcx_debug_stamps(eng.h, st)
    print("   last optimize_lh launch: warm %d p0 %d after-inner %d entered %d left %d p %d fallback %d newton-step/weights %.2e nonpositive %d dual passes %d candidates left %d"
          % (st[20], st[21], st[22], st[23], st[24], st[25], st[26], st[27] / 1e15, st[28], st[29], st[30]))
    if os.environ.get("OPT_PROFILE"):     # library built with EXTRA=-DBCX_OPT_PROFILE: ticks of 10 ns by phase, workgroup 0
        names = ["dual pass", "pick + gather g", "u = H g", "barrier (u)", "fetch u, Schur, border update", "inner loop search / copies",
                 "publish rows", "barrier (leave)", "fetch rows, moved row, z", "downdate"]
        vals = [st[i] for i in range(8)] + [st[19], st[31]]
        print("   phases, ms: " + "; ".join("%s %.2f" % (nm, v * 1e-5) for nm, v in zip(names, vals)) + "; sum %.2f" % (sum(vals) * 1e-5))
    if os.environ.get("OPT_PROFILE") and k >= 1024:     # per workgroup: wait at the barrier after u = H g, own phases, wait at the leave's barrier
        buf = (ctypes.c_double * 768)()
        eng.lib.bcx_debug_wbak.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        if eng.lib.bcx_debug_wbak(eng.h, buf, 768) == 0:
            nwg = int(os.environ.get("BCX_OPT_WGS", "128"))
            w = np.array(buf[1:nwg]) * 1e-5; c = np.array(buf[257:256 + nwg]) * 1e-5; l = np.array(buf[513:512 + nwg]) * 1e-5
            print("   per workgroup 1..%d, ms: wait(u) min %.2f median %.2f max %.2f | own phases min %.2f median %.2f max %.2f | wait(leave) min %.2f median %.2f max %.2f"
                  % (nwg - 1, w.min(), np.median(w), w.max(), c.min(), np.median(c), c.max(), l.min(), np.median(l), l.max()))
            by_xcd = [("%.2f/%.2f" % (np.mean(w[np.arange(1, nwg)[:len(w)] % 8 == x]), np.mean(c[np.arange(1, nwg)[:len(c)] % 8 == x]))) for x in range(8)]
            print("   by workgroup index mod 8 (XCD), mean wait(u) / own phases: " + " ".join(by_xcd))
            order = np.argsort(w)
            print("   shortest waits (the stragglers): workgroups " + " ".join("%d:%.2f" % (order[i] + 1, w[order[i]]) for i in range(8)))
    if not st[20]:
        print("   the warm launch before it:  p0 %d after-inner %d entered %d left %d p %d fallback %d newton-step/weights %.2e nonpositive %d dual passes %d candidates left %d"
              % (st[9], st[10], st[11], st[12], st[13], st[14], st[15] / 1e15, st[16], st[17], st[18]))
    print("alg %d N=%d d=%d k=%d: optimize %.2f ms accepted=%s err %.6g -> %.10g  (Gram flops %.2e; refined-solve fallbacks %d)"
          % (alg, N, d, k, dt * 1e3, ok, e0, eng.error(), 2.0 * k * k * d, eng.omp_stats()["resolves"]))

if __name__ == "__main__":
    print("BCX_OPT_COLD =", os.environ.get("BCX_OPT_COLD"))
    if len(sys.argv) > 1:      # alg,N,d,its
        a, N, d, its = (int(v) for v in sys.argv[1].split(","))
        run(a, N, d, its)
        sys.exit(0)
    run(nat.ALG_FW, 1000000, 512, 400)
    run(nat.ALG_GIGA, 1000000, 512, 1000)
    run(nat.ALG_FW, 1000000, 1024, 1500)
    run(nat.ALG_FW, 1000000, 2048, 1500)
