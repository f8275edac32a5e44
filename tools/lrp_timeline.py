#!/usr/bin/env python3
"""dev: per-step timeline of the cooperative D x D factorisation (csrc/lrpost.hip lrp_chol_kernel) from the wall-clock stamps
its workgroups leave under BCX_DEV=1 BCX_LRP_DBG=1: the chain's diag / publish / side-work / step times and how far behind
the helpers run.  Also times the two launches with events.
    BCX_DEV=1 BCX_LRP_DBG=1 python tools/lrp_timeline.py [--D 301 --k 300 --reps 20]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "bayesian-coresets_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--D", type=int, default=301)
    ap.add_argument("--k", type=int, default=300)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import torch
    from bayesiancoresets_amd import _native
    lib = _native.load()
    D, k = a.D, a.k
    rs = np.random.RandomState(1)
    ld = D + D % 2
    ldk = (k + 31) // 32 * 32
    XT = np.zeros((D, ldk))
    XT[:, :k] = rs.rand(D, k)
    w = np.abs(rs.randn(k)) * 50
    d = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()
    need = int(lib.bcx_linreg_posterior_factor_scratch_bytes(D))
    work = torch.zeros(need // 8, dtype=torch.float64, device="cuda")
    U, mu = torch.zeros(D, ld, dtype=torch.float64, device="cuda"), torch.zeros(D, dtype=torch.float64, device="cuda")
    w_d, X_d, y_d, S_d, r_d = d(w), d(XT), d(rs.randn(k)), d(np.eye(D) * 0.03), d(np.ones(D))
    st = int(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.bcx_linreg_posterior_factor(st, k, D, ldk, w_d.data_ptr(), X_d.data_ptr(), y_d.data_ptr(), S_d.data_ptr(), D, r_d.data_ptr(),
                                                   0.02, work.data_ptr(), work.numel() * 8, U.data_ptr(), ld, mu.data_ptr(), None)
    for _ in range(3):
        assert call() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    assert lib.bcx_linreg_posterior_factor_status(st, D, work.data_ptr()) == 0
    print("form + chol: %.1f us per call (events, %d calls back to back)" % (e0.elapsed_time(e1) * 1e3 / a.reps, a.reps))
    nt = (D + 31) // 32
    ntiles = 5 * nt * nt + (nt + 1) * nt + 3 * nt
    flag_bytes = (4 * nt + 1 + 15) // 16 * 16 * 4
    off = ntiles * 1024 + nt * 32 + flag_bytes // 8
    nwg = 63
    stamps = work[off:off + nwg * nt * 8].view(torch.int64).cpu().numpy().reshape(nwg, nt, 8).astype(np.float64) * 0.01      # us
    if not os.environ.get("BCX_LRP_DBG"):
        print("(no stamps: run with BCX_DEV=1 BCX_LRP_DBG=1)")
        return
    t0 = stamps[0, 0, 0]
    ch = stamps[0] - t0
    asst = stamps[1:3] - t0
    hp = stamps[3:] - t0
    hp[stamps[3:] == 0] = np.nan
    print("(all times us from the chain's first diag)")
    print("step  diag0   diag  +F1   side_ready side_done step_end | assistants ready (vs diag0) | helpers: start(max) F1seen(max) signalled(max) (vs F1)")
    for p in range(nt):
        c = ch[p]
        line = "%3d %7.2f %6.2f %5.2f" % (p, c[0], c[1] - c[0], c[2] - c[1])
        if p + 1 < nt:
            line += "  %9.2f %9.2f %8.2f" % (c[3] - c[0], c[4] - c[0], c[5] - c[0])
            line += " | %8.2f %8.2f       " % ((asst[0, p, 1] - c[0], asst[1, p, 1] - c[0]) if p >= 1 else (0.0, 0.0))
        else:
            line += "  %9s %9s %8s | %8s %8s       " % ("-", "-", "-", "-", "-")
        h = hp[:, p, :]
        line += " | %10.2f %11.2f %14.2f" % (np.nanmax(h[:, 0]) - c[2], np.nanmax(h[:, 1]) - c[2], np.nanmax(h[:, 2]) - c[2])
        print(line)
    print("chain total %.2f us; last helper signal %.2f us" % (ch[nt - 1, 2], np.nanmax(hp[:, nt - 1, 2])))
    allst = stamps[stamps > 0]
    print("first stamp of any workgroup %.2f us, last %.2f us (relative to the chain's first diag)" % (allst.min() - t0, allst.max() - t0))


if __name__ == "__main__":
    main()
