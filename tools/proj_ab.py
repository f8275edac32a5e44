"""dev: A/B of the projection kernel between two builds of the library on the SAME box, launches interleaved
(box-to-box spread of these kernels is +-8 %: only an interleaved comparison says whether a change cost speed).
    python tools/proj_ab.py [--lib-a bayesian-coresets_amd/lib_r02/libbcx.so --lib-b bayesian-coresets_amd/lib/libbcx.so]"""
import argparse, ctypes as C, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--lib-a", default=os.path.join(ROOT, "bayesian-coresets_amd", "lib_r02", "libbcx.so"))
ap.add_argument("--lib-b", default=os.path.join(ROOT, "bayesian-coresets_amd", "lib", "libbcx.so"))
ap.add_argument("--rows", type=int, default=625000)
ap.add_argument("--dim", type=int, default=300)
ap.add_argument("--samples", type=int, default=256)
ap.add_argument("--reps", type=int, default=15)
a = ap.parse_args()
libs = [C.CDLL(a.lib_a), C.CDLL(a.lib_b)]
vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
common = [vp, i32, vp, i64, i64, i32, i32, vp, i32, i32, dbl]
for L in libs:
    L.bcx_project_colsum.argtypes = common + [vp, vp]
    L.bcx_project_write.argtypes = common + [vp, i64, vp]
    L.bcx_project_select.argtypes = common + [vp, dbl, vp, vp]
rs = np.random.RandomState(0)
S, D, N = a.samples, a.dim, a.rows
work = torch.empty(2048 * S, dtype=torch.float64, device="cuda")
col = torch.empty(S, dtype=torch.float64, device="cuda")
res = torch.empty(2, dtype=torch.float64, device="cuda")
out = torch.empty((N, S), dtype=torch.float64, device="cuda")
resid = torch.randn(S, dtype=torch.float64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for fam, fid in (("linreg", 2), ("logistic", 0), ("poisson", 1)):
    cols = D if fam == "logistic" else D + 1
    Z = torch.randn(N, cols + (cols % 2), dtype=torch.float64, device="cuda")[:, :cols]
    if fam == "poisson":
        Z[:, -1] = torch.poisson(torch.ones(N, dtype=torch.float64, device="cuda"))
    theta = torch.from_numpy(0.1 * rs.randn(S, D + (D % 2))).cuda()[:, :D]
    ycol = -1 if fam == "logistic" else cols - 1
    base = [stream, fid, Z.data_ptr(), N, Z.stride(0), D, ycol, theta.data_ptr(), S, theta.stride(0), 1.0]
    for mode in ("colsum", "select", "write"):
        def launch(L):
            if mode == "colsum":
                rc = L.bcx_project_colsum(*base, col.data_ptr(), work.data_ptr())
            elif mode == "select":
                rc = L.bcx_project_select(*base, resid.data_ptr(), float(resid.sum()), res.data_ptr(), work.data_ptr())
            else:
                rc = L.bcx_project_write(*base, out.data_ptr(), S, None)
            assert rc == 0, rc
        tot = [0.0, 0.0]
        for L in libs:
            launch(L); launch(L)
        torch.cuda.synchronize()
        for _ in range(a.reps):
            for i, L in enumerate(libs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); launch(L); e1.record(); torch.cuda.synchronize()
                tot[i] += e0.elapsed_time(e1)
        fl = 2.0 * N * D * S
        ms = [t / a.reps for t in tot]
        print("%-8s %-6s A %.3f ms = %5.1f TFLOP/s   B %.3f ms = %5.1f TFLOP/s   B/A time %.3f" %
              (fam, mode, ms[0], fl / ms[0] / 1e9, ms[1], fl / ms[1] / 1e9, ms[1] / ms[0]), flush=True)
