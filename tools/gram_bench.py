"""dev: the Gram operator of the dense re-weight (bcx_gram: csrc/gram.hip gram_sk_kernel; BCX_GRAM_NCT=-1: csrc/moments.hip
gram_tile_kernel + moments_reduce_kernel; BCX_GRAM_NCT=4 / 8 forces the tile width) --
kernel time by hipEvents over `reps` back-to-back calls, upper-triangle flops k (k + 1) d over that time against the fp64 MFMA
peak; beside it round 2's direct kernel through optimize() is selected with BCX_GRAM_DIRECT=1 (tools/optimize_bench.py).
    python tools/gram_bench.py [k,d ...]"""
import os, sys
os.environ.setdefault("BCX_DEV", "1")   # dev switches are read only under this gate (csrc/dev_util.h)
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
from bayesiancoresets_amd import _native as nat

PEAK = 78.6
lib = nat.load()
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(400, 512), (999, 512), (1497, 1024), (1024, 1024), (2048, 2048), (4096, 1024), (512, 4096), (8192, 1024)]
for k, d in shapes:
    V = torch.randn(k, d, dtype=torch.float64, device="cuda")
    G = torch.empty(k, k, dtype=torch.float64, device="cuda")
    need = int(lib.bcx_gram_scratch_bytes(k, d))
    work = torch.empty((need + 7) // 8, dtype=torch.float64, device="cuda")
    st = int(torch.cuda.current_stream().cuda_stream)
    call = lambda: lib.bcx_gram(st, V.data_ptr(), k, d, d, G.data_ptr(), k, work.data_ptr(), work.numel() * 8)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    reps = 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ref = V @ V.T
    err = float((G - ref).abs().max() / ref.abs().max())
    upper = float(k) * (k + 1) * d          # 2 flops x k (k + 1) / 2 entries x d
    blocks = ((k + 63) // 64) * ((k + 63) // 64 + 1) // 2 * 4096 * 2.0 * d     # what the 64 x 64 block pairs execute
    print("k=%d d=%d: %.1f us per call (both kernels)  %.1f TFLOP/s on the upper triangle = %.2f of %.1f  (%.1f TFLOP/s executed incl. block padding)  max rel err %.1e  scratch %.1f MB"
          % (k, d, us, upper / us / 1e6, upper / us / 1e6 / PEAK, PEAK, blocks / us / 1e6, err, need / 1e6), flush=True)
