"""dev: host -> device ingest rate of bcx_load_rows (pinned bounce buffers + DMA + ingest kernel), rows resident in host memory.
    python tools/upload_bench.py [rows] [d]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bayesian-coresets_amd"))
import torch
from bayesiancoresets_amd import _native as nat

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256
rs = np.random.RandomState(0)
for dt in (np.float64, np.float32):
    X = rs.standard_normal((N, d)).astype(dt)
    for keep in (True, False):
        if dt == np.float32 and keep:
            continue
        e = nat.Engine(nat.ALG_FW, N, d, keep_exact_rows=keep)
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.load_rows_any(X)
            torch.cuda.synchronize()
            dt_s = time.perf_counter() - t0
            print("rows %d d %d src %s keep_exact %s: %.1f ms  %.1f GB/s" % (N, d, X.dtype, keep, dt_s * 1e3, X.nbytes / dt_s / 1e9), flush=True)
        e.close()
