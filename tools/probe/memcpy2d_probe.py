"""Round-1 GPUTEST abort hunt: pinned (hipHostRegister) host rows -> hipMemcpy2DAsync with mismatched pitches
(d = 17 doubles: 136 B source pitch, 144 B destination pitch), over many sub-page source offsets."""
import ctypes, sys
import numpy as np
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
vp, sz = ctypes.c_void_p, ctypes.c_size_t
rows, d, ld64 = 120000, 17, 18
span = ((rows - 1) * d + d) * 8
dev = vp()
assert hip.hipMalloc(ctypes.byref(dev), sz(rows * ld64 * 8)) == 0
host = np.random.RandomState(0).randn(rows * d + 1024)
back = np.empty(rows * ld64)
step = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for off in range(0, 512, step):          # element offsets: byte offsets 0..4088 within a page
    src = host.ctypes.data + off * 8
    rc = hip.hipHostRegister(vp(src), sz(span), 0)
    if rc != 0:
        print("register failed", off, rc, flush=True); continue
    rc = hip.hipMemcpy2DAsync(dev, sz(ld64 * 8), vp(src), sz(d * 8), sz(d * 8), sz(rows), 1, None)
    rc2 = hip.hipStreamSynchronize(None)
    rc3 = hip.hipHostUnregister(vp(src))
    if rc or rc2 or rc3:
        print("off", off, "rc", rc, rc2, rc3, flush=True)
    hip.hipMemcpy(vp(back.ctypes.data), dev, sz(rows * ld64 * 8), 2)
    got = back.reshape(rows, ld64)[:, :d]
    want = host[off:off + rows * d].reshape(rows, d)
    if not np.array_equal(got, want):
        print("MISMATCH at offset", off, flush=True)
print("probe done", flush=True)
