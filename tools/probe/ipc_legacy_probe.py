"""Which HIP host-memory calls survive HSA_ENABLE_IPC_MODE_LEGACY=1 on this driver?  (round-1 GPUTEST abort hunt)
usage: python tools/probe/ipc_legacy_probe.py register|hostmalloc|memcpy"""
import ctypes, sys, os
import numpy as np
hip = ctypes.CDLL("/opt/rocm/lib/libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
what = sys.argv[1]
print("env", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), what, flush=True)
n = 32 << 20
dev = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(n)) == 0
a = np.ones(n // 8)
if what == "register":
    rc = hip.hipHostRegister(ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), 0)
    print("hipHostRegister rc", rc, hip.hipGetErrorString(rc), flush=True)
    rc = hip.hipMemcpy(dev, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), 1)
    print("memcpy rc", rc, flush=True)
    if rc == 0: print("unregister", hip.hipHostUnregister(ctypes.c_void_p(a.ctypes.data)), flush=True)
elif what == "hostmalloc":
    p = ctypes.c_void_p()
    rc = hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(n), 0)
    print("hipHostMalloc rc", rc, flush=True)
    ctypes.memmove(p, a.ctypes.data, n)
    print("memcpy rc", hip.hipMemcpy(dev, p, ctypes.c_size_t(n), 1), flush=True)
else:
    print("memcpy rc", hip.hipMemcpy(dev, ctypes.c_void_p(a.ctypes.data), ctypes.c_size_t(n), 1), flush=True)
print("done", flush=True)
