import faulthandler
import os
import sys
import time

# RCCL / cross-process device-memory sharing needs dmabuf IPC on this host driver (hipIpcGetMemHandle fails in legacy
# mode): force it before torch / HIP load, whatever the launching environment exported.  The round-1 driver run had
# legacy mode and died in it (profiles/README.md, "round-1 GPUTEST abort"); BCX_KEEP_IPC_MODE=1 keeps the caller's
# value for crash hunts.
if not os.environ.get("BCX_KEEP_IPC_MODE"):
    os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"     # (same rule as bayesiancoresets_amd._native.ensure_ipc_mode, before anything loads)
if not os.environ.get("BCX_NO_FAULTHANDLER"):   # (crash hunts preload tools/probe/abort_trace.so instead)
    faulthandler.enable(all_threads=True)

# the library reads its development / test switches (BCX_OPT_LH, BCX_PROJ_NCT, BCX_INGEST_SCALAR ... : forced code
# paths some tests compare against each other) only under BCX_DEV=1 (csrc/dev_util.h bcx_dev_env)
os.environ.setdefault("BCX_DEV", "1")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "bayesian-coresets_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


_MARK_DIR = os.path.join(ROOT, "gpurun_out")
_MARK_FILE = os.path.join(_MARK_DIR, "current_test.txt")


def _mark(line):
    """Crash attribution: the nodeid is on disk (fsync'd) before the test body runs, so a HIP
    runtime abort that kills the interpreter still leaves the name of the test it died in."""
    try:
        os.makedirs(_MARK_DIR, exist_ok=True)
        with open(_MARK_FILE, "a") as f:
            f.write(line + "\n")
            f.flush()
            os.fsync(f.fileno())
    except OSError:
        pass


def pytest_runtest_logstart(nodeid, location):
    _mark("%.3f START %s" % (time.time(), nodeid))


def pytest_runtest_logreport(report):
    if report.when == "call" or (report.when == "setup" and report.outcome != "passed"):
        _mark("%.3f %s %s (%.2fs)" % (time.time(), report.outcome.upper(), report.nodeid, report.duration))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU case")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "snnls_golden.npz")
    return np.load(path)


def sha256(x):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(x).tobytes()).hexdigest()


@pytest.fixture(scope="session")
def normal_inputs(golden):
    """Regenerate seeded inputs and check their digests against the fixture file."""
    cache = {}

    def get(seed, N, d, key=None):
        k = (seed, N, d)
        if k not in cache:
            X = np.random.RandomState(seed).randn(N, d)
            if key is not None:
                assert sha256(X) == str(golden[key]), "seeded input drifted from the golden digest"
            cache[k] = X
        return cache[k]

    return get
