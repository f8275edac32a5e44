import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG_DIR = os.path.join(ROOT, "bayesian-coresets_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)


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
