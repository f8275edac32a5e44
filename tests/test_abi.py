"""CPU-only: the C-ABI library loads and exports every symbol include/bcx.h declares; the
product path fails loudly (no CPU fallback) when no GPU is usable."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "bcx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bcx_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported():
    from bayesiancoresets_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _native.load()
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libbcx.so does not export " + name
    assert sorted(_native.SYMBOLS) == declared, "python binding list out of sync with include/bcx.h"
    assert lib.bcx_version().decode().startswith("bcx ")


def test_config_struct_layout():
    """ctypes mirror of bcx_config matches the C layout (8 int32 then 3 int64 = 56 bytes)."""
    import ctypes
    from bayesiancoresets_amd import _native
    assert ctypes.sizeof(_native.Config) == 56
    assert _native.Config.n_local.offset == 32


def test_row_length_limit_matches_header():
    from bayesiancoresets_amd import _native
    text = open(os.path.join(ROOT, "include", "bcx.h")).read()
    assert int(re.search(r"#define\s+BCX_MAX_ROW_LENGTH\s+(\d+)", text).group(1)) == _native.MAX_ROW_LENGTH
    assert int(re.search(r"#define\s+BCX_LOAD_CENTER_ROWS\s+(\d+)", text).group(1)) == _native.LOAD_CENTER_ROWS


def test_no_cpu_fallback():
    """Without a GPU the solver constructors raise; they never fall back to host arithmetic."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import bayesiancoresets_amd as bc
    from bayesiancoresets_amd import _native
    X = np.random.RandomState(0).randn(32, 4)
    for cls in (bc.snnls.GIGA, bc.snnls.FrankWolfe, bc.snnls.OrthoPursuit):
        with pytest.raises(_native.EngineError):
            cls(X.T, X.sum(axis=0))


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under bayesian-coresets_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "bayesian-coresets_amd")
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|snnls_oracle|oracle/_ref", re.M)
    for dp, dn, fn in os.walk(pkg):
        for f in fn:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert not pat.search(src), os.path.join(dp, f)
