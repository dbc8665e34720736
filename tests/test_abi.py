"""The C-ABI library loads here (no GPU) and exports every symbol the header declares."""
import ctypes
import os
import re

import pytest

from bayesic_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "bayesic_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(bsc_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from bayesic_amd.build import build
    build()
    return _ffi.load_library()


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "bsc_blr_data_pass" in syms and "bsc_ctx_create" in syms
    assert len(syms) >= 20


def test_every_declared_symbol_is_exported_and_bound(lib):
    for name in declared_symbols():
        assert name in _ffi.SIGNATURES, "%s declared in the header but not bound in _ffi" % name
        assert getattr(lib, name) is not None


def test_every_bound_symbol_is_declared():
    assert sorted(_ffi.SIGNATURES) == declared_symbols()


def test_version_and_error_string_without_gpu(lib):
    assert lib.bsc_version() == 100
    # a null ctx is rejected with a message, not a crash (no GPU needed)
    rc = lib.bsc_ctx_sync(None)
    assert rc == -1
    assert b"null" in lib.bsc_last_error()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from bayesic_amd.device import Context
    with pytest.raises(_ffi.BayesicHipError):
        Context()
