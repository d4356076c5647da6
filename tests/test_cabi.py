"""The C-ABI shared library loads and exports every symbol include/mphsir.h declares (no compute)."""
import ctypes
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "mp-hsir_amd", "libmphsir.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        sys.path.insert(0, os.path.join(ROOT, "mp-hsir_amd"))
        import build
        build.build(verbose=False)
    return ctypes.CDLL(LIB)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mphsir.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mphsir_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), "libmphsir.so does not export %s" % n


def test_python_binding_covers_the_header():
    import mp_hsir_amd._lib as L
    assert sorted(L.symbols()) == declared_symbols()


def test_version_and_error_text(lib):
    lib.mphsir_version.restype = ctypes.c_char_p
    lib.mphsir_last_error.restype = ctypes.c_char_p
    assert lib.mphsir_version().decode().endswith("gfx950")
    assert lib.mphsir_gemm_tok(None, 0, None) == -1            # EINVAL, nothing launched
    assert b"null pointer" in lib.mphsir_last_error()


def test_library_contains_gfx950_code_objects():
    blob = open(LIB, "rb").read()
    assert b"gfx950" in blob and b"gfx90a" not in blob and b"gfx942" not in blob


def test_ops_refuse_cpu_tensors_without_fallback(lib):
    """Product path must fail loudly: no CPU fallback, no silent oracle routing."""
    import torch
    import mp_hsir_amd._lib as L
    from mp_hsir_amd import ops
    saved = (L._lib, L._is_emu)
    try:
        L._lib, L._is_emu = None, False
        L.load()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ops.gemm_tok(torch.zeros(64, 32), torch.zeros(16, 32))
        with pytest.raises(RuntimeError, match="not found"):
            L.load("/nonexistent/libmphsir.so")
    finally:
        L._lib, L._is_emu = saved
