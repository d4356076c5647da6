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


def test_workspace_size_queries_match_the_python_allocations(lib):
    """mphsir_*_workspace_bytes (the caller owns every buffer and asks the library for its size) agree with what
    mp-hsir_amd/ops.py allocates for the same shapes (pure host arithmetic: no GPU needed)."""
    for f in ("mphsir_gemm_tn_workspace_bytes", "mphsir_dwconv_gram_workspace_bytes", "mphsir_pg_gate_bwd_workspace_bytes",
              "mphsir_win_attn_bwd_workspace_bytes"):
        getattr(lib, f).restype = ctypes.c_int64
    assert lib.mphsir_gemm_tn_workspace_bytes(704, 128, 42, 1, 1) == 42 * (704 * 128 + 704) * 4
    assert lib.mphsir_dwconv_gram_workspace_bytes(32, 16, 128, 2) == 32 * 16 * (2 * 64 * 64 + 256) * 4
    kl, kr = ctypes.c_int32(0), ctypes.c_int32(0)
    n = lib.mphsir_pg_gate_bwd_workspace_bytes(2048, 128, 16, 1, ctypes.byref(kl), ctypes.byref(kr))
    assert (kl.value, kr.value) == (464, 216) and n == 2048 * (464 + 216) * 2          # ops.pg_gate_bwd: round_up(C+5r+256, 8), round_up(5r+1+C, 8)
    M = 32 * 64 * 64
    assert lib.mphsir_win_attn_bwd_workspace_bytes(32, 64, 64, 128, 2, 1) == M * 5 * 128 * 2 + (M // 64) * 225 * 2 * 4
    assert lib.mphsir_gemm_tn_workspace_bytes(0, 1, 1, 1, 0) < 0
