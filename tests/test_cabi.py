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


# ---- the three descriptions of every args struct agree: include/mphsir.h, the ctypes mirror, INTEGRATION.md ------------------------
_CTYPE_OF = {"uint32_t": ctypes.c_uint32, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "int": ctypes.c_int}


def header_structs():
    """{struct name: [(field, ctypes type)]} of every `typedef struct mphsir_* {...}` in include/mphsir.h, in declaration order"""
    text = open(os.path.join(ROOT, "include", "mphsir.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for name, body in re.findall(r"typedef struct (mphsir_\w+) \{(.*?)\}\s*\1;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            if "*" in decl:                                     # one or more pointers of one type: `const float* a` (one per declaration)
                m = re.match(r"^(?:const )?\w+\s*\*\s*(\w+)$", decl)
                assert m, "cannot parse %r in %s" % (decl, name)
                fields.append((m.group(1), ctypes.c_void_p))
            else:
                ty, names = decl.split(" ", 1)
                for n in names.split(","):
                    fields.append((n.strip(), _CTYPE_OF[ty]))
        out[name] = fields
    return out


def test_header_structs_match_the_ctypes_mirrors():
    import mp_hsir_amd._lib as L
    hs = header_structs()
    mirrors = {c.__doc__.split("struct ")[1].split()[0]: c for c in vars(L).values()
               if isinstance(c, type) and issubclass(c, ctypes.Structure) and c.__doc__ and "mirror of struct" in c.__doc__}
    assert sorted(mirrors) == sorted(hs), "a struct of the header has no ctypes mirror (or the other way round)"
    assert sum(n.endswith("_args") for n in hs) >= 14
    for name, fields in hs.items():
        got = [(n, t) for n, t in mirrors[name]._fields_]
        want = [(n, t) for n, t in fields]
        assert [n for n, _ in got] == [n for n, _ in want], "%s: field names / order differ: %s vs %s" % (name, got, want)
        for (n, tg), (_, tw) in zip(got, want):
            assert ctypes.sizeof(tg) == ctypes.sizeof(tw), "%s.%s: %s vs %s" % (name, n, tg, tw)
        if name.endswith("_args"):
            assert fields[0] == ("struct_size", ctypes.c_uint32), "%s must start with struct_size" % name
            assert mirrors[name]().struct_size == ctypes.sizeof(mirrors[name])


def test_integration_md_structs_match_the_header():
    """The binding INTEGRATION.md shows a maintainer is executable ctypes code: its Structure classes must be the header's structs
    (round 5 shipped a doc whose MlpArgs ended 13 fields early)."""
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    hs = header_structs()
    found = 0
    for cls, comment, body in re.findall(r"class (\w+)\(ctypes\.Structure\):\s*#\s*struct (mphsir_\w+)[^\n]*\n(\s+_fields_ = \[.*?\])[ \t]*(?:#[^\n]*)?\n", md, flags=re.S):
        ns = {"ctypes": ctypes}
        exec("class %s(ctypes.Structure):\n%s" % (cls, body), ns)
        got = ns[cls]._fields_
        want = hs[comment]
        assert [n for n, _ in got] == [n for n, _ in want], "INTEGRATION.md %s != %s" % (cls, comment)
        assert all(ctypes.sizeof(a[1]) == ctypes.sizeof(b[1]) for a, b in zip(got, want))
        found += 1
    assert found >= 1


def test_a_short_args_struct_is_refused(lib):
    """struct_size: a caller compiled against an older, shorter struct gets EINVAL and a message, nothing is read past its end"""
    import mp_hsir_amd._lib as L
    lib.mphsir_last_error.restype = ctypes.c_char_p
    a = L.MlpArgs()
    assert a.struct_size == ctypes.sizeof(L.MlpArgs)
    a.struct_size -= 8
    assert lib.mphsir_gated_mlp_fwd(ctypes.byref(a), 1, None) == -1
    assert b"struct_size" in lib.mphsir_last_error()
    g = L.GemmArgs()
    g.struct_size = 0
    assert lib.mphsir_gemm_tok(ctypes.byref(g), 1, None) == -1 and b"struct_size" in lib.mphsir_last_error()
