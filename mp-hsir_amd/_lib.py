"""ctypes binding of libmphsir.so (include/mphsir.h).  No CPU fallback: if the HIP library is
missing or does not load, every op raises -- build it with ``python mp-hsir_amd/build.py``.
"""
import ctypes
import os

import torch  # noqa: F401  -- must come first: libmphsir.so binds to the HIP runtime PyTorch-ROCm already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, os.environ.get("MPHSIR_LIB_AB", "libmphsir.so"))      # MPHSIR_LIB_AB: another in-tree build of the same sources, for A/B timing on one box
_lib = None
_is_emu = False

c_void_p, c_int64, c_int32, c_int, c_float_p = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_float))


class _Args(ctypes.Structure):
    """base of every mphsir_*_args mirror: the first member, struct_size, is filled in on construction (the library refuses a struct
    of another size: include/mphsir.h)"""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = ctypes.sizeof(self)


_SZ = [("struct_size", ctypes.c_uint32)]


class GemmArgs(_Args):
    """mirror of struct mphsir_gemm_args"""
    _fields_ = _SZ + [("X", c_void_p), ("ldx", c_int64), ("W", c_void_p), ("w_batch_stride", c_int64),
                ("rows_per_batch", c_int64), ("bias", c_void_p), ("ln_w", c_void_p), ("ln_b", c_void_p),
                ("Y", c_void_p), ("ldy", c_int64), ("M", c_int64), ("N", c_int64), ("K", c_int64),
                ("epi", c_int), ("R", c_void_p), ("ldr", c_int64), ("SA", c_void_p), ("ldsa", c_int64),
                ("gate", c_void_p), ("keep", c_void_p), ("H", c_int32), ("Wimg", c_int32), ("shift", c_int32), ("form", c_int32)]


class MlpArgs(_Args):
    """mirror of struct mphsir_mlp_args"""
    _fields_ = _SZ + [("X", c_void_p), ("ldx", c_int64), ("ln_w", c_void_p), ("ln_b", c_void_p), ("W1", c_void_p),
                ("b1", c_void_p), ("W2", c_void_p), ("b2", c_void_p), ("keep", c_void_p), ("rows_per_batch", c_int64),
                ("Y", c_void_p), ("ldy", c_int64), ("M", c_int64), ("C", c_int32), ("HP", c_int32), ("tiles_per_wave", c_int32),
                ("hsplit", c_int32), ("ypart", c_void_p), ("R", c_void_p), ("ldr", c_int64),
                ("PV", c_void_p), ("ldpv", c_int64), ("PM", c_void_p), ("pm_batch_stride", c_int64), ("PSA", c_void_p), ("ldpsa", c_int64),
                ("pgate", c_void_p), ("pkeep", c_void_p), ("Yb", c_void_p), ("ldyb", c_int64), ("H", c_int32), ("Wimg", c_int32), ("shift", c_int32)]


class WinAttnArgs(_Args):
    """mirror of struct mphsir_win_attn_args"""
    _fields_ = _SZ + [(n, c_void_p) for n in ("X", "ln_w", "ln_b", "Wqkv", "bqkv", "rpb", "Wproj", "bproj", "SA", "mu", "Oattn")] + \
               [(n, c_int32) for n in ("B", "H", "W", "C", "heads", "shift")]


class PgFwdArgs(_Args):
    """mirror of struct mphsir_pg_fwd_args"""
    _fields_ = _SZ + [(n, c_void_p) for n in ("mu", "Wprompt", "prompt_param", "Wq", "Wkv", "Wdown", "Wpproj", "bpproj", "Wup", "gate")] + \
               [(n, c_int32) for n in ("nW", "C", "r")]


class GramArgs(_Args):
    """mirror of struct mphsir_gram_args"""
    _fields_ = _SZ + [("Tq", c_void_p), ("ldq", c_int64), ("Tk", c_void_p), ("ldk", c_int64), ("Tv", c_void_p), ("ldv", c_int64),
                ("wq", c_void_p), ("wk", c_void_p), ("wv", c_void_p), ("ldw", c_int64), ("V", c_void_p), ("ldvo", c_int64),
                ("Gpart", c_void_p), ("Spart", c_void_p)] + [(n, c_int32) for n in ("B", "H", "W", "C", "heads", "nsplit")] + \
               [("QK", c_void_p), ("ldqk", c_int64)]


class FusedGramArgs(_Args):
    """mirror of struct mphsir_fused_gram_args"""
    _fields_ = _SZ + [("X", c_void_p), ("ldx", c_int64), ("ln_w", c_void_p), ("ln_b", c_void_p), ("Wqkv", c_void_p), ("w9", c_void_p),
                ("ldw", c_int64), ("V", c_void_p), ("ldvo", c_int64), ("Gpart", c_void_p), ("Spart", c_void_p)] + \
               [(n, c_int32) for n in ("B", "H", "W", "C", "heads", "nsplit", "head_groups")] + \
               [("T", c_void_p), ("ldt", c_int64), ("QK", c_void_p), ("ldqk", c_int64), ("row_segments", c_int32)]


class FoldArgs(_Args):
    """mirror of struct mphsir_fold_args"""
    _fields_ = _SZ + [("Gpart", c_void_p), ("Spart", c_void_p), ("temperature", c_void_p), ("Wo", c_void_p), ("M", c_void_p),
                ("MT", c_void_p)] + \
               [(n, c_int32) for n in ("B", "C", "heads", "nsplit")] + [("Gsum", c_void_p), ("Ssum", c_void_p)]


class GateArgs(_Args):
    """mirror of struct mphsir_gate_args"""
    _fields_ = _SZ + [("T", c_void_p), ("ldt", c_int64), ("w9", c_void_p), ("ldw", c_int64), ("U", c_void_p), ("ldu", c_int64)] + \
               [(n, c_int32) for n in ("B", "H", "W", "HP")]


class GdfnArgs(_Args):
    """mirror of struct mphsir_gdfn_args"""
    _fields_ = _SZ + [("X", c_void_p), ("ldx", c_int64), ("ln_w", c_void_p), ("ln_b", c_void_p), ("Win", c_void_p), ("w9", c_void_p),
                ("ldw", c_int64), ("Wout", c_void_p), ("Y", c_void_p), ("ldy", c_int64)] + \
               [(n, c_int32) for n in ("B", "H", "W", "D", "HP", "nsplit")] + [("T", c_void_p), ("ldt", c_int64)]


class MlpBwdArgs(_Args):
    """mirror of struct mphsir_mlp_bwd_args"""
    _fields_ = _SZ + [(n, c_void_p) for n in ("X", "dY", "DM", "ln_w", "ln_b", "W1", "b1", "W1T", "W2T", "dX", "XN", "H", "DPRE", "part")] + \
               [("M", c_int64), ("C", c_int32), ("HP", c_int32), ("variant", c_int32), ("keep", c_void_p), ("rows_per_batch", c_int64),
                ("hsplit", c_int32), ("dxn_part", c_void_p)]


class MlpWgradArgs(_Args):
    """mirror of struct mphsir_mlp_wgrad_args"""
    _fields_ = _SZ + [(n, c_void_p) for n in ("XN", "DM", "W1", "b1", "W2T", "dW1p", "dW2p", "db1p", "db2p")] + \
               [("M", c_int64), ("C", c_int32), ("HP", c_int32), ("ranges", c_int32), ("chunks_per_wg", c_int32)]


class WinAttnBwdArgs(_Args):
    """mirror of struct mphsir_win_attn_bwd_args"""
    _fields_ = _SZ + [(n, c_void_p) for n in ("X", "dSA", "dmu", "ln_w", "ln_b", "Wqkv", "bqkv", "rpb", "WprojT", "dQKV", "XNw",
                                         "dSAt", "drpb")] + [(n, c_int32) for n in ("B", "H", "W", "C", "heads", "shift", "head_split")]


class FoldBwdArgs(_Args):
    """mirror of struct mphsir_fold_bwd_args"""
    _fields_ = _SZ + [(n, c_void_p) for n in ("Gpart", "Spart", "temperature", "Wo", "dM", "W2", "dWo", "dtemp")] + \
               [(n, c_int32) for n in ("B", "C", "heads", "nsplit", "dM_nsplit")] + \
               [("DO", c_void_p), ("lddo", c_int64), ("V", c_void_p), ("ldv", c_int64), ("N", c_int32), ("dm_scale", c_void_p),
                ("w2_blocks", c_int32)]


class PgBwdArgs(_Args):
    """mirror of struct mphsir_pg_bwd_args"""
    _fields_ = _SZ + [(n, c_void_p) for n in ("mu", "dgate", "Wprompt", "prompt_param", "Wq", "Wkv", "Wdown", "Wpproj", "bpproj", "Wup",
                                         "dmu", "L", "R")] + [(n, c_int32) for n in ("nW", "C", "r", "KL", "KR", "lr_bf16")]


class SpectralBwdArgs(_Args):
    """mirror of struct mphsir_spectral_bwd_args"""
    _fields_ = _SZ + [("QK", c_void_p), ("ldqk", c_int64), ("DO", c_void_p), ("lddo", c_int64), ("T", c_void_p), ("ldt", c_int64),
                      ("W2", c_void_p), ("MbT", c_void_p), ("w9", c_void_p), ("ldw", c_int64), ("dT", c_void_p), ("lddt", c_int64),
                      ("part", c_void_p)] + [(n, c_int32) for n in ("B", "H", "W", "C", "heads", "nblk", "round_dall")] + [("vscale", c_void_p)]


class TnProblem(ctypes.Structure):
    """mirror of struct mphsir_gemm_tn_problem"""
    _fields_ = [("A", c_void_p), ("lda", c_int64), ("B", c_void_p), ("ldb", c_int64), ("Cpart", c_void_p), ("colsum_part", c_void_p),
                ("M", c_int64), ("N1", c_int32), ("N2", c_int32), ("nsplit", c_int32), ("pad_", c_int32)]


TN_GROUP_MAX = 8


class ReduceSeg(ctypes.Structure):
    """mirror of struct mphsir_reduce_seg"""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("n", c_int64), ("stride", c_int64), ("src_batch_stride", c_int64),
                ("dst_batch_stride", c_int64), ("nsplit", c_int32), ("nbatch", c_int32), ("rows", c_int32),
                ("dst_col_stride", c_int32), ("src_ld", c_int64), ("dst_ld", c_int64)]


REDUCE_MAX_SEGS = 32


_SYMBOLS = {
    # name: (restype, argtypes)
    "mphsir_version": (ctypes.c_char_p, []),
    "mphsir_last_error": (ctypes.c_char_p, []),
    "mphsir_device_arch": (c_int, [ctypes.c_char_p, c_int]),
    "mphsir_kernel_name": (ctypes.c_char_p, [c_int]),
    "mphsir_gemm_tn_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32]),
    "mphsir_dwconv_gram_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    "mphsir_pg_gate_bwd_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int, ctypes.POINTER(c_int32), ctypes.POINTER(c_int32)]),
    "mphsir_win_attn_bwd_workspace_bytes": (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int]),
    "mphsir_prof_enable": (c_int, [c_int]),
    "mphsir_prof_read": (c_int, [ctypes.POINTER(c_int), c_float_p]),
    "mphsir_gemm_tok": (c_int, [ctypes.POINTER(GemmArgs), c_int, c_void_p]),
    "mphsir_tvsp_text_map": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mphsir_tvsp_text_map_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mphsir_resize_bilinear": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "mphsir_layernorm_tok": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int32, c_void_p]),
    "mphsir_win_attn_fwd": (c_int, [ctypes.POINTER(WinAttnArgs), c_int, c_void_p]),
    "mphsir_win_attn_hdp": (c_int, [c_int, c_int]),
    "mphsir_pg_gate_fwd": (c_int, [ctypes.POINTER(PgFwdArgs), c_void_p]),
    "mphsir_dwconv_gram": (c_int, [ctypes.POINTER(GramArgs), c_int, c_void_p]),
    "mphsir_dwconv_gram_keeps_qk": (c_int, [c_int32, c_int32, c_int]),
    "mphsir_qkv_dwconv_gram": (c_int, [ctypes.POINTER(FusedGramArgs), c_int, c_void_p]),
    "mphsir_qkv_dwconv_gram_fits": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int]),
    "mphsir_qkv_dwconv_gram_rows_fits": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int, c_int32]),
    "mphsir_debug": (c_int, [c_int, c_void_p]),
    "mphsir_dwconv3x3_wgrad_tiled": (c_int, [c_int32, c_int32, c_int32, c_int]),
    "mphsir_spectral_fold": (c_int, [ctypes.POINTER(FoldArgs), c_int, c_void_p]),
    "mphsir_dwconv_gate": (c_int, [ctypes.POINTER(GateArgs), c_int, c_void_p]),
    "mphsir_dwconv_gate_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "mphsir_dwconv_gate_bwd_fits": (c_int, [c_int32, c_int32, c_int32, c_int]),
    "mphsir_gdfn_dw_bwd": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "mphsir_gdfn_dw_bwd_fits": (c_int, [c_int32, c_int32, c_int32, c_int]),
    "mphsir_gdfn_fused": (c_int, [ctypes.POINTER(GdfnArgs), c_int, c_void_p]),
    "mphsir_gdfn_fused_fits": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int]),
    "mphsir_gdfn_fused_tile_width": (c_int, [c_int32]),
    "mphsir_dwconv3x3": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32,
                                 c_int32, c_int, c_void_p]),
    "mphsir_dwconv3x3_bwd": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32,
                                     c_int32, c_int32, c_int, c_void_p]),
    "mphsir_dwconv3x3_wgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                       c_int32, c_int, c_void_p]),
    "mphsir_flat_adamw": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, ctypes.c_float, ctypes.c_float,
                                  ctypes.c_float, ctypes.c_float, ctypes.c_float, c_int32, ctypes.c_float, c_void_p, c_void_p]),
    "mphsir_grad_check": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "mphsir_flat_adamw_scaled": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                         ctypes.c_float, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p, c_void_p]),
    "mphsir_scaler_update": (c_int, [c_void_p, ctypes.c_float, ctypes.c_float, c_int32, c_void_p]),
    "mphsir_combine_bwd": (c_int, [c_void_p] * 7 + [c_int32] * 5 + [c_int, c_void_p]),
    "mphsir_win_attn_bwd": (c_int, [ctypes.POINTER(WinAttnBwdArgs), c_int, c_void_p]),
    "mphsir_win_attn_bwd_fits": (c_int, [c_int32, c_int32, c_int]),
    "mphsir_ln_bwd_win": (c_int, [c_void_p] * 6 + [c_int32] * 5 + [c_void_p, c_void_p, c_int32, c_int, c_void_p]),
    "mphsir_ln_bwd_win_dxn": (c_int, [c_void_p] * 8 + [c_int32] * 5 + [c_int, c_void_p]),
    "mphsir_ln_bwd_win_dxn_fits": (c_int, [c_int32, c_int]),
    "mphsir_ln_bwd_tok_dxn": (c_int, [c_void_p] * 9 + [c_int64, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "mphsir_gemm_tn": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                               c_int32, c_int32, c_int32, c_int, c_void_p]),
    "mphsir_spectral_fold_bwd": (c_int, [ctypes.POINTER(FoldBwdArgs), c_int, c_void_p]),
    "mphsir_pg_gate_bwd": (c_int, [ctypes.POINTER(PgBwdArgs), c_void_p]),
    "mphsir_spectral_dqkv_bwd": (c_int, [ctypes.POINTER(SpectralBwdArgs), c_int, c_void_p]),
    "mphsir_spectral_dqkv_bwd_fits": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int]),
    "mphsir_spectral_dqkv_bwd_slabs": (c_int, [c_int32, c_int32]),
    "mphsir_conv3x3_tok": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_int, c_void_p]),
    "mphsir_conv3x3_wgrad": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "mphsir_im2col3x3": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "mphsir_gdfn_gate_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p]),
    "mphsir_gated_mlp_bwd": (c_int, [ctypes.POINTER(MlpBwdArgs), c_int, c_void_p]),
    "mphsir_gated_mlp_wgrad": (c_int, [ctypes.POINTER(MlpWgradArgs), c_int, c_void_p]),
    "mphsir_gated_mlp_wgrad_fits": (c_int, [c_int32, c_int32, c_int]),
    "mphsir_gated_mlp_fwd": (c_int, [ctypes.POINTER(MlpArgs), c_int, c_void_p]),
    "mphsir_gated_mlp_fwd_fuses": (c_int, [c_int32, c_int64, c_int]),
    "mphsir_reduce_parts": (c_int, [ctypes.POINTER(ReduceSeg), c_int32, c_void_p]),
    "mphsir_gemm_tn_group": (c_int, [ctypes.POINTER(TnProblem), c_int32, c_int32, c_int, c_void_p]),
    "mphsir_pack_gather": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "mphsir_multi_copy": (c_int, [c_void_p, c_int32, c_int64, c_void_p]),
    "mphsir_nchw_to_cl": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int32, c_int, c_void_p]),
    "mphsir_cl_to_nchw_add": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int64, c_int, c_void_p]),
    "mphsir_task_weights": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "mphsir_mix_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, ctypes.c_float, c_int32, c_void_p]),
    "mphsir_l1_clamp_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p]),
}


def symbols():
    """Every entry point include/mphsir.h declares (checked by tests/test_cabi.py)."""
    return sorted(_SYMBOLS)


def _bind(lib):
    for name, (res, args) in _SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def load(path=None):
    """Load (once) and return the library.  Raises RuntimeError if it cannot be loaded."""
    global _lib, _is_emu
    if _lib is not None and path is None:
        return _lib
    p = path or _PATH
    if not os.path.exists(p):
        raise RuntimeError("mp-hsir_amd: HIP library %s not found -- run `python mp-hsir_amd/build.py` "
                           "(there is no CPU fallback)" % p)
    try:
        lib = _bind(ctypes.CDLL(p))
    except (OSError, AttributeError) as e:
        raise RuntimeError("mp-hsir_amd: cannot load %s: %s" % (p, e))
    _lib = lib
    _is_emu = path is not None and "hipemu" in os.path.abspath(path)
    return _lib


def use_library_for_tests(path):
    """TESTS ONLY: bind the ops to another build of the same sources (tests/hipemu)."""
    return load(path)


def is_emulated():
    return _is_emu


def check(rc, what):
    if rc != 0:
        raise RuntimeError("mp-hsir_amd: %s failed (%d): %s" % (what, rc, load().mphsir_last_error().decode()))
