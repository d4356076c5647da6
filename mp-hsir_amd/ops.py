"""Thin tensor-level wrappers over the C ABI (include/mphsir.h): shape checks, output allocation,
pointer/stream plumbing.  No arithmetic happens here."""
import ctypes

import torch

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1}


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _check(*tensors):
    emu = _lib.is_emulated()
    for t in tensors:
        if t is None:
            continue
        if emu and t.is_cuda:
            raise RuntimeError("mp-hsir_amd: emulated test library bound but got a GPU tensor")
        if not emu and not t.is_cuda:
            raise RuntimeError("mp-hsir_amd: ops run on the GPU only (got a CPU tensor; there is no CPU fallback)")


def _rows(t):
    """(rows, ld) of a 2-D view whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D view"
    return t.shape[0], t.stride(0)


def gemm_tok(x, w, bias=None, ln=None, epi=0, res=None, sa=None, gate=None, keep=None, geom=None, out=None):
    """Y = epi(pro(X) @ W^T).  x (M,K) row-major view; w (N,K) or per-sample (B,N,K) in x.dtype.
    ln = (weight, bias) fp32 -> LayerNorm prologue.  epi 0/1/2 as in include/mphsir.h.
    geom = (H, W, shift) for epi 2.  Returns (M,N)."""
    lib = _lib.load()
    _check(x, w, bias, res, sa, gate, keep)
    M, ldx = _rows(x)
    K = x.shape[1]
    per_sample = w.dim() == 3
    N = w.shape[-2]
    assert w.shape[-1] == K and w.is_contiguous() and w.dtype == x.dtype, (w.shape, K, w.dtype, x.dtype)
    y = out if out is not None else torch.empty((M, N), dtype=x.dtype, device=x.device)
    a = _lib.GemmArgs()
    a.X, a.ldx = _p(x), ldx
    a.W = _p(w)
    a.w_batch_stride = N * K if per_sample else 0
    a.rows_per_batch = M // w.shape[0] if per_sample else 0
    a.bias = _p(bias)
    if ln is not None:
        a.ln_w, a.ln_b = _p(ln[0]), _p(ln[1])
    a.Y, a.ldy = _p(y), _rows(y)[1]
    a.M, a.N, a.K, a.epi = M, N, K, epi
    if res is not None:
        a.R, a.ldr = _p(res), _rows(res)[1]
    if sa is not None:
        a.SA, a.ldsa = _p(sa), _rows(sa)[1]
    a.gate, a.keep = _p(gate), _p(keep)
    if geom is not None:
        a.H, a.Wimg, a.shift = geom
    _lib.check(lib.mphsir_gemm_tok(ctypes.byref(a), _DT[x.dtype], _stream(x)), "gemm_tok")
    return y
