"""Kernel-logic checks on CPU: the product's .hip sources compiled against tests/hipemu (a fiber-
based stand-in for the HIP runtime) and compared with the oracle.  These do not replace the `-m gpu`
parity tests -- they catch index/layout/fragment-map mistakes before a GPU run is spent on them."""
import pytest
import torch

from emu import bind_emulator
from oracle import mp_hsir_oracle as O
from util import rel_l2

DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-6, torch.bfloat16: 1.5e-2}


@pytest.fixture(scope="module", autouse=True)
def _emu():
    bind_emulator()


def rnd(shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,ln,epi", [(64, 64, 32, False, 0), (128, 96, 96, True, 0), (64, 48, 192, False, 1),
                                          (128, 32, 64, True, 1)])
def test_gemm_tok(dtype, M, N, K, ln, epi):
    from mp_hsir_amd import ops
    x, w = rnd((M, K), 1, dtype), rnd((N, K), 2, dtype, K ** -0.5)
    bias = rnd((N,), 3)
    lnw, lnb = 1 + 0.1 * rnd((K,), 4), 0.1 * rnd((K,), 5)
    res = rnd((M, N), 6, dtype)
    y = ops.gemm_tok(x, w, bias=bias, ln=(lnw, lnb) if ln else None, epi=epi, res=res if epi else None)
    xd = x.double()
    if ln:
        xd = O.layer_norm_c(xd, lnw.double(), lnb.double())
        if dtype == torch.bfloat16:
            xd = xd.to(dtype).double()
    ref = xd @ w.double().t() + bias.double() + (res.double() if epi else 0)
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_tok_per_sample_combine(dtype):
    """epi 2 with a per-sample weight: the folded channel attention + PGSSTB branch sum."""
    from mp_hsir_amd import ops
    B, H, W, C, shift = 2, 16, 16, 32, 4
    M = B * H * W
    v, Wb = rnd((M, C), 1, dtype), rnd((B, C, C), 2, dtype, C ** -0.5)
    res, sa = rnd((M, C), 3, dtype), rnd((M, C), 4, dtype)
    gate = rnd((B * (H // 8) * (W // 8), C), 5)
    keep = torch.tensor([1.25, 0.0])
    y = ops.gemm_tok(v, Wb, epi=2, res=res, sa=sa, gate=gate, keep=keep, geom=(H, W, shift))
    acc = torch.einsum("bnk,bck->bnc", v.double().reshape(B, H * W, C), Wb.double()).reshape(B, H, W, C)
    # gate lives in the shifted window frame: expand to windows, un-window, roll back
    gw = gate.double()[:, None, :].expand(-1, 64, -1)
    gimg = torch.roll(O.from_windows(gw, B, H, W), shifts=(shift, shift), dims=(1, 2))
    ref = res.double().reshape(B, H, W, C) + keep.double().reshape(B, 1, 1, 1) * (sa.double().reshape(B, H, W, C) * gimg + acc)
    assert rel_l2(y.reshape(B, H, W, C), ref) < TOL[dtype]
