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


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,hid", [(32, 85), (96, 255), (128, 340)])
def test_gated_mlp(dtype, C, hid):
    from mp_hsir_amd import ops
    M = 128
    x = rnd((M, C), 1, dtype)
    P = {"fc1.weight": rnd((2 * hid, C), 2, scale=C ** -0.5), "fc1.bias": 0.1 * rnd((2 * hid,), 3),
         "fc2.weight": rnd((C, hid), 4, scale=hid ** -0.5), "fc2.bias": 0.1 * rnd((C,), 5)}
    lnw, lnb = 1 + 0.1 * rnd((C,), 6), 0.1 * rnd((C,), 7)
    keep = torch.tensor([1.0, 1.5])
    W1, b1, W2 = ops.pack_gated_mlp(P["fc1.weight"], P["fc1.bias"], P["fc2.weight"], dtype)
    y = ops.gated_mlp_fwd(x, lnw, lnb, W1, b1, W2, P["fc2.bias"], keep=keep, rows_per_batch=64)
    Pd = {k: (v.to(dtype) if k.endswith("weight") else v).double() for k, v in P.items()}
    xn = O.layer_norm_c(x.double(), lnw.double(), lnb.double())
    ref = x.double() + keep.double().repeat_interleave(64)[:, None] * O.gated_mlp(Pd, "", xn)
    assert rel_l2(y, ref) < TOL[dtype]


def _block_params(manifest_entry, prefix, seed_shift=0):
    from util import params_from_manifest
    return params_from_manifest(manifest_entry, prefix, dtype=torch.float32)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("man,prefix,heads,shift,shape", [
    ("tiny", "encoder_level1.blocks.1.", 1, 4, (1, 16, 24, 32)),
    ("natural_mode0", "encoder_level1.blocks.1.", 2, 4, (2, 16, 16, 64)),
    ("natural_mode0", "refinement.blocks.0.", 2, 0, (1, 8, 16, 128)),
    ("remote_mode8", "encoder_level1.blocks.1.", 2, 4, (1, 16, 8, 96)),
])
def test_win_attn(dtype, man, prefix, heads, shift, shape, manifest):
    from mp_hsir_amd import ops
    P = _block_params(manifest[man], prefix)
    B, H, W, C = shape
    x = rnd(shape, 11, dtype)
    wq = P["attn.qkv.weight"].to(dtype)
    wp = P["attn.proj.weight"].to(dtype)
    pg = {k[len("local_spectral_attn."):]: v.contiguous() for k, v in P.items() if k.startswith("local_spectral_attn.")}
    pg["prompt_param"] = pg["prompt_param"].reshape(128, -1).contiguous()
    sa, gate = ops.win_attn_fwd(x, P["norm1.weight"], P["norm1.bias"], wq, P["attn.qkv.bias"],
                                P["attn.relative_position_bias_table"], ops.pack_win_proj(P["attn.proj.weight"], heads, dtype),
                                P["attn.proj.bias"], pg, heads, shift)
    # oracle on the same (dtype-rounded) weights, fp64 arithmetic
    Pd = {k: v.double() for k, v in P.items()}
    Pd["attn.qkv.weight"], Pd["attn.proj.weight"] = wq.double(), wp.double()
    xn = O.layer_norm_c(x.double(), Pd["norm1.weight"], Pd["norm1.bias"])
    if shift:
        xn = torch.roll(xn, (-4, -4), (1, 2))
    mask = O.shift_mask(H, W, torch.float64) if shift else None
    saw = O.spatial_attention(Pd, "attn.", O.to_windows(xn), heads, mask)
    g_ref = O.pg_spectral_gate(Pd, "local_spectral_attn.", saw)
    sa_ref = O.from_windows(saw, B, H, W)
    if shift:
        sa_ref = torch.roll(sa_ref, (4, 4), (1, 2))
    assert rel_l2(sa, sa_ref) < TOL[dtype] * (2 if dtype == torch.bfloat16 else 1)
    assert rel_l2(gate, g_ref) < TOL[dtype] * (4 if dtype == torch.bfloat16 else 1)
