"""Kernel-logic checks on CPU: the product's .hip sources compiled against tests/hipemu (a fiber-
based stand-in for the HIP runtime) and compared with the oracle.  These do not replace the `-m gpu`
parity tests -- they catch index/layout/fragment-map mistakes before a GPU run is spent on them."""
import pytest
import torch

import kernel_checks as K
from emu import bind_emulator


@pytest.fixture(scope="module", autouse=True)
def _emu():
    bind_emulator()


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("M,N,K_,ln,epi", K.GEMM_CASES)
def test_gemm_tok(dtype, M, N, K_, ln, epi):
    K.check_gemm_tok("cpu", dtype, M, N, K_, ln, epi)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K_,epi,ps,ldx,ldy", [(256, 64, 64, 0, 0, None, None), (512, 96, 96, 1, 0, 128, 104), (1024, 208, 160, 0, 4, None, 256),
                                                    (512, 128, 64, 2, 2, None, None), (256, 272, 32, 0, 0, None, None), (768, 48, 224, 1, 3, 256, None)])
def test_gemm_tok_ring(dtype, M, N, K_, epi, ps, ldx, ldy):
    K.check_gemm_tok_ring("cpu", dtype, M, N, K_, epi, ps, ldx, ldy)


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_gemm_tok_per_sample_combine(dtype):
    K.check_gemm_tok_per_sample_combine("cpu", dtype)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,hid", K.MLP_CASES)
def test_gated_mlp(dtype, C, hid):
    K.check_gated_mlp("cpu", dtype, C, hid)
    K.check_gated_mlp("cpu", dtype, C, hid, tpw=2, M=256)
    if dtype != torch.float32:
        K.check_gated_mlp("cpu", dtype, C, hid, tpw=3, M=256)          # eight waves, one / two tiles per wave
        K.check_gated_mlp("cpu", dtype, C, hid, tpw=4, M=256)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("man,prefix,heads,shift,shape", K.WIN_CASES)
def test_win_attn(dtype, man, prefix, heads, shift, shape, manifest):
    K.check_win_attn("cpu", dtype, man, prefix, heads, shift, shape, manifest)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,heads,shape,nsplit", K.SPEC_CASES)
def test_spectral_attention_chain(dtype, C, heads, shape, nsplit):
    K.check_spectral_attention_chain("cpu", dtype, C, heads, shape, nsplit)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,heads,shape,nsplit,ln", K.FUSED_CASES)
def test_fused_pass_a(dtype, C, heads, shape, nsplit, ln):
    K.check_fused_pass_a("cpu", dtype, C, heads, shape, nsplit, ln)


@pytest.mark.parametrize("C,heads,shape,nsplit,ln", K.FUSED_SLAB_CASES)
def test_fused_pass_a_slabs(C, heads, shape, nsplit, ln):
    K.check_fused_pass_a("cpu", torch.bfloat16, C, heads, shape, nsplit, ln)


@pytest.mark.parametrize("C,heads,shape,nsplit,ln,hg", K.FUSED_HG_CASES)
def test_fused_pass_a_head_groups(C, heads, shape, nsplit, ln, hg):
    K.check_fused_pass_a("cpu", torch.bfloat16, C, heads, shape, nsplit, ln, hgroups=hg)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,heads,shape,rs,ln", K.ROWS_CASES)
def test_fused_pass_a_rows(dtype, C, heads, shape, rs, ln):
    K.check_fused_pass_a("cpu", dtype, C, heads, shape, None, ln, row_segments=rs)


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_gdfn_chain(dtype):
    K.check_gdfn_chain("cpu", dtype)


@pytest.mark.parametrize("D,hid,shape,nsplit", K.GDFN_FUSED_CASES)
def test_gdfn_fused(D, hid, shape, nsplit):
    K.check_gdfn_fused("cpu", torch.bfloat16, D, hid, shape, nsplit)


def test_gdfn_fused_f16():
    K.check_gdfn_fused("cpu", F16, 128, 340, (1, 8, 32), 2)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (1, 5, 8, 96), (2, 16, 32, 128), (1, 8, 16, 32)])      # the last two: the LDS-tile form (16-bit)
def test_dwconv_plain(dtype, shape):
    K.check_dwconv_plain("cpu", dtype, shape)


@pytest.mark.parametrize("shape", K.DW_BWD_CASES)
def test_dwconv_bwd(shape):
    K.check_dwconv_bwd("cpu", torch.bfloat16, shape)


def test_dwconv_bwd_f16():
    K.check_dwconv_bwd("cpu", torch.float16, (1, 8, 32, 64))


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,hid,hsplit", [(96, 255, 2), (96, 255, 4)])
def test_gated_mlp_hidden_split(dtype, C, hid, hsplit):
    """the hidden dimension dealt to hsplit workgroups per token tile (small launches), partial fc2 products summed in order"""
    if dtype == torch.float32 and C >= 256:
        pytest.skip("fp32 at C >= 256 does not fit the LDS-staged form the split is built on")
    K.check_gated_mlp("cpu", dtype, C, hid, M=256, hsplit=hsplit)
    if dtype != torch.float32 or C < 192:
        K.check_gated_mlp_bwd("cpu", dtype, C, hid, hsplit=hsplit)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,hid,M,nch,ranges,keep", [(32, 85, 256, 1, 8, True), (64, 170, 640, 2, 8, False), (96, 255, 576, 1, 8, True),
                                                      (128, 340, 1088, 1, 16, True), (128, 340, 320, 2, 8, False)])
def test_gated_mlp_wgrad(dtype, C, hid, M, nch, ranges, keep):
    K.check_gated_mlp_wgrad("cpu", dtype, C, hid, M=M, nch=nch, ranges=ranges, keep=keep)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,hid", [(32, 85), (96, 255)])
def test_gated_mlp_bwd(dtype, C, hid):
    K.check_gated_mlp_bwd("cpu", dtype, C, hid)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,hid,variant", [(32, 85, 1), (32, 85, 3), (96, 255, 3), (32, 85, 4), (64, 170, 4)])
def test_gated_mlp_bwd_kernel_forms(dtype, C, hid, variant):
    K.check_gated_mlp_bwd("cpu", dtype, C, hid, variant=variant)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("M,N1,N2,nsplit,batch", [(256, 64, 64, 2, 0), (200, 96, 32, 3, 0), (128, 32, 32, 1, 2)])
def test_gemm_tn(dtype, M, N1, N2, nsplit, batch):
    K.check_gemm_tn("cpu", dtype, M, N1, N2, nsplit, batch)


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_gemm_tn_tile128(dtype):
    K.check_gemm_tn("cpu", dtype, 192, 160, 136, 2, 0, tile128=True)


@pytest.mark.parametrize("M,N1,N2,nsplit,batch", [(200, 64, 128, 2, 0), (130, 136, 48, 1, 0), (96, 40, 56, 3, 2)])
@pytest.mark.parametrize("form", [1, 2])
def test_gemm_tn_transposed_read_widths(M, N1, N2, nsplit, batch, form):
    import torch
    with K.tn_form(form):
        K.check_gemm_tn("cpu", torch.bfloat16, M, N1, N2, nsplit, batch, tile128=True)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 8, 8, 31, 32), (2, 8, 4, 32, 31), (1, 8, 8, 64, 48)])
@pytest.mark.parametrize("form", [1, 2])
def test_conv3x3(dtype, B, H, W, Cin, Cout, form):
    if form == 1 and dtype == torch.float32:
        pytest.skip("fp32 has one form")
    with K.tn_form(form):
        K.check_conv3x3("cpu", dtype, B, H, W, Cin, Cout)


def test_l1_clamp_loss():
    K.check_l1_clamp_loss("cpu")


def test_multi_copy():
    K.check_multi_copy("cpu")


def test_reduce_parts():
    K.check_reduce_parts("cpu")


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_pack_gather(dtype):
    K.check_pack_gather("cpu", dtype)


def test_reduce_block():
    K.check_reduce_block("cpu")


@pytest.mark.parametrize("form", [1, 2])
def test_gemm_tn_grouped(form):
    with K.tn_form(form):
        K.check_gemm_tn_grouped("cpu")


@pytest.mark.parametrize("dtype,name", [(torch.float32, "nat_refine"), (torch.bfloat16, "rs_refine"), (torch.float32, "nat_latent")])
def test_pgsstb_backward_vs_oracle_autograd(dtype, name):
    K.check_pgsstb_backward_oracle("cpu", dtype, name, B=2, hw=(8, 16))


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_combine_bwd(dtype):
    K.check_combine_bwd("cpu", dtype)


def test_pg_gate_fwd():
    K.check_pg_gate_fwd("cpu", 128, 8)
    K.check_pg_gate_fwd("cpu", 512, 32, nW=5)      # r = 16 at C > 256: not the fixed-channel path
    K.check_pg_gate_fwd("cpu", 192, 8, nW=7)       # r = 24: the remote-sensing dec1 / refinement width


def test_pg_gate_bwd():
    K.check_pg_gate_bwd("cpu", 128, 8)
    K.check_pg_gate_bwd("cpu", 384, 32, nW=5, factor_dtype=torch.bfloat16)
    K.check_pg_gate_bwd("cpu", 192, 8, nW=6)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,heads,nsp", [(64, 2, 5), (128, 4, 12), (96, 2, 3)])
def test_fold_bwd_takes_split_partials_of_dm(dtype, C, heads, nsp):
    K.check_fold_bwd_split_dm("cpu", dtype, C, heads, nsp=nsp)


@pytest.mark.parametrize("dtype,C,heads,shape,cross", [(torch.float32, 64, 2, (1, 16, 16), False), (torch.bfloat16, 128, 2, (2, 8, 16), True)])
def test_channel_attention_bwd(dtype, C, heads, shape, cross):
    K.check_channel_attention_bwd("cpu", dtype, C, heads, shape, cross)


@pytest.mark.parametrize("dtype,C,shape,shift", [(torch.bfloat16, 64, (2, 16, 16), 4), (torch.float16, 128, (1, 8, 16), 0), (torch.bfloat16, 96, (1, 16, 8), 4),
                                                 (torch.bfloat16, 256, (1, 8, 8), 4)])
def test_ln_bwd_win_with_the_dxn_gemm_inside(dtype, C, shape, shift):
    print(K.check_ln_bwd_win_dxn("cpu", dtype, C, shape, shift))


@pytest.mark.parametrize("dtype,shape,hid", [(torch.bfloat16, (2, 8, 16), 85), (torch.float16, (1, 16, 32), 40), (torch.bfloat16, (1, 8, 32), 170)])
def test_gdfn_gate_and_depthwise_backward_in_one_launch(dtype, shape, hid):
    print(K.check_gdfn_dw_bwd("cpu", dtype, shape, hid))


@pytest.mark.parametrize("dtype,C,Kd,M", [(torch.bfloat16, 64, 384, 128), (torch.float16, 128, 704, 64), (torch.bfloat16, 256, 160, 64), (torch.bfloat16, 96, 96, 64)])
def test_ln_bwd_tok_with_the_conv_gradient_inside(dtype, C, Kd, M):
    print(K.check_ln_bwd_tok_dxn("cpu", dtype, C, Kd, M))


@pytest.mark.parametrize("dtype,C,heads,N", [(torch.bfloat16, 64, 2, 256), (torch.float16, 128, 2, 128), (torch.bfloat16, 96, 2, 64)])
def test_fold_bwd_forms_dm_itself(dtype, C, heads, N):
    print(K.check_fold_bwd_forms_dm("cpu", dtype, C, heads, B=2, N=N))


@pytest.mark.parametrize("dtype,C,heads,shape", [(torch.bfloat16, 32, 1, (2, 8, 16)), (torch.bfloat16, 64, 2, (1, 16, 32)), (torch.float16, 128, 2, (1, 8, 16)),
                                                 (torch.bfloat16, 96, 2, (1, 8, 16))])
def test_spectral_dqkv_bwd_against_the_three_launches(dtype, C, heads, shape):
    print(K.check_spectral_dqkv_bwd("cpu", dtype, C, heads, shape))


def test_channel_attention_bwd_self_16bit_takes_the_fused_launch():
    from mp_hsir_amd import ops
    ops.ACCOUNT = {}
    try:
        K.check_channel_attention_bwd("cpu", torch.bfloat16, 64, 2, (1, 16, 16), False)
        assert "spectral_dqkv_bwd" in ops.ACCOUNT and "dwconv3x3_bwd" not in ops.ACCOUNT, sorted(ops.ACCOUNT)
    finally:
        ops.ACCOUNT = None


# ---- fp16 storage (dtype code 2, the reference's 16-mixed precision): same kernels, v_mfma_f32_16x16x32_f16 ------------------
F16 = torch.float16


def test_fp16_forward_kernels(manifest):
    K.check_gemm_tok("cpu", F16, 128, 96, 96, True, 0)
    K.check_gemm_tok("cpu", F16, 64, 48, 192, False, 1)
    K.check_gemm_tok_per_sample_combine("cpu", F16)
    K.check_gated_mlp("cpu", F16, 96, 255)
    K.check_win_attn("cpu", F16, "remote_mode8", "encoder_level1.blocks.1.", 2, 4, (1, 16, 8, 96), manifest)
    K.check_spectral_attention_chain("cpu", F16, 64, 2, (1, 16, 16), 2)
    K.check_gdfn_chain("cpu", F16)
    K.check_conv3x3("cpu", F16, 1, 8, 8, 31, 32)


def test_fp16_backward_kernels():
    K.check_gated_mlp_bwd("cpu", F16, 96, 255)
    K.check_gemm_tn("cpu", F16, 200, 96, 32, 3, 0)
    K.check_gemm_tn("cpu", F16, 256, 136, 72, 2, 0, tile128=True)
    K.check_dwconv_plain("cpu", F16, (2, 8, 8, 32))
    K.check_combine_bwd("cpu", F16, 96, 4)
    K.check_channel_attention_bwd("cpu", F16, 64, 2, (1, 16, 16), True)
    K.check_pgsstb_backward_oracle("cpu", F16, "rs_enc1", B=2, hw=(8, 16))


def test_loss_scaler_kernels():
    K.check_loss_scaler("cpu")


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_resamplers(dtype):
    K.check_resamplers("cpu", dtype)
    K.check_resamplers("cpu", dtype, B=2, ps=8, D=16, H=64, W=64)       # the 8x upsample of the 512x512 path, scaled down
    K.check_resamplers("cpu", dtype, B=1, ps=4, D=16, H=48, W=64)       # ratios 12 and 16: the backward's gather bounds


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_layernorm_tok(dtype):
    K.check_layernorm_tok("cpu", dtype)
    K.check_layernorm_tok("cpu", dtype, M=70, C=20)          # element-wise form, a ragged last workgroup


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_heads(dtype):
    K.check_heads("cpu", dtype)
    K.check_heads("cpu", dtype, B=1, C=40, H=9, W=7, T=7, n=1)          # ragged pixel tiles, a second channel pad


def test_win_attn_bwd_head_split():
    K.check_win_attn_bwd_head_split("cpu", torch.bfloat16)


def test_dwconv_gate_bwd():
    K.check_dwconv_gate_bwd("cpu", torch.bfloat16)
    K.check_dwconv_gate_bwd("cpu", torch.float16, shape=(1, 8, 16), hid=170)


def test_entry_points_reject_bad_arguments():
    """The C-ABI entry points validate their arguments and report through mphsir_last_error (no launch, no crash): shapes a kernel
    does not cover, splits that do not divide, aliasing outputs, missing alignment."""
    from mp_hsir_amd import ops
    K._use("cpu")
    bf = torch.bfloat16
    x = K.rnd((1, 8, 16, 64), 1, bf)
    w9 = torch.zeros((9, 64))
    # depthwise backward: odd sizes are not covered by the one-launch form (the wrapper falls back; the entry point refuses)
    assert not ops.dwconv3x3_bwd_fits(8, 24, 64, bf) and not ops.dwconv3x3_bwd_fits(8, 16, 64, torch.float32)
    # fused GDFN: widths / dtypes outside the table, a tile split that does not divide
    assert not ops.gdfn_fused_fits(96, 256, 8, 16, bf) and not ops.gdfn_fused_fits(64, 192, 8, 16, torch.float32)
    D, HP = 64, 192
    x2 = x.reshape(-1, D)
    ln = (torch.ones(D), torch.zeros(D))
    w_in, w_out, w9g = torch.zeros((2 * HP, D), dtype=bf), torch.zeros((D, HP), dtype=bf), torch.zeros((9, 2 * HP))
    with pytest.raises(RuntimeError, match="nsplit"):
        ops.gdfn_fused(x2, ln, w_in, w9g, w_out, 1, 8, 16, nsplit=3)
    # window-attention backward: head_split must divide the heads
    C, heads = 128, 4
    xw, dsa = K.rnd((1, 8, 8, C), 2, bf), K.rnd((1, 8, 8, C), 3, bf)
    args = (xw, dsa, torch.zeros((1, C)), torch.ones(C), torch.zeros(C), torch.zeros((3 * C, C), dtype=bf), torch.zeros(3 * C),
            torch.zeros((225, heads)), torch.zeros((C, C), dtype=bf), heads, 0)
    with pytest.raises(RuntimeError, match="head_split"):
        ops.win_attn_bwd(*args, head_split=3)
    # the row-walking pass A refuses a segment count that leaves fewer than four rows
    xa = K.rnd((1 * 8 * 32, 64), 4, bf)
    with pytest.raises(RuntimeError, match="row_segments"):
        ops.qkv_dwconv_gram(xa, torch.zeros((192, 64), dtype=bf), torch.zeros((9, 192)), 1, 8, 32, 64, 2, row_segments=4)
    # conv weight gradient without im2col: 16-bit types only
    with pytest.raises((RuntimeError, AssertionError)):
        ops.conv3x3_wgrad(torch.zeros((128, 32)), torch.zeros((1, 8, 16, 32)))


# ---- fp32 twins of the 16-bit-only forms: the fp32 kernels on the rounded inputs, ~3x tighter than the oracle tolerance ----------
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_fp32_twins_of_16bit_only_forms(dt):
    K.check_gated_mlp_fp32_twin("cpu", dt, 128, 340, 256, 3)                 # eight waves
    K.check_gated_mlp_fp32_twin("cpu", dt, 256, 680, 256, 0, hsplit=2)        # hidden split + ordered combine
    K.check_pass_a_rows_fp32_twin("cpu", dt, 64, 2, (1, 32, 64), 2)
    K.check_pass_a_rows_fp32_twin("cpu", dt, 128, 4, (1, 32, 64), 2)
    K.check_gdfn_fused_fp32_twin("cpu", dt, 64, 170, (1, 16, 32))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,hid,shift,keep,want_y,res", [(64, 170, 0, True, True, False), (128, 340, 4, False, False, True),
                                                          (32, 85, 4, True, True, True), (96, 255, 0, True, False, False)])
def test_gated_mlp_fused_branch_sum(dtype, C, hid, shift, keep, want_y, res):
    """the PGSSTB branch sum formed inside the gated-MLP launch == gemm_tok epi 2 followed by the plain launch, bit for bit"""
    K.check_gated_mlp_branch_sum("cpu", dtype, C, hid, shift=shift, keep=keep, want_y=want_y, res=res)


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_gated_mlp_second_residual(dtype):
    """mphsir_gated_mlp_fwd with R: every kernel form adds the BaseBlock skip in its epilogue (direct form, LDS forms, hidden-split combine)"""
    K.check_gated_mlp("cpu", dtype, 32, 85, res=True)
    K.check_gated_mlp("cpu", dtype, 128, 340, M=256, res=True)
    if dtype != torch.float32:
        K.check_gated_mlp("cpu", dtype, 128, 340, tpw=3, M=256, res=True)
        K.check_gated_mlp("cpu", dtype, 256, 680, M=256, hsplit=2, res=True)
