"""Parity of the HIP kernels on a real MI355X, called through the C ABI (libmphsir.so), against the
fp64 oracle on the same seeded inputs.  Same checks as tests/test_emu_kernels.py, device = cuda."""
import os

import pytest
import torch

import kernel_checks as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _real_library():
    import mp_hsir_amd._lib as L
    L._lib = None
    L._is_emu = False
    lib = L.load()
    buf = (b" " * 64)
    import ctypes
    cbuf = ctypes.create_string_buffer(64)
    assert lib.mphsir_device_arch(cbuf, 64) == 0
    assert cbuf.value.decode().startswith("gfx950"), cbuf.value
    assert not L.is_emulated()


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("M,N,K_,ln,epi", K.GEMM_CASES + [(4096, 192, 64, True, 0), (2048, 128, 352, False, 1), (65536, 384, 128, True, 0), (32768, 128, 64, False, 1),
                                              (65536, 208, 96, True, 1)])
def test_gemm_tok(dtype, M, N, K_, ln, epi):
    K.check_gemm_tok("cuda", dtype, M, N, K_, ln, epi)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,hid,B,H,W,shift,keep,want_y,res", [(64, 170, 2, 8, 16, 0, True, True, False), (128, 340, 2, 8, 16, 4, False, False, True),
                                                               (32, 85, 2, 16, 16, 4, True, True, True), (96, 255, 2, 8, 16, 0, True, False, False),
                                                               (128, 340, 4, 64, 64, 4, True, True, False), (64, 170, 4, 128, 128, 0, True, True, True),
                                                               (128, 340, 32, 64, 64, 0, True, False, False)])
def test_gated_mlp_fused_branch_sum(dtype, C, hid, B, H, W, shift, keep, want_y, res):
    """the PGSSTB branch sum formed inside the gated-MLP launch == gemm_tok epi 2 followed by the plain launch, bit for bit -- also at
    the sizes of the training step and of the 512x512 forward, where the launch it replaces is the ring form of gemm_tok"""
    K.check_gated_mlp_branch_sum("cuda", dtype, C, hid, B=B, H=H, W=W, shift=shift, keep=keep, want_y=want_y, res=res)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K_,epi,ps,ldx,ldy", [(256, 64, 64, 0, 0, None, None), (512, 96, 96, 1, 0, 128, 104), (1024, 208, 160, 0, 4, None, 256),
                                                    (512, 128, 64, 2, 2, None, None), (256, 272, 32, 0, 0, None, None), (768, 48, 224, 1, 3, 256, None), (131072, 256, 256, 0, 32, 256, 384), (131072, 128, 384, 1, 0, 384, 128), (131072, 128, 128, 2, 32, 128, 128), (65536, 352, 128, 0, 0, None, None), (70016, 64, 192, 1, 0, None, None)])
def test_gemm_tok_ring(dtype, M, N, K_, epi, ps, ldx, ldy):
    K.check_gemm_tok_ring("cuda", dtype, M, N, K_, epi, ps, ldx, ldy)


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_gemm_tok_per_sample_combine(dtype):
    K.check_gemm_tok_per_sample_combine("cuda", dtype)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,hid", K.MLP_CASES + [(64, 170), (192, 510), (256, 680), (384, 1021)])
def test_gated_mlp(dtype, C, hid):
    K.check_gated_mlp("cuda", dtype, C, hid)
    K.check_gated_mlp("cuda", dtype, C, hid, tpw=2, M=256)
    if dtype != torch.float32:
        K.check_gated_mlp("cuda", dtype, C, hid, tpw=3, M=256)         # eight waves, one / two tiles per wave
        K.check_gated_mlp("cuda", dtype, C, hid, tpw=4, M=256)


def test_gated_mlp_large_auto_tiles():
    K.check_gated_mlp("cuda", torch.bfloat16, 128, 340, M=65536)


GPU_WIN_CASES = K.WIN_CASES + [
    ("natural_mode0", "encoder_level2.blocks.1.", 4, 4, (2, 32, 32, 128)),
    ("natural_mode0", "latent.blocks.1.", 8, 4, (2, 16, 16, 256)),
    ("natural_mode0", "refinement.blocks.1.", 2, 4, (1, 64, 64, 128)),
    ("remote_mode8", "encoder_level2.blocks.1.", 4, 4, (1, 32, 32, 192)),
    ("remote_mode8", "latent.blocks.0.", 8, 0, (1, 16, 16, 384)),
    ("remote_mode8", "refinement.blocks.1.", 2, 4, (1, 32, 32, 192)),
]


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("man,prefix,heads,shift,shape", GPU_WIN_CASES)
def test_win_attn(dtype, man, prefix, heads, shift, shape, manifest):
    K.check_win_attn("cuda", dtype, man, prefix, heads, shift, shape, manifest)


# (last case: more than 64 partials per sample -> the fold's inputs are pre-reduced, both in one reduce_parts launch)
GPU_SPEC_CASES = K.SPEC_CASES + [(32, 2, (1, 96, 96), 144), (64, 2, (2, 64, 64), 16), (128, 4, (1, 32, 32), 4), (256, 8, (2, 16, 16), 2),
                                 (128, 2, (1, 64, 64), 8), (192, 2, (1, 32, 32), 4), (384, 8, (1, 16, 16), 1),
                                 (64, 4, (1, 32, 32), 2), (128, 8, (1, 32, 32), 2)]


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,heads,shape,nsplit", GPU_SPEC_CASES)
def test_spectral_attention_chain(dtype, C, heads, shape, nsplit):
    K.check_spectral_attention_chain("cuda", dtype, C, heads, shape, nsplit)


GPU_FUSED_CASES = [(64, 2, (2, 64, 64), 8, False), (128, 4, (1, 32, 32), 2, False), (256, 8, (2, 16, 16), 1, False),
                   (128, 4, (1, 64, 64), 32, True), (256, 8, (1, 32, 32), 4, True), (96, 2, (1, 32, 32), 2, False),
                   (192, 4, (1, 16, 32), 1, True), (384, 8, (1, 16, 16), 2, False), (64, 2, (1, 128, 128), 16, False),
                   (128, 2, (1, 64, 64), 8, False), (128, 2, (2, 32, 32), 2, True), (192, 2, (1, 32, 32), 4, False)] + K.FUSED_SLAB_CASES


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,heads,shape,nsplit,ln", K.FUSED_CASES + GPU_FUSED_CASES)
def test_fused_pass_a(dtype, C, heads, shape, nsplit, ln):
    K.check_fused_pass_a("cuda", dtype, C, heads, shape, nsplit, ln)


@pytest.mark.parametrize("C,heads,shape,nsplit,ln,hg", K.FUSED_HG_CASES + [(256, 8, (1, 32, 32), 2, False, 8), (256, 8, (1, 32, 32), 2, True, 1),
                                                                   (384, 8, (1, 16, 16), 1, False, 4), (128, 4, (2, 64, 64), 4, False, 2)])
def test_fused_pass_a_head_groups(C, heads, shape, nsplit, ln, hg):
    K.check_fused_pass_a("cuda", torch.bfloat16, C, heads, shape, nsplit, ln, hgroups=hg)


@pytest.mark.parametrize("C,heads,shape,nsplit,ln", K.FUSED_CASES)
def test_fused_pass_a_fp32(C, heads, shape, nsplit, ln):
    K.check_fused_pass_a("cuda", torch.float32, C, heads, shape, nsplit, ln)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,heads,shape,rs,ln", K.ROWS_CASES + K.ROWS_CASES_GPU)
def test_fused_pass_a_rows(dtype, C, heads, shape, rs, ln):
    """the row-walking form of the fused pass A == two-kernel path and oracle, every shipped (C, head width)"""
    K.check_fused_pass_a("cuda", dtype, C, heads, shape, None, ln, row_segments=rs)


def test_fused_pass_a_is_deterministic_and_used_by_inference():
    """bitwise equal repeats (fixed-order partials, no atomics), and the no-grad forward of a block goes through it"""
    from mp_hsir_amd import ops
    x = torch.randn(2 * 64 * 64, 128, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(384, 128, device="cuda") * 128 ** -0.5).to(torch.bfloat16)
    w9 = torch.randn(9, 384, device="cuda") / 3
    a = ops.qkv_dwconv_gram(x, w, w9, 2, 64, 64, 128, 4, nsplit=8)
    b = ops.qkv_dwconv_gram(x, w, w9, 2, 64, 64, 128, 4, nsplit=8)
    assert all(torch.equal(p, q) for p, q in zip(a[:3], b[:3]))
    from mp_hsir_amd.net.MP_HSIR import PGSSTB
    from mp_hsir_amd import autograd_ops
    blk = PGSSTB(64, 2, [64, 64], 8, 0, 0.0, 2.66, 8, 128).cuda().eval()
    xin = torch.randn(1, 64, 64, 64, device="cuda", dtype=torch.bfloat16)
    ops.ACCOUNT = {}
    try:
        with torch.no_grad():
            autograd_ops.pgsstb(blk, xin, None, None)
        st = dict(ops.ACCOUNT)
    finally:
        ops.ACCOUNT = None
    assert "qkv_dwconv_gram" in st and "dwconv_gram" not in st, sorted(st)


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_gdfn_chain(dtype):
    K.check_gdfn_chain("cuda", dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D,hid,shape,nsplit", K.GDFN_FUSED_CASES + K.GDFN_FUSED_CASES_GPU)
def test_gdfn_fused(dtype, D, hid, shape, nsplit):
    K.check_gdfn_fused("cuda", dtype, D, hid, shape, nsplit)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (1, 5, 8, 96), (4, 64, 64, 384), (2, 32, 32, 704), (2, 16, 32, 128), (1, 8, 16, 32)])
def test_dwconv_plain(dtype, shape):
    K.check_dwconv_plain("cuda", dtype, shape)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", K.DW_BWD_CASES + K.DW_BWD_CASES_GPU)
def test_dwconv_bwd(dtype, shape):
    K.check_dwconv_bwd("cuda", dtype, shape)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,hid,hsplit", [(96, 255, 2), (96, 255, 4), (256, 680, 2), (192, 510, 4), (384, 1021, 4), (384, 1021, 8), (128, 340, 11)])
def test_gated_mlp_hidden_split(dtype, C, hid, hsplit):
    """the hidden dimension dealt to hsplit workgroups per token tile (small launches), partial fc2 products summed in order"""
    if dtype == torch.float32 and C >= 256:
        pytest.skip("fp32 at C >= 256 does not fit the LDS-staged form the split is built on")
    K.check_gated_mlp("cuda", dtype, C, hid, M=256, hsplit=hsplit)
    if dtype != torch.float32 or C < 192:
        K.check_gated_mlp_bwd("cuda", dtype, C, hid, hsplit=hsplit)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,hid,M,nch,ranges,keep", [(32, 85, 256, 1, 8, True), (64, 170, 640, 1, 8, False), (64, 170, 8192, 2, 16, True), (96, 255, 4160, 1, 8, True),
                                                      (96, 255, 2048, 2, 8, False), (128, 340, 16384, 1, 48, True), (128, 340, 8256, 2, 24, False),
                                                      (192, 510, 4096, 1, 16, True)])
def test_gated_mlp_wgrad(dtype, C, hid, M, nch, ranges, keep):
    """the parameter gradients of the gated MLP by recomputation (no h / [dval | dgate] in HBM): vs the oracle's autograd, vs the
    operand path it replaces, bitwise repeatable; ragged token ranges, empty ranges, one and two chunks per workgroup"""
    K.check_gated_mlp_wgrad("cuda", dtype, C, hid, M=M, nch=nch, ranges=ranges, keep=keep)


def test_gated_mlp_wgrad_training_shape():
    """the benchmark's level-1 shape (batch 32 x 64 x 64 tokens, C = 128) with the plan the library chooses"""
    K.check_gated_mlp_wgrad("cuda", torch.bfloat16, 128, 340, M=131072, nch=None, ranges=None, keep=True)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,hid", [(32, 85), (96, 255), (64, 170), (128, 340), (256, 680), (192, 510)])
def test_gated_mlp_bwd(dtype, C, hid):
    K.check_gated_mlp_bwd("cuda", dtype, C, hid)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,hid,variant", [(32, 85, 1), (32, 85, 3), (96, 255, 3), (128, 340, 1), (128, 340, 2), (128, 340, 3), (64, 170, 3),
                                           (256, 680, 1), (256, 680, 2), (32, 85, 4), (64, 170, 4), (96, 255, 4), (128, 340, 4)])
def test_gated_mlp_bwd_kernel_forms(dtype, C, hid, variant):
    K.check_gated_mlp_bwd("cuda", dtype, C, hid, variant=variant)


def test_gated_mlp_bwd_c384_bf16():
    K.check_gated_mlp_bwd("cuda", torch.bfloat16, 384, 1021)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("M,N1,N2,nsplit,batch", [(256, 64, 64, 2, 0), (200, 96, 32, 3, 0), (128, 32, 32, 1, 2),
                                                   (131072, 704, 128, None, 0), (4096, 128, 128, None, 32), (32768, 384, 128, None, 0)])
def test_gemm_tn(dtype, M, N1, N2, nsplit, batch):
    K.check_gemm_tn("cuda", dtype, M, N1, N2, nsplit, batch)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 8, 8, 31, 32), (2, 8, 4, 32, 31), (1, 8, 8, 64, 48), (4, 64, 64, 31, 64),
                                             (2, 16, 16, 256, 512), (2, 64, 64, 128, 31)])
@pytest.mark.parametrize("form", [1, 2])
def test_conv3x3(dtype, B, H, W, Cin, Cout, form):
    if form == 1 and dtype == torch.float32:
        pytest.skip("fp32 has one form")
    with K.tn_form(form):
        K.check_conv3x3("cuda", dtype, B, H, W, Cin, Cout)


def test_l1_clamp_loss():
    K.check_l1_clamp_loss("cuda")


def test_multi_copy():
    K.check_multi_copy("cuda")


def test_reduce_parts():
    K.check_reduce_parts("cuda")


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_pack_gather(dtype):
    K.check_pack_gather("cuda", dtype)


@pytest.mark.parametrize("M,N1,N2,nsplit,batch", [(200, 64, 128, 2, 0), (130, 136, 48, 1, 0), (96, 40, 56, 3, 2), (8192, 704, 128, 8, 0),
                                                  (4096, 128, 352, 5, 0), (4096, 64, 64, 4, 3), (4096, 384, 64, 7, 0)])
@pytest.mark.parametrize("form", [1, 2])
def test_gemm_tn_transposed_read_widths(M, N1, N2, nsplit, batch, form):
    with K.tn_form(form):
        K.check_gemm_tn("cuda", torch.bfloat16, M, N1, N2, nsplit, batch, tile128=True)


def test_reduce_block():
    K.check_reduce_block("cuda")


@pytest.mark.parametrize("form", [1, 2])
def test_gemm_tn_grouped(form):
    with K.tn_form(form):
        K.check_gemm_tn_grouped("cuda")


# ---- backward kernels at every benchmarked width: fp64 autograd of the oracle, full tensors --------------
from golden.cases import BLOCK_CASES

PGSSTB_CASES = [n for n, c in BLOCK_CASES.items() if c["kind"] == "pgsstb"]


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("name", PGSSTB_CASES)
def test_pgsstb_backward_vs_oracle_autograd(dtype, name):
    """C in {64,96,128,192,256,384} x head_dim in {32,48,64,96}, batch 2, DropPath factors on, 32x32 (shifted windows
    of every mask region): dX and all 29 parameter gradients <= 2e-5 (fp32) / 6e-2 (bf16) rel-L2."""
    errs = K.check_pgsstb_backward_oracle("cuda", dtype, name, B=2, hw=(32, 32))
    print(name, str(dtype), "worst", max(errs, key=errs.get), max(errs.values()))


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_pgsstb_backward_vs_oracle_autograd_bench_shape(dtype):
    """the benchmark's top level: C=128, 2 heads of 64, 64x64 patches, batch 4, no DropPath"""
    K.check_pgsstb_backward_oracle("cuda", dtype, "nat_refine", B=4, hw=(64, 64), drop_path=False)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,shift", [(64, 4), (128, 0), (256, 4), (96, 4), (192, 0), (384, 4)])
def test_combine_bwd(dtype, C, shift):
    K.check_combine_bwd("cuda", dtype, C, shift)


@pytest.mark.parametrize("C,cr,nW", [(64, 8, 320), (128, 8, 77), (256, 32, 64), (384, 32, 16), (512, 32, 40), (768, 64, 9), (192, 8, 33)])
def test_pg_gate_fwd(C, cr, nW):
    """the gate kernel alone vs the oracle, incl. C > 256 with r in {8, 12, 16} (C = 512 / r = 16, C = 768 / r = 12: a thread's
    channel changes from round to round there -- the register-resident Wup row path is for C | 256 only)"""
    K.check_pg_gate_fwd("cuda", C, cr, nW=nW)


@pytest.mark.parametrize("factor_dtype", K.DTYPES)
@pytest.mark.parametrize("C,cr", [(64, 8), (128, 16), (256, 32), (128, 8), (96, 8), (192, 16), (384, 32), (192, 8)])
def test_pg_gate_bwd(factor_dtype, C, cr):
    K.check_pg_gate_bwd("cuda", C, cr, nW=320, factor_dtype=factor_dtype)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,heads,nsp", [(64, 2, 5), (128, 4, 12), (96, 2, 3)])
def test_fold_bwd_takes_split_partials_of_dm(dtype, C, heads, nsp):
    K.check_fold_bwd_split_dm("cuda", dtype, C, heads, nsp=nsp)


@pytest.mark.parametrize("dtype", K.DTYPES)
@pytest.mark.parametrize("C,heads,shape,cross", [(64, 2, (2, 64, 64), True), (128, 2, (2, 32, 32), True), (128, 4, (2, 64, 64), False),
                                                 (256, 8, (2, 32, 32), False), (96, 2, (1, 64, 64), True), (192, 4, (1, 64, 64), False),
                                                 (384, 8, (1, 32, 32), False), (192, 2, (1, 32, 32), True)])
def test_channel_attention_bwd(dtype, C, heads, shape, cross):
    """the prompt modules' attention backward at both configurations' widths (TVSP: D, 2 heads; PromptFusion: 2D/4D, 4/8 heads)"""
    K.check_channel_attention_bwd("cuda", dtype, C, heads, shape, cross)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,heads,shape", [(64, 2, (2, 64, 64)), (128, 2, (2, 64, 64)), (128, 4, (3, 32, 32)), (256, 8, (4, 16, 16)), (96, 2, (1, 64, 64)),
                                           (192, 4, (2, 32, 32)), (192, 2, (1, 64, 64)), (384, 8, (2, 16, 16)), (32, 1, (1, 8, 16))])
def test_spectral_dqkv_bwd_against_the_three_launches(dtype, C, heads, shape):
    """every (C, head width) of both configurations: dt bitwise the three-launch path's with its rounding switched on (head widths 32 / 64),
    the product form within rounding of it and no further from fp64 than it"""
    print(K.check_spectral_dqkv_bwd("cuda", dtype, C, heads, shape))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,shape,shift", [(64, (4, 64, 64), 4), (128, (32, 64, 64), 0), (128, (32, 32, 32), 4), (256, (32, 16, 16), 4), (96, (2, 64, 64), 0),
                                           (192, (16, 32, 32), 4), (32, (1, 8, 8), 0)])
def test_ln_bwd_win_with_the_dxn_gemm_inside(dtype, C, shape, shift):
    print(K.check_ln_bwd_win_dxn("cuda", dtype, C, shape, shift))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape,hid", [((4, 64, 64), 340), ((2, 64, 64), 170), ((3, 32, 32), 680), ((2, 32, 32), 340), ((1, 64, 64), 255), ((2, 16, 32), 510), ((1, 8, 16), 85)])
def test_gdfn_gate_and_depthwise_backward_in_one_launch(dtype, shape, hid):
    print(K.check_gdfn_dw_bwd("cuda", dtype, shape, hid))


def test_gdfn_dw_bwd_at_the_training_shape():
    print(K.check_gdfn_dw_bwd("cuda", torch.bfloat16, (32, 64, 64), 340, nblk=16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,Kd,M", [(128, 704, 131072), (64, 384, 131072), (256, 1408, 32768), (128, 384, 131072), (256, 768, 32768), (192, 1024, 4096), (96, 512, 65536)])
def test_ln_bwd_tok_with_the_conv_gradient_inside(dtype, C, Kd, M):
    print(K.check_ln_bwd_tok_dxn("cuda", dtype, C, Kd, M))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,heads,N", [(128, 4, 1024), (256, 8, 256), (64, 2, 1024), (192, 4, 1024), (384, 8, 256), (128, 2, 1024)])
def test_fold_bwd_forms_dm_itself(dtype, C, heads, N):
    """the fold backward with dM = d_out^T v formed inside (the lower pyramid levels of both configurations, batch 32)"""
    print(K.check_fold_bwd_forms_dm("cuda", dtype, C, heads, B=32, N=N))


def test_spectral_dqkv_bwd_at_the_training_shape():
    print(K.check_spectral_dqkv_bwd("cuda", torch.bfloat16, 128, 2, (32, 64, 64), nblk=80))


# ---- fp16 storage (dtype code 2; BASELINE configs[4]: remote-sensing training in fp16 + loss scaling) ------------------------
F16 = torch.float16


def test_fp16_forward_kernels(manifest):
    K.check_gemm_tok("cuda", F16, 4096, 192, 64, True, 0)
    K.check_gemm_tok("cuda", F16, 65536, 208, 96, True, 1)
    K.check_gemm_tok_per_sample_combine("cuda", F16)
    for C, hid in [(96, 255), (192, 510), (384, 1021), (128, 340)]:
        K.check_gated_mlp("cuda", F16, C, hid)
    for man, prefix, heads, shift, shape in GPU_WIN_CASES:
        K.check_win_attn("cuda", F16, man, prefix, heads, shift, shape, manifest)
    for C, heads, shape, nsplit in GPU_SPEC_CASES:
        K.check_spectral_attention_chain("cuda", F16, C, heads, shape, nsplit)
    K.check_gdfn_chain("cuda", F16)
    K.check_conv3x3("cuda", F16, 4, 64, 64, 31, 64)
    K.check_pack_gather("cuda", F16)


@pytest.mark.parametrize("name", PGSSTB_CASES)
def test_pgsstb_backward_vs_oracle_autograd_fp16(name):
    """every shape class in fp16: dX and all parameter gradients <= 2e-2 rel-L2 against fp64 autograd of the oracle"""
    errs = K.check_pgsstb_backward_oracle("cuda", F16, name, B=2, hw=(32, 32))
    print(name, "fp16 worst", max(errs, key=errs.get), max(errs.values()))


def test_fp16_backward_kernels():
    for C, hid in [(96, 255), (192, 510), (384, 1021), (128, 340)]:
        K.check_gated_mlp_bwd("cuda", F16, C, hid)
    K.check_gemm_tn("cuda", F16, 131072, 704, 128, None, 0)
    K.check_gemm_tn("cuda", F16, 4096, 128, 352, 5, 0, tile128=True)
    K.check_gemm_tn_grouped("cuda", F16)
    K.check_dwconv_plain("cuda", F16, (4, 64, 64, 384))
    K.check_channel_attention_bwd("cuda", F16, 96, 2, (1, 64, 64), True)
    K.check_channel_attention_bwd("cuda", F16, 384, 8, (1, 32, 32), False)


def test_loss_scaler_kernels():
    K.check_loss_scaler("cuda")


@pytest.mark.parametrize("dtype", K.DTYPES + [F16])
def test_resamplers(dtype):
    K.check_resamplers("cuda", dtype)
    K.check_resamplers("cuda", dtype, B=2, ps=64, D=64, H=512, W=512)
    K.check_resamplers("cuda", dtype, B=4, ps=32, D=128, H=32, W=48)
    K.check_resamplers("cuda", dtype, B=1, ps=4, D=32, H=48, W=64)          # ratios 12 and 16: the backward's gather bounds


@pytest.mark.parametrize("dtype", K.DTYPES + [F16])
def test_layernorm_tok(dtype):
    K.check_layernorm_tok("cuda", dtype)
    K.check_layernorm_tok("cuda", dtype, M=131072, C=64)
    K.check_layernorm_tok("cuda", dtype, M=4099, C=192)
    K.check_layernorm_tok("cuda", dtype, M=70, C=20)


@pytest.mark.parametrize("dtype", K.DTYPES + [F16])
def test_heads(dtype):
    K.check_heads("cuda", dtype)
    K.check_heads("cuda", dtype, B=32, C=31, H=64, W=64)
    K.check_heads("cuda", dtype, B=1, C=172, H=40, W=36, T=7, n=1)       # the remote-sensing cube's channel count; ragged pixel tiles


def test_win_attn_bwd_head_split():
    K.check_win_attn_bwd_head_split("cuda", torch.bfloat16)
    K.check_win_attn_bwd_head_split("cuda", torch.float16, C=256, heads=8, shape=(4, 16, 16))


def test_dwconv_gate_bwd():
    K.check_dwconv_gate_bwd("cuda", torch.bfloat16)
    K.check_dwconv_gate_bwd("cuda", torch.float16, shape=(1, 8, 16), hid=170)
    K.check_dwconv_gate_bwd("cuda", torch.bfloat16, shape=(4, 64, 64), hid=340)


# ---- fp32 twins of the 16-bit-only forms: the fp32 kernels on the rounded inputs, ~3x tighter than the oracle tolerance ----------
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_fp32_twins_of_16bit_only_forms(dt):
    K.check_gated_mlp_fp32_twin("cuda", dt, 128, 340, 4096, 3)                 # eight waves
    K.check_gated_mlp_fp32_twin("cuda", dt, 256, 680, 1024, 0, hsplit=2)        # hidden split + ordered combine
    K.check_pass_a_rows_fp32_twin("cuda", dt, 64, 2, (1, 64, 128), 2)
    K.check_pass_a_rows_fp32_twin("cuda", dt, 128, 4, (1, 64, 128), 2)
    K.check_gdfn_fused_fp32_twin("cuda", dt, 64, 170, (1, 64, 64))


@pytest.mark.parametrize("dtype", K.DTYPES)
def test_gated_mlp_second_residual(dtype):
    """mphsir_gated_mlp_fwd with R: every kernel form adds the BaseBlock skip in its epilogue (direct form, LDS forms, hidden-split combine)"""
    K.check_gated_mlp("cuda", dtype, 32, 85, res=True)
    K.check_gated_mlp("cuda", dtype, 128, 340, M=1024, res=True)
    if dtype != torch.float32:
        K.check_gated_mlp("cuda", dtype, 128, 340, tpw=3, M=1024, res=True)
        K.check_gated_mlp("cuda", dtype, 256, 680, M=256, hsplit=2, res=True)
