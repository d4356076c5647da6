"""Register spills of the built kernels, read from the code-object metadata hipcc writes (CPU test: no GPU needed).

Every bf16 / fp16 instantiation the natural-scene training step and the 512x512 forward launch (profiles/*_replay_only.csv,
*_kernel_stats.csv) must be spill-free; the few that are not -- cold launches, 1-6 per step -- are listed with their present counts so that
a regression (or a new spilling instantiation) fails here instead of showing up as scratch traffic in a profile."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# the kernels that carry the step (serial trace): zero spills, no scratch
HOT = [
    r"gated_mlp_lds_kernelIDF16[b_]Li(64|128)ELi1ELi8ELb[01]E",          # forward, with and without the fused branch sum
    r"gated_mlp_lds_kernelIDF16[b_]Li256ELi1ELi4ELb0E",
    r"gated_mlp_bwd2_kernelIDF16[b_]Li(64|128)ELi1ELi8E",
    r"gated_mlp_bwd2_kernelIDF16[b_]Li256ELi1ELi4E",
    r"gated_mlp_wgrad_kernelIDF16[b_]Li(64|128)ELi1E",
    r"win_attn_kernelIDF16[b_]Li(64|128|256)ELi(32|64)E",
    r"win_attn_bwd_kernelIDF16[b_]Li(64|128|256)ELi(32|64)E",
    r"qkv_dwconv_gram_rows_kernelIDF16[b_]Li(64|128)ELi(32|64)ELb[01]ELb0ELb0E",
    r"gemm_tok_kernelIDF16[b_]", r"gemm_tok_ring_kernelIDF16[b_]", r"gemm_tn_tr_kernelIDF16[b_]", r"gemm_tn_tr_group_kernelIDF16[b_]",
    r"gemm_tn_ring", r"dwconv3x3_bwd_tile_kernelIDF16[b_]", r"dwconv3x3_tile_kernelIDF16[b_]", r"dwconv_gate_tile_kernel",
    r"combine_bwd_kernelIDF16[b_]", r"ln_bwd_win_kernelIDF16[b_]Li(4|8)E", r"reduce_parts_kernel", r"pg_gate_(fwd|bwd)_kernel",
    r"spectral_fold_kernelIDF16[b_]", r"spectral_fold_bwd_kernelIDF16[b_]", r"conv3x3_pipe_kernel", r"spectral_dqkv_bwd_kernelIDF16[b_]Li(32|64|128|256)E",
    r"flat_adamw_kernel", r"multi_copy_kernel", r"pack_gather_kernel", r"l1_clamp_loss_kernel",
]
# cold instantiations of the 16-bit paths at the natural widths that spill today: (pattern, ceiling).  Remote-sensing widths
# (C = 96 / 192 / 384) and fp32 are reported by tools/kernel_meta.py --spills and discussed in DESIGN.md 5.
KNOWN = [
    (r"gdfn_fused_kernelIDF16[b_]Li(128|256)ELb[01]E", 16),                # 1-2 launches per step
    (r"qkv_dwconv_gram_kernelIDF16[b_]Li256ELi32ELb0", 19),              # tile form at the latent level (W = 16): 6 launches per step
    (r"qkv_dwconv_gram_kernelIDF16[b_]Li256ELi32ELb1", 115),             # ... with the LayerNorm prologue: not launched (the row form took fusion2)
    (r"qkv_dwconv_gram_rows_kernelIDF16[b_]Li256ELi32E", 25),            # C = 256 row form (fusion2): 1 launch per step
    (r"qkv_dwconv_gram_rows_kernelIDF16[b_]Li128ELi64ELb0ELb1ELb0E", 14),  # the shader-clock-stamped diagnostic build
    (r"dwconv_gram2_kernelIDF16[b_]Li128ELi(32|64)E", 32),               # cross attention of TVSP level 2: 1 launch per step
    (r"gated_mlp_bwd2_kernelIDF16[b_]Li256ELi(1ELi8|2ELi4)E", 140),
    (r"ln_bwd_win_dxn_kernelIDF16_Li256E", 9),                               # fp16 at C = 256 (no shipped configuration trains that width in fp16)      # forms the host does not choose at C = 256 (tests only)
]


@pytest.fixture(scope="module")
def kernels():
    build = os.path.join(ROOT, "mp-hsir_amd", "build")
    if not os.path.isdir(build) or not any(f.endswith(".o") for f in os.listdir(build)):
        sys.path.insert(0, os.path.join(ROOT, "mp-hsir_amd"))
        import build as B
        B.build(verbose=False)
    import kernel_meta
    ks = kernel_meta.all_kernels(build)
    assert len(ks) > 300
    return ks


def test_hot_kernels_do_not_spill(kernels):
    for pat in HOT:
        hit = [k for k in kernels if re.search(pat, k["name"])]
        if "spectral_dqkv_bwd" in pat and not hit:
            continue
        assert hit, "no kernel matches %s (renamed? update tests/test_kernel_meta.py)" % pat
        for k in hit:
            assert k.get("vgpr_spill_count", 0) == 0 and k.get("private_segment_fixed_size", 0) == 0, \
                "%s spills %d registers (%d bytes of scratch)" % (k["demangled"], k.get("vgpr_spill_count", 0), k.get("private_segment_fixed_size", 0))


def test_no_new_spilling_instantiation_at_the_natural_widths(kernels):
    """every 16-bit kernel whose first template integer is a natural-scene width (or that has none) is spill-free or listed in KNOWN"""
    bad = []
    for k in kernels:
        n = k["name"]
        sp = k.get("vgpr_spill_count", 0)
        if sp == 0 or not re.search(r"IDF16[b_]", n):
            continue
        m = re.search(r"IDF16[b_]Li(\d+)E", n)
        if m and int(m.group(1)) not in (32, 64, 128, 256):
            continue                                  # remote-sensing widths
        lim = [c for p, c in KNOWN if re.search(p, n)]
        if not lim or sp > max(lim):
            bad.append((k["demangled"], sp))
    assert not bad, "spilling kernels not covered by KNOWN: %s" % bad
