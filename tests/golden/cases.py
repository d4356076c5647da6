"""Case tables shared by make_golden.py (generator, runs the reference) and the tests (checkers).

Test infrastructure only.  Shapes are chosen to pin the reference quirks listed in SURVEY.md §7.4:
Q1 (TVSP batch coupling, B=4 vs prompt size), Q8 (mask rebuilt when (H,W) != input_resolution),
Q15 (bilinear prompt upsample at 128x128), Q16 (H,W multiples of 32), Q17 (1-D vs 2-D task ids).
"""
import zlib

import torch

TINY_CFG = dict(in_channel=8, out_channel=8, dim=32, num_blocks=[2, 2, 2], num_refinement_blocks=2,
                heads=[1, 2, 4], task_classes=6)

TINY_CASES = {
    "t64_b1": dict(shape=(1, 8, 64, 64), task=[0]),
    "t64_b2_ids2d": dict(shape=(2, 8, 64, 64), task=[[1], [3]]),
    "t32_b4": dict(shape=(4, 8, 32, 32), task=[0, 2, 4, 5]),
    "t96x64_b1": dict(shape=(1, 8, 96, 64), task=[2]),
    "t128_b1": dict(shape=(1, 8, 128, 128), task=[5]),
    "t64_b2_T7": dict(shape=(2, 8, 64, 64), task=[5, 6], task_classes=7),
    "t64_b1_T1": dict(shape=(1, 8, 64, 64), task=[0], task_classes=1),
    # the benchmark's batch size: TVSP's text map depends on B (clip[floor(i*B/ps)]).  Stored: samples `keep`
    # in full + the per-sample norms of all 32 outputs.
    "t32_b32": dict(shape=(32, 8, 32, 32), task=[i % 6 for i in range(32)], keep=[0, 13, 31]),
}

# One PGSSTB per shape class of both shipped configurations (SURVEY §8d stage table), plus the
# prompt modules at natural level-2 width.
BLOCK_CASES = {
    "nat_enc1": dict(man="natural_mode0", prefix="encoder_level1.blocks.1.", kind="pgsstb", C=64, heads=2, cr=8, shift=4, shape=(2, 64, 32, 32), intermediates=True, grad=True, grad_full=True),
    "nat_enc2": dict(man="natural_mode0", prefix="encoder_level2.blocks.0.", kind="pgsstb", C=128, heads=4, cr=16, shift=0, shape=(1, 128, 32, 32), grad=True),
    "nat_latent": dict(man="natural_mode0", prefix="latent.blocks.1.", kind="pgsstb", C=256, heads=8, cr=32, shift=4, shape=(1, 256, 16, 16), grad=True),
    "nat_refine": dict(man="natural_mode0", prefix="refinement.blocks.1.", kind="pgsstb", C=128, heads=2, cr=8, shift=4, shape=(1, 128, 32, 32), grad=True),
    "rs_enc1": dict(man="remote_mode8", prefix="encoder_level1.blocks.1.", kind="pgsstb", C=96, heads=2, cr=8, shift=4, shape=(1, 96, 32, 32), grad=True),
    "rs_enc2": dict(man="remote_mode8", prefix="encoder_level2.blocks.0.", kind="pgsstb", C=192, heads=4, cr=16, shift=0, shape=(1, 192, 32, 32), grad=True),
    "rs_latent": dict(man="remote_mode8", prefix="latent.blocks.1.", kind="pgsstb", C=384, heads=8, cr=32, shift=4, shape=(1, 384, 16, 16), grad=True),
    "rs_refine": dict(man="remote_mode8", prefix="refinement.blocks.1.", kind="pgsstb", C=192, heads=2, cr=8, shift=4, shape=(1, 192, 32, 32), grad=True),
    "tvsp_l2": dict(man="natural_mode0", prefix="prompt2.", kind="tvsp", T=6, ps=32, D=128, shape=(2, 128, 32, 32), task=[0, 3], grad=True),
    "fusion_l2": dict(man="natural_mode0", prefix="fusion2.", kind="fusion", D=128, heads=8, shape=(1, 128, 32, 32), grad=True),
}

NATURAL_CFG = dict(in_channel=31, out_channel=31, dim=64, task_classes=6)       # test.py:39
REMOTE_CFG = dict(in_channel=100, out_channel=100, dim=96, task_classes=7)      # train.py:45

FULL_CASES = {
    # BASELINE.json configs[0]: single 64x64x31 patch, Gaussian denoise mode 0 (sigma 70), task id 0
    "natural_mode0": dict(cfg=NATURAL_CFG, shape=(1, 31, 64, 64), recipe="gaussian70", task=[0]),
    # remote-sensing model, inpaint mode 8 recipe (mask ratio 0.9), task id 4
    "remote_mode8": dict(cfg=REMOTE_CFG, shape=(1, 100, 64, 64), recipe="inpaint90", task=[4]),
}

# Whole-net gradients at REAL width (train.py:58-67 over net/MP_HSIR.py:810-844): both shipped configurations, batch 2 of 64x64
# patches, the training recipe (clean + Gaussian noise, L1 after clamp), eval mode (DropPath = identity).  full_grad.npz holds,
# from the REFERENCE run in fp64: the loss, and per parameter gradient its norm + FULLGRAD_SAMPLES seeded samples (tensors up to
# that many elements: in full), and the names of the parameters autograd leaves without gradient.
FULLGRAD_CASES = {
    "natural_b2": dict(cfg=NATURAL_CFG, shape=(2, 31, 64, 64), task=[0, 3], sigma=50.0, autocast=(torch.bfloat16, 1.0)),
    "remote_b2": dict(cfg=REMOTE_CFG, shape=(2, 100, 64, 64), task=[4, 6], sigma=30.0, autocast=(torch.float16, 1024.0)),
}
FULLGRAD_SAMPLES = 4096


def fullgrad_inputs(name):
    from golden.detfill import seeded_input
    c = FULLGRAD_CASES[name]
    clean = seeded_input(name + ":clean", c["shape"])
    degraded = clean + seeded_input(name + ":noise", c["shape"], "normal") * (c["sigma"] / 255.0)      # degradation_utils.py:25-31
    return c, clean, degraded


# Full-SIZE cubes: the reference run once per case in fp32 on the shapes test.py feeds the model
# (test.py:150-188: 512x512x31 natural cubes; :440-469 / BASELINE configs[3]: the 172-band remote-sensing width), stored as
# summary statistics -- norm, mean, CUBE_SAMPLES seeded samples, per-band means, PSNR -- because the tensors are 32-180 MB.
RS172_CFG = dict(in_channel=172, out_channel=172, dim=96, task_classes=7)
CUBE_CASES = {
    # min_rows / min_gdfn: launches of the row-walking pass A / the fused GDFN the 16-bit forward must take at that size
    "nat512": dict(cfg=NATURAL_CFG, shape=(1, 31, 512, 512), recipe="gaussian70", task=[0], min_rows=16, min_gdfn=2),
    "rs172_256": dict(cfg=RS172_CFG, shape=(1, 172, 256, 256), recipe="inpaint90", task=[4], min_rows=10, min_gdfn=1),
}
CUBE_SAMPLES = 4096


def cube_inputs(name):
    from golden.detfill import seeded_input
    c = CUBE_CASES[name]
    clean = seeded_input(name + ":clean", c["shape"])
    if c["recipe"] == "gaussian70":          # dataset_utils.py:293-298, test.py:554
        degraded = clean + seeded_input(name + ":noise", c["shape"], "normal") * (70.0 / 255.0)
    else:                                    # inpaint, dataset_utils.py:743-749: keep where rand > ratio (0.9)
        degraded = clean * (seeded_input(name + ":mask", c["shape"]) > 0.9).float()
    return c, clean, degraded


# parameter-gradient tensors stored in full in tiny_grad.npz (everything else: norm/sum/samples)
GRAD_KEYS_FULL = ("encoder_level1.blocks.1.", "prompt1.", "fusion1.", "patch_embed.", "output.",
                  "reduce_chan_level2.", "down1_2.", "up2_1.")


# block gradients (blocks.npz): tensors up to this many elements are stored in full, larger ones as their norm plus
# GRAD_SAMPLES seeded samples (the fixture stays a few MB; full-tensor checks of the large ones run against the
# oracle's fp64 autograd, which these fixtures pin)
GRAD_FULL_MAX = 32768
GRAD_SAMPLES = 4096


def sample_indices(key, numel, n=12):
    g = torch.Generator().manual_seed(zlib.crc32(("samp:" + key).encode()) & 0x7FFFFFFF)
    return torch.randint(0, numel, (min(n, numel),), generator=g)


def cotangent(name, shape):
    g = torch.Generator().manual_seed(zlib.crc32(("cot:" + name).encode()) & 0x7FFFFFFF)
    return torch.randn(tuple(shape), generator=g, dtype=torch.float32)
