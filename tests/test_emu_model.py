"""Whole-network checks on CPU through the emulated kernels: the boundary module's wiring (NHWC glue,
weight packing, TVSP batch coupling, autograd plumbing) against the reference's golden outputs and
gradients.  Slow (fibers), so only the two most informative cases run here; the rest are `-m gpu`."""
import pytest

import model_checks as M
from emu import bind_emulator


@pytest.fixture(scope="module", autouse=True)
def _emu():
    bind_emulator()


def test_tiny_forward_b4_matches_reference():
    assert M.check_tiny_forward("cpu", "t32_b4") < 2e-6


@pytest.mark.skipif(not __import__("os").environ.get("MPHSIR_SLOW_TESTS"), reason="3 minutes on the emulator; runs on the GPU in "
                    "tests/test_gpu_model.py::test_tiny_net_gradients_fp32 (set MPHSIR_SLOW_TESTS=1 to run here)")
def test_tiny_gradients_match_reference():
    assert M.check_tiny_gradients("cpu") < 1e-4


@pytest.mark.parametrize("name", ["nat_enc1", "rs_latent", "tvsp_l2", "fusion_l2"])
def test_block_gradients_match_reference(name):
    """stand-alone modules (PGSSTB C=64 / 2 heads / shifted and C=384 / 8 heads of 48, TVSP, PromptFusion): the HIP
    backward kernels vs the reference's dx / dparams (the other shape classes run on the GPU, tests/test_gpu_model.py)."""
    assert M.check_block_gradients("cpu", name, tol=2e-5) < 2e-5


def test_block_gradients_bf16_widest():
    import torch
    worst, ratio, med = M.check_block_gradients("cpu", "rs_latent", torch.bfloat16, tol=None)      # per-tensor bars (model_checks.block_grad_bars)
    print("worst rel-L2 %.3g, worst error / bar %.2f" % (worst, ratio))


def test_fused_block_equals_unfused_bf16():
    """the whole-block Function (branch sum inside the gated-MLP launch) == attention Function + MLP Function, bit for bit"""
    M.check_fused_block_equals_unfused("cpu", "nat_enc1")


def test_pack_plan_matches_per_module_packers():
    import torch
    M.check_pack_plan("cpu")
    M.check_pack_plan("cpu", torch.bfloat16, steps=2)
