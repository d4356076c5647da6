"""Boundary checks that need no GPU: constructor surface, state_dict keys/shapes/dtypes against the
manifest dumped from the reference, checkpoint filtering as train.py:109-116 does it."""
import pytest
import torch

from golden.cases import NATURAL_CFG, REMOTE_CFG, TINY_CFG


def _net(cfg):
    from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
    return MP_HSIR_Net(**cfg, clip_prompt="surrogate")


def test_missing_clip_is_an_error_not_a_silent_surrogate():
    """without the OpenAI `clip` package the constructor must not fall back to other text embeddings on its own
    (reference checkpoints would evaluate wrongly): it raises unless a table is injected or the surrogate is requested."""
    import importlib.util
    from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
    if importlib.util.find_spec("clip") is not None:
        pytest.skip("OpenAI clip is installed here")
    with pytest.raises(RuntimeError, match="clip_prompt"):
        MP_HSIR_Net(**TINY_CFG)
    assert MP_HSIR_Net(**TINY_CFG, clip_prompt="surrogate").text_prompt.clip_source == "surrogate"
    with pytest.raises(ValueError):
        MP_HSIR_Net(**TINY_CFG, clip_prompt="random")


@pytest.mark.parametrize("name,cfg", [("natural_mode0", NATURAL_CFG), ("remote_mode8", REMOTE_CFG), ("tiny", TINY_CFG)])
def test_state_dict_matches_reference(name, cfg, manifest):
    sd = _net(cfg).state_dict()
    man = manifest[name]
    assert set(sd) == set(man)
    for k, v in sd.items():
        assert list(v.shape) == man[k][0] and str(v.dtype).replace("torch.", "") == man[k][1], k
    if name + ":nparams" in manifest:
        assert sum(p.numel() for p in _net(cfg).parameters()) == manifest[name + ":nparams"]


def test_default_config_is_the_natural_scene_model(manifest):
    net = _net({})
    assert len(net.state_dict()) == 658 and sum(p.numel() for p in net.parameters()) == 14527484
    assert net.clip_prompts.shape == (6, 512) and net.prompts is None
    assert net.text_prompt.get_clip_prompt() is net.clip_prompts


def test_bad_task_classes_raises_like_the_reference():
    with pytest.raises(ValueError, match="task_classes must be 6 or 7"):
        _net(dict(task_classes=5))


def test_buffers_match_closed_forms():
    net = _net(TINY_CFG)
    blk = net.encoder_level1.blocks[1]
    idx = blk.attn.relative_position_index
    assert idx.dtype == torch.int64 and int(idx[0, 63]) == 0 and int(idx[63, 0]) == 224 and int(idx[5, 5]) == 112
    m = blk.attn_mask
    assert m.shape == (64, 64, 64) and set(m.unique().tolist()) == {-100.0, 0.0}
    assert int((m.abs().sum(dim=(1, 2)) > 0).sum()) == 15
    assert net.encoder_level1.blocks[0].attn_mask is None
    # stochastic-depth rates: linspace(0, 0.1, sum(num_blocks)) sliced per stage (ref :780-805)
    assert abs(net.latent.blocks[-1].drop_prob - 0.1) < 1e-7 and net.encoder_level1.blocks[0].drop_prob == 0.0


def test_partial_checkpoint_load_like_train_py():
    """train.py:109-116: keep entries whose key AND shape match, strict=False, `net.` prefix."""
    net = _net(TINY_CFG)
    ckpt = {"net." + k: torch.full_like(v, 0.5) if v.is_floating_point() else v for k, v in net.state_dict().items()}
    ckpt["net.output.weight"] = torch.zeros(3, 3)          # wrong shape -> filtered
    ckpt["net.not_a_key"] = torch.zeros(1)
    own = {"net." + k: v for k, v in net.state_dict().items()}
    filt = {k: v for k, v in ckpt.items() if k in own and own[k].shape == v.shape}
    missing, unexpected = net.load_state_dict({k[4:]: v for k, v in filt.items()}, strict=False)
    assert missing == ["output.weight"] and not unexpected
    assert float(net.patch_embed.proj.weight.mean()) == 0.5


def test_forward_without_gpu_fails_loudly():
    net = _net(TINY_CFG)
    import mp_hsir_amd._lib as L
    saved = (L._lib, L._is_emu)
    try:
        L._lib, L._is_emu = None, False
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            net(torch.rand(1, 8, 32, 32), torch.tensor([0]))
    finally:
        L._lib, L._is_emu = saved
