"""Whole-network checks shared by the emulator (CPU) and GPU test files."""
import json
import os

import numpy as np
import torch

from golden.cases import TINY_CFG, TINY_CASES, FULL_CASES
from golden.detfill import det_fill_, seeded_input, surrogate_clip_prompt
from util import rel_l2

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build_net(cfg, dev, dtype=torch.float32):
    from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
    net = MP_HSIR_Net(**cfg, clip_prompt=surrogate_clip_prompt(cfg["task_classes"])).eval()
    det_fill_(net)
    return net.to(dev).set_compute_dtype(dtype)


def check_tiny_forward(dev, name, dtype=torch.float32, tol=2e-5):
    case = TINY_CASES[name]
    cfg = dict(TINY_CFG, task_classes=case.get("task_classes", TINY_CFG["task_classes"]))
    net = build_net(cfg, dev, dtype)
    x = seeded_input(name, case["shape"]).to(dev)
    with torch.no_grad():
        y = net(x, torch.tensor(case["task"]).to(dev))
    g = np.load(os.path.join(GOLDEN, "tiny_fwd.npz"))
    want = g[name + "/out"]
    if "keep" in case:
        assert rel_l2(y.float().cpu().flatten(1).norm(dim=1), g[name + "/norms"]) < tol, name
        y = y[case["keep"]]
    err = rel_l2(y.float().cpu(), want)
    assert err < tol, (name, err)
    return err


def full_case_inputs(name):
    c = FULL_CASES[name]
    clean = seeded_input(name + ":clean", c["shape"])
    if c["recipe"] == "gaussian70":
        degraded = clean + seeded_input(name + ":noise", c["shape"], "normal") * (70.0 / 255.0)
    else:
        degraded = clean * (seeded_input(name + ":mask", c["shape"]) > 0.9).float()
    return c, clean, degraded


def psnr(restored, clean):
    r = restored.detach().double().cpu().clamp(0, 1)
    c = clean.detach().double().cpu().clamp(0, 1)
    mse = ((r - c) ** 2).mean(dim=(-1, -2))
    return float((10.0 * torch.log10(1.0 / mse)).mean(dim=1).mean())


def check_full_forward(dev, name, dtype=torch.float32, tol=1e-3, dpsnr=0.01):
    c, clean, degraded = full_case_inputs(name)
    net = build_net(c["cfg"], dev, dtype)
    with torch.no_grad():
        y = net(degraded.to(dev), torch.tensor(c["task"]).to(dev))
    g = np.load(os.path.join(GOLDEN, "full.npz"))
    err = rel_l2(y.float().cpu(), g[name + "/out"])
    dp = abs(psnr(y, clean) - float(g[name + "/psnr_restored"]))
    assert err < tol and dp < dpsnr, (name, err, dp)
    return err, dp


def check_full_size_cube(dev, name, oracle_threads=32):
    """Full-SIZE parity (test.py:150-188 / :440-469 shapes): the HIP forward of one whole cube
      (1) fp32 and bf16 against summary statistics of the REFERENCE run at that size (tests/golden/cubes.npz: norm, mean,
          4096 seeded samples, per-band norms, PSNR), and
      (2) fp32 and bf16 against the fp32 oracle run here on the host, as full tensors: rel-L2 < 1e-3 and dPSNR < 0.01 dB for
          fp32 (the north_star bar), < 4e-2 for bf16;
    and asserts that the 16-bit forward ran the forms that only exist at this size: the row-walking pass A, the fused GDFN,
    the pre-reduced Gram partials (> 64 slots), the host-free shift mask at thousands of windows."""
    import time
    from golden.cases import CUBE_SAMPLES, cube_inputs, sample_indices
    from mp_hsir_amd import ops
    from oracle import mp_hsir_oracle as O
    c, clean, degraded = cube_inputs(name)
    g = np.load(os.path.join(GOLDEN, "cubes.npz"))
    net = build_net(c["cfg"], dev, torch.float32)
    task = torch.tensor(c["task"])
    x = degraded.to(dev)
    with torch.no_grad():
        y32 = net(x, task.to(dev)).float().cpu()
        net.set_compute_dtype(torch.bfloat16)
        ops.ACCOUNT = {}
        try:
            y16 = net(x, task.to(dev)).float().cpu()
        finally:
            acct, ops.ACCOUNT = ops.ACCOUNT, None
    forms = {k: v[0] for k, v in acct.items()}
    res = {"forms": {k: forms.get(k, 0) for k in ("qkv_dwconv_gram:rows", "qkv_dwconv_gram:tile", "gdfn_fused", "dwconv_gram")}}
    if dev != "cpu":
        assert forms.get("qkv_dwconv_gram:rows", 0) >= c["min_rows"] and forms.get("gdfn_fused", 0) >= c["min_gdfn"], forms
    # (1) the reference's own statistics
    idx = sample_indices("cube:" + name, y32.numel(), CUBE_SAMPLES)
    want_s, want_n = torch.from_numpy(g[name + "/samples"]), float(g[name + "/norm"])
    for tag, y, tol, dps in (("f32", y32, 1e-3, 0.01), ("bf16", y16, 4e-2, 0.25)):
        e_s = rel_l2(y.flatten()[idx], want_s)
        e_n = abs(float(y.double().norm()) - want_n) / want_n
        e_b = rel_l2(y.double().flatten(2).norm(dim=2)[0], g[name + "/band_norms"])
        dp = abs(psnr(y, clean) - float(g[name + "/psnr_restored"]))
        res["ref_" + tag] = dict(samples=e_s, norm=e_n, band_norms=e_b, dpsnr=dp)
        assert e_s < tol and e_n < tol and e_b < tol and dp < dps, (name, tag, res)
    # (2) the oracle on this host, full tensors
    P = {k: v.detach().cpu().float() for k, v in net.state_dict().items() if not k.endswith("attn_mask")}
    prev = torch.get_num_threads()
    torch.set_num_threads(max(1, min(oracle_threads, os.cpu_count() or 1)))
    t0 = time.time()
    try:
        with torch.no_grad():
            want = O.mp_hsir_forward(P, O.make_cfg(**c["cfg"]), degraded, task, surrogate_clip_prompt(c["cfg"]["task_classes"]))
    finally:
        torch.set_num_threads(prev)
    res["oracle_seconds"] = time.time() - t0
    res["oracle_vs_ref_samples"] = rel_l2(want.flatten()[idx], want_s)
    assert res["oracle_vs_ref_samples"] < 1e-4, res
    for tag, y, tol, dps in (("f32", y32, 1e-3, 0.01), ("bf16", y16, 4e-2, 0.25)):
        e = rel_l2(y, want)
        dp = abs(psnr(y, clean) - psnr(want, clean))
        res["oracle_" + tag] = dict(rel_l2=e, dpsnr=dp)
        assert e < tol and dp < dps, (name, tag, res)
    return res


def check_b16_forward(dev, oracle_threads=32):
    """BASELINE configs[1] at real width: the natural-scene net, batch 16 of 64x64x31 patches with six different task ids (TVSP's batch
    coupling, SURVEY Q1, couples the 16 samples: `clip[floor(i*B/ps)]`), forward only -- fp32 and bf16 HIP outputs against the fp32
    oracle run on this host, as full tensors: rel-L2 < 1e-3 / dPSNR < 0.01 dB (fp32, the north_star bar), < 4e-2 / 0.25 dB (bf16)."""
    import time
    from oracle import mp_hsir_oracle as O
    c, clean1, degraded1 = full_case_inputs("natural_mode0")
    shape = (16,) + tuple(c["shape"][1:])
    clean = seeded_input("b16:clean", shape)
    degraded = clean + seeded_input("b16:noise", shape, "normal") * (70.0 / 255.0)
    task = torch.arange(16) % 6
    net = build_net(c["cfg"], dev, torch.float32)
    x = degraded.to(dev)
    with torch.no_grad():
        y32 = net(x, task.to(dev)).float().cpu()
        net.set_compute_dtype(torch.bfloat16)
        y16 = net(x, task.to(dev)).float().cpu()
    P = {k: v.detach().cpu().float() for k, v in net.state_dict().items() if not k.endswith("attn_mask")}
    prev = torch.get_num_threads()
    torch.set_num_threads(max(1, min(oracle_threads, os.cpu_count() or 1)))
    t0 = time.time()
    try:
        with torch.no_grad():
            want = O.mp_hsir_forward(P, O.make_cfg(**c["cfg"]), degraded, task, surrogate_clip_prompt(c["cfg"]["task_classes"]))
    finally:
        torch.set_num_threads(prev)
    res = {"oracle_seconds": time.time() - t0}
    for tag, y, tol, dps in (("f32", y32, 1e-3, 0.01), ("bf16", y16, 4e-2, 0.25)):
        e = rel_l2(y, want)
        worst = max(rel_l2(y[b], want[b]) for b in range(16))                # per sample: no sample hides behind the others
        dp = abs(psnr(y, clean) - psnr(want, clean))
        res[tag] = dict(rel_l2=e, worst_sample=worst, dpsnr=dp)
        assert e < tol and worst < 2 * tol and dp < dps, res
    return res


def check_tiny_gradients(dev, dtype=torch.float32, tol=1e-4):
    """L1-after-clamp loss and every parameter gradient of the tiny net vs the reference (tiny_grad.npz)."""
    from golden.cases import GRAD_KEYS_FULL, sample_indices
    net = build_net(TINY_CFG, dev, dtype)
    x = seeded_input("grad_x", (2, 8, 64, 64)).to(dev)
    clean = seeded_input("grad_clean", (2, 8, 64, 64)).to(dev)
    task = torch.tensor([[1], [3]]).to(dev)
    y = net(x, task)
    loss = (y.clamp(0, 1) - clean).abs().mean()
    loss.backward()
    g = np.load(os.path.join(GOLDEN, "tiny_grad.npz"))
    assert abs(float(loss) - float(g["loss"])) < tol * 10, (float(loss), float(g["loss"]))
    none_keys = set(str(k) for k in g["none_keys"])
    worst = 0.0
    for k, p in net.named_parameters():
        if k in none_keys:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        want = float(g["norm/" + k])
        err = abs(float(p.grad.float().norm()) - want) / (want + 1e-20)
        if any(k.startswith(pref) for pref in GRAD_KEYS_FULL):
            err = max(err, rel_l2(p.grad.float().cpu(), g["full/" + k]))
        worst = max(worst, err)
        assert err < tol, (k, err)
    return worst


def oracle_step_gradients(P, cfg, degraded, clean, task, checkpoint_blocks=False):
    """loss = L1(clamp(net(degraded), 0, 1), clean) (train.py:58-61) through the oracle, and d loss / d P[k] for every floating
    entry of P (None where autograd leaves none: SURVEY Q3).  checkpoint_blocks: every PGSSTB block is recomputed in the backward
    (torch.utils.checkpoint around oracle.pgsstb) -- a batch of 32 at real width then needs a few GB instead of tens."""
    from oracle import mp_hsir_oracle as O
    keys = [k for k, v in P.items() if v.dtype.is_floating_point]
    Pg = {k: (v.detach().clone().requires_grad_(True) if k in keys else v) for k, v in P.items()}
    saved = O.pgsstb
    if checkpoint_blocks:
        from torch.utils.checkpoint import checkpoint

        def pgsstb_ck(Pd, pre, x, heads, shifted, keep=None, intermediates=None):
            return checkpoint(lambda t: saved(Pd, pre, t, heads, shifted, keep), x, use_reentrant=False)
        O.pgsstb = pgsstb_ck
    try:
        y = O.mp_hsir_forward(Pg, O.make_cfg(**cfg), degraded, task, surrogate_clip_prompt(cfg["task_classes"]).to(degraded.dtype))
        loss = O.l1_after_clamp(y, clean)
        grads = torch.autograd.grad(loss, [Pg[k] for k in keys], allow_unused=True)
    finally:
        O.pgsstb = saved
    return float(loss.detach()), y.detach(), dict(zip(keys, grads))


def fixture_grad_err(got, g, name, k):
    """relative error of a parameter gradient against full_grad.npz: |norm - want| / want and rel-L2 over the stored samples
    (tensors up to FULLGRAD_SAMPLES elements are stored in full)"""
    from golden.cases import FULLGRAD_SAMPLES, sample_indices
    got = torch.as_tensor(got).detach().double().cpu()
    want_n = float(g["%s/norm/%s" % (name, k)])
    e = abs(float(got.norm()) - want_n) / (want_n + 1e-300)
    if "%s/full/%s" % (name, k) in g.files:
        return max(e, rel_l2(got, g["%s/full/%s" % (name, k)]))
    idx = sample_indices(name + ":" + k, got.numel(), FULLGRAD_SAMPLES)
    return max(e, rel_l2(got.flatten()[idx], g["%s/samp/%s" % (name, k)]))


# 16-bit gradient bars (whole net).  The fixture holds, per parameter tensor, the deviation of the REFERENCE's own mixed-precision step
# (torch.autocast, train.py:118 trains in 16-mixed) from its fp64 gradients: 2 % in the median for bf16, 8-10 % across the prompt
# modules, 20-45 % for norm1.bias / prompt_param, and 100 % for the fp16 gradient of linear_prompt.weight (it underflows there).
# The 16-bit HIP backward is held, tensor by tensor, to GRAD_BAR_FACTOR (3) x that deviation plus GRAD_BAR_FLOOR (2) x the median deviation
# over all tensors (a tensor on which the reference was lucky must not fail ours for being average).
# The parameters of the spectral-prompt gate (local_spectral_attn.*) are the exception to "tensor by tensor": their gradients are sums
# over a handful of windows per sample (4 at the latent level) of products of small numbers, and which BLOCK's sum happens to cancel
# is luck -- the reference's own deviation for kv.weight runs from 0.4 % to 17 % across the 22 blocks of one net.  They are held to
# the reference's WORST block of the same tensor type instead of the same block.
GRAD_BAR_FACTOR = 3.0
GRAD_BAR_FLOOR = 2.0


def grad_bars(g, name):
    import re
    ref = {k.split("/autocast_err/")[1]: float(g[k]) for k in g.files if k.startswith(name + "/autocast_err/")}
    med = sorted(ref.values())[len(ref) // 2]
    worst_of_type = {}
    for k, v in ref.items():
        if ".local_spectral_attn." in k:
            t = re.sub(r"^.*\.local_spectral_attn\.", "", k)
            worst_of_type[t] = max(worst_of_type.get(t, 0.0), v)
    bars = {}
    for k, v in ref.items():
        if ".local_spectral_attn." in k:
            v = worst_of_type[re.sub(r"^.*\.local_spectral_attn\.", "", k)]
        bars[k] = GRAD_BAR_FACTOR * v + GRAD_BAR_FLOOR * med
    return bars, med


def _hip_step_gradients(net, x, clean, task, scale=1.0):
    """scale: the loss scale of an fp16 step (the engine scales by 65536 and unscales in the optimizer kernel: d loss / d y = 1 / numel
    ~ 1e-6 is below fp16's normal range): gradients are returned unscaled"""
    for p in net.parameters():
        p.grad = None
    y = net(x, task)
    loss = (y.clamp(0, 1) - clean).abs().mean()
    (loss * scale).backward()
    return float(loss.detach()), y.detach().float().cpu(), {k: (None if p.grad is None else p.grad.detach().float().cpu() / scale) for k, p in net.named_parameters()}


def check_full_gradients(dev, name, low=torch.bfloat16, oracle_threads=32):
    """Whole-net gradient parity at REAL width (train.py:58-67 over net/MP_HSIR.py:810-844; golden.cases.FULLGRAD_CASES): batch 2 of
    64x64 patches through MP_HSIR_Net(31,31,64,T=6) / (100,100,96,T=7), L1-after-clamp loss, every parameter gradient of
      (1) the fp32 HIP backward against the REFERENCE's (full_grad.npz: norm + 4096 samples per tensor) and against the oracle's fp64
          autograd on this host as full tensors: rel-L2 < 1e-4; the parameters autograd leaves without gradient stay untouched;
      (2) the 16-bit HIP backward against the same fp64 gradients with the per-tensor bar of grad_bar().
    Returns a dict of the worst errors."""
    from golden.cases import fullgrad_inputs
    c, clean, degraded = fullgrad_inputs(name)
    g = np.load(os.path.join(GOLDEN, "full_grad.npz"))
    net = build_net(c["cfg"], dev, torch.float32)
    x, cl, task = degraded.to(dev), clean.to(dev), torch.tensor(c["task"]).to(dev)
    loss32, y32, g32 = _hip_step_gradients(net, x, cl, task)
    none_keys = set(str(k) for k in g[name + "/none_keys"])
    res = {"loss_f32": abs(loss32 - float(g[name + "/loss"])) / float(g[name + "/loss"])}
    assert res["loss_f32"] < 1e-5, res
    # (1a) the reference's fixture
    worst = ("", 0.0)
    for k, gr in g32.items():
        if k in none_keys:
            assert gr is None or float(gr.abs().max()) == 0.0, k
            continue
        assert gr is not None, k
        e = fixture_grad_err(gr, g, name, k)
        if e > worst[1]:
            worst = (k, e)
        assert e < 1e-4, (name, "fp32 vs reference fixture", k, e)
    res["f32_vs_reference"] = worst
    # (1b) the oracle's fp64 autograd, full tensors
    P = {k: (v.detach().cpu().double() if v.dtype.is_floating_point else v.detach().cpu()) for k, v in net.state_dict().items() if not k.endswith("attn_mask")}
    prev = torch.get_num_threads()
    torch.set_num_threads(max(1, min(oracle_threads, os.cpu_count() or 1)))
    try:
        loss64, y64, g64 = oracle_step_gradients(P, c["cfg"], degraded.double(), clean.double(), torch.tensor(c["task"]))
    finally:
        torch.set_num_threads(prev)
    assert abs(loss64 - float(g[name + "/loss"])) < 1e-9 * abs(loss64) + 1e-12
    worst = ("", 0.0)
    for k, gr in g32.items():
        if k in none_keys:
            assert g64.get(k) is None, k
            continue
        e = rel_l2(gr, g64[k])
        if e > worst[1]:
            worst = (k, e)
        assert e < 1e-4, (name, "fp32 vs oracle fp64", k, e)
    res["f32_vs_oracle"] = worst
    # (2) 16-bit storage
    net.set_compute_dtype(low)
    loss16, y16, g16 = _hip_step_gradients(net, x, cl, task, scale=65536.0 if low == torch.float16 else 1.0)
    bars, med = grad_bars(g, name)
    res["e_fwd"] = rel_l2(y16, y32)
    res["e_fwd_reference_autocast"] = float(g[name + "/autocast_out_err"])
    assert res["e_fwd"] < 2.0 * res["e_fwd_reference_autocast"] + 1e-3 and abs(loss16 - loss64) < 2e-2 * abs(loss64), (res, loss16, loss64)
    ratios, bad = [], []
    for k, gr in g16.items():
        if k in none_keys:
            assert gr is None or float(gr.abs().max()) == 0.0, k
            continue
        e = rel_l2(gr, g64[k])
        ratios.append((e / bars[k], k, e))
        if not e < bars[k]:
            bad.append((k, e, bars[k]))
    ratios.sort(reverse=True)
    errs = sorted(e for _, _, e in ratios)
    res["low_closest_to_bar"] = [(round(r, 2), k, round(e, 4)) for r, k, e in ratios[:6]]
    res["low_median_err"], res["reference_autocast_median_err"] = errs[len(errs) // 2], med
    print(name, str(low), res)
    assert not bad, (name, str(low), bad[:10])
    return res


def check_batch32_step(dev, low=torch.bfloat16, n_sampled=16, oracle_threads=64):
    """The benchmark's step shape -- natural net, batch 32 of 64x64x31 patches, bf16 storage (BASELINE configs[2]) -- as ONE
    forward + backward of the HIP path against the oracle run on this host in fp32 with per-block recomputation: the loss, and the
    full gradient tensors of n_sampled seeded parameters (plus the first and the last layer), bar grad_bar()."""
    import zlib
    from golden.cases import NATURAL_CFG
    B = 32
    clean = seeded_input("b32:clean", (B, 31, 64, 64))
    sig = (30.0 + 40.0 * seeded_input("b32:sigma", (B, 1, 1, 1))) / 255.0            # sigma ~ U(30, 70) / 255: degradation_utils.py:25-31
    degraded = clean + seeded_input("b32:noise", (B, 31, 64, 64), "normal") * sig
    task = torch.arange(B) % 6
    net = build_net(NATURAL_CFG, dev, torch.float32)
    x, cl = degraded.to(dev), clean.to(dev)
    with torch.no_grad():
        y32 = net(x, task.to(dev)).float().cpu()
    net.set_compute_dtype(low)
    loss16, y16, g16 = _hip_step_gradients(net, x, cl, task.to(dev))
    e_fwd = rel_l2(y16, y32)
    P = {k: (v.detach().cpu().float() if v.dtype.is_floating_point else v.detach().cpu()) for k, v in net.state_dict().items() if not k.endswith("attn_mask")}
    prev = torch.get_num_threads()
    torch.set_num_threads(max(1, min(oracle_threads, os.cpu_count() or 1)))
    try:
        loss_o, y_o, g_o = oracle_step_gradients(P, NATURAL_CFG, degraded, clean, task, checkpoint_blocks=True)
    finally:
        torch.set_num_threads(prev)
    res = {"e_fwd": e_fwd, "fwd_f32_vs_oracle": rel_l2(y32, y_o), "loss": (loss16, loss_o)}
    g = np.load(os.path.join(GOLDEN, "full_grad.npz"))
    bars, med = grad_bars(g, "natural_b2")            # the same net: the reference's own bf16 deviation per tensor, measured at batch 2
    assert res["fwd_f32_vs_oracle"] < 1e-4 and e_fwd < 2.0 * float(g["natural_b2/autocast_out_err"]) + 1e-3 and abs(loss16 - loss_o) < 2e-2 * abs(loss_o), res
    names = sorted(k for k, v in g16.items() if v is not None)
    rng = torch.Generator().manual_seed(zlib.crc32(b"b32:params"))
    pick = [names[i] for i in torch.randperm(len(names), generator=rng)[:n_sampled].tolist()] + ["patch_embed.proj.weight", "output.weight"]
    worst = []
    for k in pick:
        e = rel_l2(g16[k], g_o[k])
        worst.append((round(e / bars[k], 2), k, round(e, 4)))
        assert e < bars[k], (k, e, bars[k])
    res["closest_to_bar"] = sorted(worst, reverse=True)[:6]
    return res


_LIB_GEMM_OPS = ("aten::mm", "aten::addmm", "aten::bmm", "aten::baddbmm", "aten::matmul", "aten::linear", "aten::convolution",
                 "aten::_convolution", "aten::conv2d", "aten::native_layer_norm", "aten::layer_norm", "aten::_softmax")


def _run_profiled(fn):
    """run fn under the torch profiler and return the set of aten op names it dispatched (forward and backward)."""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        out = fn()
    return out, {e.key for e in prof.key_averages()}


def block_grad_bars(name):
    """bf16 bars of one block fixture, tensor by tensor: GRAD_BAR_FACTOR x the deviation of the REFERENCE's own bf16 autocast backward from
    its fp64 gradients for that tensor (tests/golden/blocks_autocast.npz, make_golden.py block_autocast) + GRAD_BAR_FLOOR x the median
    over the block's tensors -- the rule of the whole-net check (grad_bars) instead of one flat 6e-2.  The spectral-prompt gate's
    parameters are held to the worst tensor of the gate in that block (sums over 64 windows whose cancellation is luck, see grad_bars)."""
    g = np.load(os.path.join(GOLDEN, "blocks_autocast.npz"))
    ref = {k.split("/bf16/")[1]: float(g[k]) for k in g.files if k.startswith(name + "/bf16/")}
    out_err = ref.pop("out")
    med = sorted(ref.values())[len(ref) // 2]
    worst_pg = max([v for k, v in ref.items() if "local_spectral_attn." in k] or [0.0])
    bars = {k: GRAD_BAR_FACTOR * (worst_pg if "local_spectral_attn." in k else v) + GRAD_BAR_FLOOR * med for k, v in ref.items()}
    bars["out"] = GRAD_BAR_FACTOR * out_err + 2e-3          # the output: the reference's forward deviation + the bf16 rounding of the result itself
    return bars, med


def check_block_gradients(dev, name="nat_enc1", dtype=torch.float32, tol=2e-4):
    """One stand-alone module of golden blocks.npz (a PGSSTB of each shape class of both shipped configurations, the
    level-2 TVSP, the level-2 PromptFusion): output, input gradient(s) and every parameter gradient of the HIP path
    against the REFERENCE's (fixtures from tests/golden/make_golden.py).  Also asserts that forward + backward ran the
    HIP kernels and dispatched no library GEMM / conv / norm op."""
    from golden.cases import BLOCK_CASES, cotangent
    from golden.detfill import det_value
    from mp_hsir_amd import ops
    from mp_hsir_amd.net.MP_HSIR import PGSSTB, TVSP, PromptFusion
    from util import golden_grad_err
    import torch.nn.functional as F
    c = BLOCK_CASES[name]
    kind = c["kind"]
    if kind == "pgsstb":
        mod = PGSSTB(c["C"], c["heads"], [64, 64], 8, c["shift"], 0.0, 2.66, c["cr"], 128)
    elif kind == "tvsp":
        mod = TVSP(c["T"], c["ps"], c["D"], c["D"])
    else:
        mod = PromptFusion(c["D"] * 2, c["D"], c["heads"], 2.66, False)
    mod = mod.eval()
    with torch.no_grad():
        for k, p in mod.named_parameters():
            p.copy_(det_value(k, p.shape).float())
    mod = mod.to(dev)

    def nhwc(t):
        return t.to(dev).permute(0, 2, 3, 1).contiguous().to(dtype)
    ins = {}
    if kind == "pgsstb":
        ins["dx"] = nhwc(seeded_input(name, c["shape"], "normal")).requires_grad_(True)
        run = lambda: mod(ins["dx"])
        want_kernels = ("win_attn", "win_attn_bwd", "combine_bwd", "ln_bwd_win", "gated_mlp_bwd", "dwconv3x3_bwd|dwconv3x3_wgrad|spectral_dqkv_bwd", "pg_gate_bwd",
                        "spectral_fold_bwd", "gemm_tn")
    elif kind == "tvsp":
        x = nhwc(seeded_input(name, c["shape"], "normal"))
        w = F.one_hot(torch.tensor(c["task"]), c["T"])
        clip = (w.unsqueeze(-1) * surrogate_clip_prompt(c["T"]).unsqueeze(0)).mean(1).to(dev)
        run = lambda: mod(x, clip, w.to(dev))
        want_kernels = ("dwconv_gram", "spectral_fold_bwd", "gdfn_gate_bwd", "ln_bwd_win", "conv3x3_tok")
    else:
        ins["dx"] = nhwc(seeded_input(name + ":x", c["shape"], "normal")).requires_grad_(True)
        ins["dprompt"] = nhwc(seeded_input(name + ":p", c["shape"], "normal")).requires_grad_(True)
        run = lambda: mod(ins["dx"], ins["dprompt"])
        want_kernels = ("dwconv_gram|qkv_dwconv_gram", "spectral_fold_bwd", "gdfn_gate_bwd", "ln_bwd_win")     # a|b: either form of pass A
    cot = cotangent(name, c["shape"]).to(dev).permute(0, 2, 3, 1).to(dtype)

    def fwd_bwd():
        y = run()
        (y * cot).sum().backward()
        return y
    ops.ACCOUNT = {}
    y, aten = _run_profiled(fwd_bwd)
    acct, ops.ACCOUNT = ops.ACCOUNT, None
    for kname in want_kernels:
        assert any(k in acct for k in kname.split("|")), "%s: forward/backward did not run the HIP kernel %s" % (name, kname)
    lib_ops = sorted(o for o in aten if o in _LIB_GEMM_OPS)
    assert not lib_ops, "%s: library ops on the hot path: %s" % (name, lib_ops)
    g = np.load(os.path.join(GOLDEN, "blocks.npz"))
    errs = {"out": rel_l2(y.detach().float().cpu().permute(0, 3, 1, 2), g[name + "/out"])}
    for key, t in ins.items():
        errs[key] = golden_grad_err(t.grad.float().cpu().permute(0, 3, 1, 2), g, name, key)
    for k, p in mod.named_parameters():
        if p.grad is None:
            assert k.startswith("text_linear.") or k.startswith("clip_linear."), k
            continue
        errs["dparam/" + k] = golden_grad_err(p.grad.float().cpu(), g, name, "dparam/" + k)
    if dtype == torch.bfloat16 and tol is None:      # per-tensor bars from the reference's own bf16 deviation
        bars, med = block_grad_bars(name)
        bad = {k: (v, bars[k]) for k, v in errs.items() if not v < bars[k]}
        assert not bad, (name, "bf16 per-tensor bars (error, bar)", bad)
        return max(errs.values()), max(v / bars[k] for k, v in errs.items()), med
    bad = {k: v for k, v in errs.items() if not v < tol}
    assert not bad, (name, str(dtype), bad)
    return max(errs.values())


def check_fused_block_equals_unfused(dev, name="nat_enc1", dtype=torch.bfloat16):
    """A PGSSTB block with the branch sum formed inside the gated-MLP launch (autograd_ops._Pgsstb / the no-grad twin) against the
    same block through _PgsstbAttn + _GatedMlp: output, input gradient and every parameter gradient bit for bit (the fused launch
    computes the same y; the backward is the same sequence of launches)."""
    from golden.cases import BLOCK_CASES, cotangent
    from golden.detfill import det_value
    from mp_hsir_amd import ops
    from mp_hsir_amd.net.MP_HSIR import PGSSTB
    c = BLOCK_CASES[name]
    mod = PGSSTB(c["C"], c["heads"], [64, 64], 8, c["shift"], 0.0, 2.66, c["cr"], 128).eval()
    with torch.no_grad():
        for k, p in mod.named_parameters():
            p.copy_(det_value(k, p.shape).float())
    mod = mod.to(dev)
    x0 = seeded_input(name, c["shape"], "normal").to(dev).permute(0, 2, 3, 1).contiguous().to(dtype)
    cot = cotangent(name, c["shape"]).to(dev).permute(0, 2, 3, 1).to(dtype)
    saved = (ops.MLP_FUSE_SUM, ops.MLP_FUSE_MIN_TILES)
    res = {}
    try:
        for fuse in (False, True):
            ops.MLP_FUSE_SUM, ops.MLP_FUSE_MIN_TILES = fuse, 1
            assert ops.gated_mlp_fuses(x0.numel() // c["C"], c["C"], c["shape"][2] * c["shape"][3], dtype) == fuse
            mod.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            ops.ACCOUNT = {}
            y = mod(x)
            (y * cot).sum().backward()
            acct, ops.ACCOUNT = ops.ACCOUNT, None
            with torch.no_grad():
                yi = mod(x0)
            res[fuse] = (y.detach(), yi, x.grad, {k: p.grad.clone() for k, p in mod.named_parameters()}, acct)
    finally:
        ops.MLP_FUSE_SUM, ops.MLP_FUSE_MIN_TILES = saved
    (y0, yi0, dx0, g0, a0), (y1, yi1, dx1, g1, a1) = res[False], res[True]
    assert a1["gemm_tok"][0] == a0["gemm_tok"][0] - 1, (a0["gemm_tok"], a1["gemm_tok"])       # the launch that disappeared
    assert torch.equal(y0, y1) and torch.equal(yi0, yi1) and torch.equal(y1, yi1) and torch.equal(dx0, dx1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k


def check_pack_plan(dev, dtype=torch.float32, steps=3):
    """engine.PackPlan (all kernel-layout weights = one gather from the flat arena) against the per-module packers:
    every module cache gets pinned, and the parameter trajectory of a few AdamW steps is bitwise the same."""
    from torch import nn
    from mp_hsir_amd import autograd_ops as AG
    from mp_hsir_amd import ops
    from mp_hsir_amd.engine import DataParallelEngine
    from mp_hsir_amd.net.MP_HSIR import PGSSTB, PromptFusion

    class Small(nn.Module):
        def __init__(self):
            super().__init__()
            self.inp = nn.Conv2d(8, 32, 3, padding=1, bias=False)
            self.blk = PGSSTB(32, 1, [8, 8], 8, 4, 0.0, 2.66, 8, 128)
            self.fuse = PromptFusion(64, 32, 2, 2.66, False)          # TransformerBlock(64) + 1x1 conv 64 -> 32
            self.out = nn.Conv2d(32, 8, 3, padding=1, bias=False)

        def forward(self, x, prompt):
            h = AG.conv3x3(x.permute(0, 2, 3, 1).contiguous().to(dtype), self.inp)
            h = self.fuse(h, self.blk(h))
            return AG.conv3x3(h, self.out).permute(0, 3, 1, 2).float() + x

    res = []
    for use_plan in (False, True):
        torch.manual_seed(3)
        net = Small().to(dev).eval()
        eng = DataParallelEngine(net, lr=1e-2, use_pack_plan=use_plan)
        g = torch.Generator().manual_seed(11)
        losses = []
        for _ in range(steps):
            x, c = torch.rand(2, 8, 16, 8, generator=g).to(dev), torch.rand(2, 8, 16, 8, generator=g).to(dev)
            losses.append(float(eng.train_step(x, c, None)))
        if use_plan:
            assert eng.plan is not None and eng.plan.pinned >= 5 and eng.plan.skipped == 0, (eng.plan.pinned, eng.plan.skipped)
            ops.ACCOUNT = {}
            eng.train_step(x, c, None)
            acct, ops.ACCOUNT = ops.ACCOUNT, None
            assert acct["pack_gather"][0] == len(eng.plan.groups) >= 1
            eng.plan.release()
        else:
            eng.train_step(x, c, None)
        res.append((losses, {k: v.detach().clone().cpu() for k, v in net.state_dict().items()}))
    assert res[0][0] == res[1][0], (res[0][0], res[1][0])
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


def check_tiny_adamw(dev, use_graph=False, steps=2):
    """a18 on the HIP path: two optimisation steps of the tiny net through engine.DataParallelEngine (HIP forward /
    backward, mphsir_flat_adamw on the flat arenas) against tests/golden/tiny_adamw.npz, i.e. the REFERENCE model
    stepped by torch.optim.AdamW(lr=2e-4) (train.py:69): losses, and per parameter the norm and seeded samples of
    (p_after - p_before)."""
    from golden.cases import sample_indices
    from mp_hsir_amd.engine import DataParallelEngine
    net = build_net(TINY_CFG, dev, torch.float32)
    p0 = {k: p.detach().clone() for k, p in net.named_parameters()}
    eng = DataParallelEngine(net, lr=2e-4, use_graph=use_graph, graph_warmup=0 if use_graph else 2)
    task = torch.tensor([[1], [3]]).to(dev)
    losses = []
    for step in range(steps):
        xs = seeded_input("adam_x%d" % step, (2, 8, 64, 64)).to(dev)
        cs = seeded_input("adam_c%d" % step, (2, 8, 64, 64)).to(dev)
        losses.append(float(eng.train_step(xs, cs, task)))
    if use_graph:
        eng.finish()
    g = np.load(os.path.join(GOLDEN, "tiny_adamw.npz"))
    np.testing.assert_allclose(losses, g["losses"][:steps], rtol=1e-5)
    # Adam turns every gradient element into an update of about lr * sign(g): elements whose gradient is comparable to
    # eps = 1e-8 are ill-conditioned (fp32-vs-fp64 noise in g moves them by a sizeable fraction of lr), so the per-tensor
    # NORM of the update is held to 2e-2 and the sampled elements to 5 % of the largest sampled update (measured worst
    # on the MI355X: 1.4e-2 / 2.7e-2, both on the spectral-prompt gate's prompt_param / linear_prompt.weight, whose
    # gradients sit at 1e-8..1e-7; every other tensor is below 2e-3); the loss after the first update (losses[1], above,
    # 1e-5) pins the step as a whole.
    bad, worst = [], 0.0
    for k, p in net.named_parameters():
        d = (p.detach() - p0[k]).double().cpu()
        want = float(g["delta_norm/" + k])
        if want == 0.0:                                   # the 8 parameters without gradient: untouched
            assert float(d.abs().max()) == 0.0, k
            continue
        e_norm = abs(float(d.norm()) - want) / want
        idx = sample_indices(k, d.numel())
        ws = torch.as_tensor(g["delta_samp/" + k], dtype=torch.float64)
        e_samp = float((d.flatten()[idx] - ws).abs().max() / ws.abs().max().clamp_min(1e-30))
        worst = max(worst, e_norm, e_samp)
        if not (e_norm < 2e-2 and e_samp < 5e-2):
            bad.append((k, e_norm, e_samp))
    assert not bad, sorted(bad, key=lambda t: -max(t[1], t[2]))[:8]
    return worst
