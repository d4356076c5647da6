#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE model (/root/reference/net/MP_HSIR.py) on CPU.

Runs only in the build container (the reference tree does not exist on the GPU box).  The
reference is imported through two tiny stand-ins for packages the image lacks (tests/golden/
refshim: timm.models.layers and clip); nothing from /root/reference is copied.  Every weight
comes from tests/golden/detfill.py (crc32(key)-seeded), every input from `seeded_input`, so the
fixtures hold only *outputs* (plus a few checksums) and stay small.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

The reference is run in float64 (module.double()) and results are stored rounded to float32:
that is the reference's arithmetic with its own fp32 rounding noise (rel-L2 6e-7)
removed, which lets the oracle be pinned at 1e-6 instead of 1e-5.  For the two full-width nets the
plain fp32 run is sampled as well (`*_f32run_first4096`).
"""
import os
import sys
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(0, os.path.dirname(HERE))  # tests/ -> `golden.detfill`
sys.path.insert(1, "/root/reference")
warnings.filterwarnings("ignore")

from golden.detfill import det_fill_, seeded_input, surrogate_clip_prompt  # noqa: E402
from golden.cases import (TINY_CFG, TINY_CASES, BLOCK_CASES, FULL_CASES, GRAD_KEYS_FULL,  # noqa: E402
                          GRAD_FULL_MAX, GRAD_SAMPLES, sample_indices, cotangent)
import net.MP_HSIR as ref  # noqa: E402  (the reference)

torch.set_num_threads(8)
F32 = np.float32


def to_np(t):
    return t.detach().to(torch.float32).cpu().numpy()


def build_ref_net(cfg, dtype=torch.float64):
    net = ref.MP_HSIR_Net(**cfg).eval()
    det_fill_(net)
    net = net.to(dtype)
    net.text_prompt.clip_prompt = net.text_prompt.clip_prompt.to(dtype)
    return net


def task_tensor(task):
    return torch.tensor(task, dtype=torch.long)


def gen_tiny():
    out = {}
    nets = {}
    for name, case in TINY_CASES.items():
        cfg = dict(TINY_CFG, task_classes=case.get("task_classes", TINY_CFG["task_classes"]))
        key = cfg["task_classes"]
        if key not in nets:
            nets[key] = build_ref_net(cfg)
        net = nets[key]
        x = seeded_input(name, case["shape"]).double()
        with torch.no_grad():
            y = net(x, task_tensor(case["task"]))
        if "keep" in case:
            out[name + "/out"] = to_np(y[case["keep"]])
            out[name + "/norms"] = y.flatten(1).norm(dim=1).numpy()
        else:
            out[name + "/out"] = to_np(y)
        print("tiny", name, tuple(y.shape), float(y.abs().mean()))
    np.savez_compressed(os.path.join(HERE, "tiny_fwd.npz"), **out)


def gen_tiny_grad():
    """L1-after-clamp loss (train.py:58-61), eval mode (DropPath = identity), all param grads."""
    net = build_ref_net(TINY_CFG)
    for p in net.parameters():
        p.requires_grad_(True)
    x = seeded_input("grad_x", (2, 8, 64, 64)).double()
    clean = seeded_input("grad_clean", (2, 8, 64, 64)).double()
    task = task_tensor([[1], [3]])
    y = net(x, task)
    loss = F.l1_loss(torch.clamp(y, 0, 1), clean)
    loss.backward()
    out = {"loss": np.array(float(loss)), "out": to_np(y)}
    none_keys = []
    for k, p in net.named_parameters():
        if p.grad is None:
            none_keys.append(k)
            continue
        g = p.grad
        out["norm/" + k] = np.array(float(g.norm()))
        out["sum/" + k] = np.array(float(g.sum()))
        idx = sample_indices(k, g.numel())
        out["samp/" + k] = g.flatten()[idx].numpy()
        if any(k.startswith(pref) for pref in GRAD_KEYS_FULL):
            out["full/" + k] = to_np(g)
    out["none_keys"] = np.array(none_keys)
    print("tiny grad: loss", float(loss), "none:", none_keys)
    np.savez_compressed(os.path.join(HERE, "tiny_grad.npz"), **out)

    # two AdamW steps with the reference's optimizer settings (train.py:69: AdamW(lr) defaults)
    net = build_ref_net(TINY_CFG)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4)
    out = {}
    losses = []
    for step in range(2):
        xs = seeded_input("adam_x%d" % step, (2, 8, 64, 64)).double()
        cs = seeded_input("adam_c%d" % step, (2, 8, 64, 64)).double()
        opt.zero_grad(set_to_none=True)
        loss = F.l1_loss(torch.clamp(net(xs, task), 0, 1), cs)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    out["losses"] = np.array(losses)
    ref0 = build_ref_net(TINY_CFG)
    p0 = dict(ref0.named_parameters())
    for k, p in net.named_parameters():
        idx = sample_indices(k, p.numel())
        out["delta_samp/" + k] = (p.detach() - p0[k].detach()).flatten()[idx].numpy()
        out["delta_norm/" + k] = np.array(float((p.detach() - p0[k].detach()).norm()))
    print("tiny adamw losses", losses)
    np.savez_compressed(os.path.join(HERE, "tiny_adamw.npz"), **out)


def put_grad(out, name, key, g):
    """gradient tensor -> fixture entries: full when small (or the case asks for it), else norm + seeded samples."""
    if g.numel() <= GRAD_FULL_MAX or BLOCK_CASES[name].get("grad_full", False):
        out["%s/%s" % (name, key)] = to_np(g)
    else:
        idx = sample_indices(name + ":" + key, g.numel(), GRAD_SAMPLES)
        out["%s/samp/%s" % (name, key)] = to_np(g.flatten()[idx])
        out["%s/norm/%s" % (name, key)] = np.array(float(g.norm()))


def gen_blocks():
    out = {}
    for name, c in BLOCK_CASES.items():
        kind = c["kind"]
        if kind == "pgsstb":
            m = ref.PGSSTB(dim=c["C"], num_heads=c["heads"], input_resolution=[64, 64], window_size=8,
                           shift_size=c["shift"], mlp_ratio=2.66, compress_ratio=c["cr"], prompt_len=128,
                           drop_path=0.0, qkv_bias=True, bias=False).eval()
            det_fill_(m)
            m = m.double()
            x = seeded_input(name, c["shape"], "normal").double().requires_grad_(c.get("grad", False))
            cap = {}
            if c.get("intermediates", False):
                m.attn.register_forward_hook(lambda mod, i, o: cap.__setitem__("sa", o))
                m.local_spectral_attn.register_forward_hook(lambda mod, i, o: cap.__setitem__("x1", o))
                m.gobal_spectral_attn.register_forward_hook(lambda mod, i, o: cap.__setitem__("x2", o))
                m.mlp.register_forward_hook(lambda mod, i, o: cap.__setitem__("mlp", o))
            if c.get("grad", False):
                for p in m.parameters():
                    p.requires_grad_(True)
                y = m(x)
                (y * cotangent(name, y.shape).double()).sum().backward()
                put_grad(out, name, "dx", x.grad)
                for k, p in m.named_parameters():
                    put_grad(out, name, "dparam/" + k, p.grad)
            else:
                with torch.no_grad():
                    y = m(x)
            out[name + "/out"] = to_np(y)
            for k, v in cap.items():
                out[name + "/" + k] = to_np(v)
        elif kind == "tvsp":
            m = ref.TVSP(task_classes=c["T"], prompt_size=c["ps"], prompt_dim=c["D"], out_dim=c["D"]).eval()
            det_fill_(m)
            m = m.double()
            B = c["shape"][0]
            x = seeded_input(name, c["shape"], "normal").double()
            w = F.one_hot(task_tensor(c["task"]), c["T"])
            clip = (w.unsqueeze(-1) * surrogate_clip_prompt(c["T"]).double().unsqueeze(0)).mean(1)
            assert clip.shape == (B, 512)
            for p in m.parameters():
                p.requires_grad_(True)
            y = m(x, clip, w)
            (y * cotangent(name, y.shape).double()).sum().backward()
            for k, p in m.named_parameters():
                if p.grad is not None:          # text_linear / clip_linear never receive one
                    put_grad(out, name, "dparam/" + k, p.grad)
            out[name + "/out"] = to_np(y)
        elif kind == "fusion":
            m = ref.PromptFusion(dim=c["D"] * 2, out_dim=c["D"], head=c["heads"]).eval()
            det_fill_(m)
            m = m.double()
            x = seeded_input(name + ":x", c["shape"], "normal").double().requires_grad_(True)
            p = seeded_input(name + ":p", c["shape"], "normal").double().requires_grad_(True)
            for q in m.parameters():
                q.requires_grad_(True)
            y = m(x, p)
            (y * cotangent(name, y.shape).double()).sum().backward()
            put_grad(out, name, "dx", x.grad)
            put_grad(out, name, "dprompt", p.grad)
            for k, q in m.named_parameters():
                put_grad(out, name, "dparam/" + k, q.grad)
            out[name + "/out"] = to_np(y)
        print("block", name, tuple(y.shape), float(y.abs().mean()))
    np.savez_compressed(os.path.join(HERE, "blocks.npz"), **out)


def _block_module(name, c):
    """the reference module of a BLOCK_CASES entry with detfill weights, fp32, plus a function that runs it on fresh leaf inputs"""
    kind = c["kind"]
    if kind == "pgsstb":
        m = ref.PGSSTB(dim=c["C"], num_heads=c["heads"], input_resolution=[64, 64], window_size=8, shift_size=c["shift"], mlp_ratio=2.66,
                       compress_ratio=c["cr"], prompt_len=128, drop_path=0.0, qkv_bias=True, bias=False).eval()
        det_fill_(m)

        def run(mod, dt):
            x = seeded_input(name, c["shape"], "normal").to(dt).requires_grad_(True)
            return mod(x), {"dx": x}
    elif kind == "tvsp":
        m = ref.TVSP(task_classes=c["T"], prompt_size=c["ps"], prompt_dim=c["D"], out_dim=c["D"]).eval()
        det_fill_(m)

        def run(mod, dt):
            x = seeded_input(name, c["shape"], "normal").to(dt)
            w = F.one_hot(task_tensor(c["task"]), c["T"])
            clip = (w.unsqueeze(-1) * surrogate_clip_prompt(c["T"]).to(dt).unsqueeze(0)).mean(1)
            return mod(x, clip, w), {}
    else:
        m = ref.PromptFusion(dim=c["D"] * 2, out_dim=c["D"], head=c["heads"]).eval()
        det_fill_(m)

        def run(mod, dt):
            x = seeded_input(name + ":x", c["shape"], "normal").to(dt).requires_grad_(True)
            p = seeded_input(name + ":p", c["shape"], "normal").to(dt).requires_grad_(True)
            return mod(x, p), {"dx": x, "dprompt": p}
    return m, run


def gen_block_autocast():
    """blocks_autocast.npz: per gradient tensor of every BLOCK_CASES entry with gradients, the deviation of the REFERENCE's OWN
    mixed-precision backward (torch.autocast bf16 / fp16: train.py:118 trains in 16-mixed) from its fp64 gradients -- the yardstick
    the 16-bit HIP block gradients are held to, tensor by tensor (tests/model_checks.py::block_grad_bars), instead of one flat bar."""
    import copy
    out = {}
    for name, c in BLOCK_CASES.items():
        if c["kind"] == "pgsstb" and not c.get("grad", False):
            continue
        m32, run = _block_module(name, c)
        m64 = copy.deepcopy(m32).double()
        for mod in (m32, m64):
            for p in mod.parameters():
                p.requires_grad_(True)
        y64, ins64 = run(m64, torch.float64)
        cot = cotangent(name, y64.shape)
        (y64 * cot.double()).sum().backward()
        g64 = {"dparam/" + k: p.grad for k, p in m64.named_parameters() if p.grad is not None}
        g64.update({k: t.grad for k, t in ins64.items()})
        for tag, low, scale in (("bf16", torch.bfloat16, 1.0),):      # (fp16 autocast of a stand-alone block overflows in fc2.bias without a tuned loss scale: the fp16 block bar stays flat)
            for p in m32.parameters():
                p.grad = None
            with torch.autocast(device_type="cpu", dtype=low):
                ya, insa = run(m32, torch.float32)
                loss = (ya.float() * cot).sum() * scale
            loss.backward()
            ga = {"dparam/" + k: p.grad for k, p in m32.named_parameters() if p.grad is not None}
            ga.update({k: t.grad for k, t in insa.items()})
            errs = []
            for k, g in g64.items():
                e = float((ga[k].double() / scale - g).norm() / (g.norm() + 1e-300))
                out["%s/%s/%s" % (name, tag, k)] = np.array(e)
                errs.append(e)
            out["%s/%s/out" % (name, tag)] = np.array(float((ya.detach().double() - y64.detach()).norm() / y64.detach().norm()))
            errs.sort()
            print("block autocast", name, tag, "out", float(out["%s/%s/out" % (name, tag)]), "median", errs[len(errs) // 2], "max", errs[-1], flush=True)
    np.savez_compressed(os.path.join(HERE, "blocks_autocast.npz"), **out)


def psnr_ref(restored, clean):
    """Band-wise PSNR as utils/val_utils.py:49-69 defines it (skimage absent: data_range=1 form)."""
    r = np.clip(restored, 0, 1).astype(np.float64)
    c = np.clip(clean, 0, 1).astype(np.float64)
    mse = ((r - c) ** 2).mean(axis=(-1, -2))  # (B,C)
    return float((10 * np.log10(1.0 / mse)).mean(axis=1).mean())


def gen_full():
    out = {}
    for name, c in FULL_CASES.items():
        clean = seeded_input(name + ":clean", c["shape"])
        if c["recipe"] == "gaussian70":  # dataset_utils.py:293-298, test.py:554
            degraded = clean + seeded_input(name + ":noise", c["shape"], "normal") * (70.0 / 255.0)
        else:  # inpaint, dataset_utils.py:743-749: keep where rand > ratio (0.9)
            mask = (seeded_input(name + ":mask", c["shape"]) > 0.9).float()
            degraded = clean * mask
        task = task_tensor(c["task"])
        net = build_ref_net(c["cfg"])
        with torch.no_grad():
            y64 = net(degraded.double(), task)
        net32 = build_ref_net(c["cfg"], torch.float32)
        with torch.no_grad():
            y32 = net32(degraded, task)
        out[name + "/out"] = to_np(y64)
        out[name + "/out_f32run_first4096"] = to_np(y32).reshape(-1)[:4096]
        out[name + "/psnr_restored"] = np.array(psnr_ref(to_np(y64), clean.numpy()))
        out[name + "/psnr_degraded"] = np.array(psnr_ref(degraded.numpy(), clean.numpy()))
        rel = float((y64.float() - y32).norm() / y64.norm())
        print("full", name, tuple(y64.shape), "psnr", out[name + "/psnr_restored"], "f32-vs-f64 rel", rel)
    np.savez_compressed(os.path.join(HERE, "full.npz"), **out)


def gen_cubes(which=None):
    """the reference itself at FULL size, once per CUBE_CASES entry, in fp32 (fp64 would need > 60 GB for the
    window-attention intermediates of a 512x512 cube); ~150-250 s each on 8 threads.  Stored: summary statistics only."""
    import time
    from golden.cases import CUBE_CASES, CUBE_SAMPLES, cube_inputs
    path = os.path.join(HERE, "cubes.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in (which or CUBE_CASES):
        c, clean, degraded = cube_inputs(name)
        net = build_ref_net(c["cfg"], torch.float32)
        t0 = time.time()
        with torch.no_grad():
            y = net(degraded, task_tensor(c["task"]))
        idx = sample_indices("cube:" + name, y.numel(), CUBE_SAMPLES)
        out[name + "/norm"] = np.array(float(y.double().norm()))
        out[name + "/mean"] = np.array(float(y.double().mean()))
        out[name + "/samples"] = to_np(y.flatten()[idx])
        out[name + "/band_means"] = y.double().mean(dim=(0, 2, 3)).numpy()
        out[name + "/band_norms"] = y.double().flatten(2).norm(dim=2)[0].numpy()
        out[name + "/psnr_restored"] = np.array(psnr_ref(to_np(y), clean.numpy()))
        out[name + "/psnr_degraded"] = np.array(psnr_ref(degraded.numpy(), clean.numpy()))
        print("cube", name, tuple(y.shape), "norm", out[name + "/norm"], "psnr", out[name + "/psnr_restored"], "%.0f s" % (time.time() - t0), flush=True)
        del net, y
    np.savez_compressed(path, **out)


def gen_full_grad(which=None):
    """the training step's gradients at real width: loss = L1(clamp(net(degraded), 0, 1), clean) (train.py:58-61) of the
    reference in fp64, batch 2, every parameter gradient as norm + seeded samples (golden.cases.FULLGRAD_CASES)"""
    import time
    from golden.cases import FULLGRAD_CASES, FULLGRAD_SAMPLES, fullgrad_inputs
    path = os.path.join(HERE, "full_grad.npz")
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in (which or FULLGRAD_CASES):
        c, clean, degraded = fullgrad_inputs(name)
        net = build_ref_net(c["cfg"])
        for p in net.parameters():
            p.requires_grad_(True)
        t0 = time.time()
        y = net(degraded.double(), task_tensor(c["task"]))
        loss = F.l1_loss(torch.clamp(y, 0, 1), clean.double())
        loss.backward()
        for k in [k for k in out if k.startswith(name + "/")]:
            del out[k]
        out[name + "/loss"] = np.array(float(loss))
        out[name + "/out_norm"] = np.array(float(y.detach().norm()))
        none_keys = []
        for k, p in net.named_parameters():
            if p.grad is None:
                none_keys.append(k)
                continue
            g = p.grad
            out["%s/norm/%s" % (name, k)] = np.array(float(g.norm()))
            if g.numel() <= FULLGRAD_SAMPLES:
                out["%s/full/%s" % (name, k)] = to_np(g)
            else:
                idx = sample_indices(name + ":" + k, g.numel(), FULLGRAD_SAMPLES)
                out["%s/samp/%s" % (name, k)] = to_np(g.flatten()[idx])
        out[name + "/none_keys"] = np.array(none_keys)
        print("full grad", name, "loss", float(loss), "none:", len(none_keys), "%.0f s" % (time.time() - t0), flush=True)
        g64 = {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}
        y64 = y.detach()
        del net, y, loss
        # the reference's OWN mixed-precision deviation, per tensor: the same step under torch.autocast (train.py:118 trains in
        # 16-mixed; bf16 for the natural case, fp16 with a fixed loss scale for the remote-sensing one) against its fp64 gradients.
        # The tests hold the 16-bit HIP backward to a small multiple of THIS, tensor by tensor.
        low, scale = c["autocast"]
        net = build_ref_net(c["cfg"], torch.float32)
        for p in net.parameters():
            p.requires_grad_(True)
        with torch.autocast(device_type="cpu", dtype=low):
            ya = net(degraded, task_tensor(c["task"]))
            la = F.l1_loss(torch.clamp(ya.float(), 0, 1), clean)
        (la * scale).backward()
        out[name + "/autocast_out_err"] = np.array(float((ya.detach().double() - y64).norm() / y64.norm()))
        out[name + "/autocast_res_err"] = np.array(float((ya.detach().double() - y64).norm() / (y64 - degraded.double()).norm()))
        for k, p in net.named_parameters():
            if k in g64:
                ga = p.grad.detach().double() / scale
                out["%s/autocast_err/%s" % (name, k)] = np.array(float((ga - g64[k]).norm() / g64[k].norm()))
        errs = sorted(float(out[k]) for k in out if k.startswith(name + "/autocast_err/"))
        print("   autocast", low, "out err", float(out[name + "/autocast_out_err"]), "residual err", float(out[name + "/autocast_res_err"]),
              "grad err median %.3g  90%% %.3g  max %.3g" % (errs[len(errs) // 2], errs[int(len(errs) * 0.9)], errs[-1]), flush=True)
        del net
    np.savez_compressed(path, **out)


def gen_schedule():
    """a18: the reference's own LinearWarmupCosineAnnealingLR (utils/schedulers.py:295-346) driven as train.py:67-86 drives it
    -- AdamW(lr), warmup_epochs = int(0.1 * epochs), max_epochs = epochs, eta_min = 1e-6, one scheduler.step() per epoch
    (the chainable get_lr form Lightning calls) -- for the reference's default run (100 epochs) and a 300-epoch run."""
    from utils.schedulers import LinearWarmupCosineAnnealingLR
    out = {}
    for epochs, lr in ((100, 2e-4), (300, 2e-4), (20, 1e-3)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.AdamW([p], lr=lr)
        sch = LinearWarmupCosineAnnealingLR(optimizer=opt, warmup_epochs=int(0.1 * epochs), max_epochs=epochs, eta_min=1e-6)
        lrs = []
        for _ in range(epochs + 1):          # lr in force during epoch e, e = 0 .. epochs
            lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        out["e%d_lr%g" % (epochs, lr)] = np.array(lrs, dtype=np.float64)
        print("schedule", epochs, lr, lrs[:3], lrs[-2:])
    np.savez_compressed(os.path.join(HERE, "schedule.npz"), **out)


def gen_keys():
    """state_dict key/shape/dtype manifests for the two shipped configurations."""
    import json
    man = {}
    for name, c in FULL_CASES.items():
        net = ref.MP_HSIR_Net(**c["cfg"])
        man[name] = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
        man[name + ":nparams"] = sum(p.numel() for p in net.parameters())
    net = ref.MP_HSIR_Net(**TINY_CFG)
    man["tiny"] = {k: [list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()}
    with open(os.path.join(HERE, "state_dict_manifest.json"), "w") as f:
        json.dump(man, f, indent=0, sort_keys=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["keys", "tiny", "grad", "blocks", "full", "schedule", "cubes"]
    if "keys" in which:
        gen_keys()
    if "tiny" in which:
        gen_tiny()
    if "grad" in which:
        gen_tiny_grad()
    if "blocks" in which:
        gen_blocks()
    if "block_autocast" in which:
        gen_block_autocast()
    if "full" in which:
        gen_full()
    if "schedule" in which:
        gen_schedule()
    if "fullgrad" in which:
        gen_full_grad()
    if "cubes" in which or any(w.startswith("cube:") for w in which):
        gen_cubes([w[5:] for w in which if w.startswith("cube:")] or None)
