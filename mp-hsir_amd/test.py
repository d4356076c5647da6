#!/usr/bin/env python3
"""Evaluation script: the 13 degradation modes of the reference's test.py (test.py:80-530, dispatch :580-645), batch-1
forward under no_grad, band-wise PSNR / SSIM on the device (metrics.py = utils/val_utils.py:49-105 without skimage).

  mode  degradation (utils/dataset_utils.py)                              prompt id   flag
   0    Gaussian noise sigma (:277-305)                                    0          --gaussian_noise_sigma 70
   1    non-iid Gaussian noise, per-band sigma from a list (:307-340)      1          --gaussian_noise_sigmas
   2    non-iid Gaussian + stripes (:342-406)                              1          --stripe_nosie_ratio
   3    non-iid Gaussian + deadlines (:408-466)                            1          --deadline_nosie_ratio
   4    non-iid Gaussian + impulse (:468-522)                              1          --impulse_nosie_ratio
   5    Gaussian blur, kernel size (:571-622)                              2          --gaussian_blur_radius 15
   6    motion blur (kernel size, angle) (:624-679)                        0          --motion_blur_radius
   7    bicubic down x nearest up (:681-726)                               3          --downsample_factor 8
   8    random mask (:728-769)                                             4          --mask_ratio 0.9
   9    haze (:771-840)                                                    5          --haze_omega 1
  10    missing bands, scored on the missing bands only (:842-879)         5 (6 RS)   --bandmis_ratio 0.3
  11    Poisson noise scale 10 (:243-275)                                  0
  12    real degraded / clean pairs from --test_degrad_dir (:197-241)      1

Test cubes: --test_dir with .mat ('data' key, as the reference) or .npy cubes, centre-cropped to multiples of 64
(crop_img, utils/image_utils.py:58-70); without it, synthetic cubes (no datasets offline).  The degradations are the
GPU functions of degrade.py.  --ckpt_path evaluates a Lightning checkpoint of the reference (`net.` key prefix).
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mp_hsir_amd import degrade as D  # noqa: E402
from mp_hsir_amd.engine import GraphedForward  # noqa: E402
from mp_hsir_amd.metrics import compute_psnr_ssim, compute_psnr_ssim2  # noqa: E402
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net  # noqa: E402


def psnr_bandwise(restored, clean):
    return compute_psnr_ssim(restored, clean)[0]


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--cuda", type=int, default=0)                       # reference default 4 (test.py:543): box specific
    p.add_argument("--seed", type=int, default=2024)
    p.add_argument("--mode", type=int, default=0, help="Used to select degradation mode.")
    p.add_argument("--test_dir", type=str, default="", help="where clean HSIs of test saves.")
    p.add_argument("--test_degrad_dir", type=str, default="", help="where real degraded HSIs of test saves.")
    p.add_argument("--degrad_id", type=int, default=1)
    p.add_argument("--gaussian_noise_sigma", type=int, default=70, help="Gaussian Noise intensity")
    p.add_argument("--gaussian_noise_sigmas", type=int, nargs="+", default=[10, 30, 50, 70], help="Gaussian Noise inid intensity")
    p.add_argument("--stripe_nosie_ratio", type=float, nargs=2, default=[0.05, 0.15], help="Stripe ratio")
    p.add_argument("--deadline_nosie_ratio", type=float, nargs=2, default=[0.05, 0.15], help="Deadline ratio")
    p.add_argument("--impulse_nosie_ratio", type=float, nargs="+", default=[0.1, 0.3, 0.5, 0.7], help="Impulse ratio")
    p.add_argument("--gaussian_blur_radius", type=int, default=15, help="Gaussian Blur")
    p.add_argument("--motion_blur_radius", type=int, nargs=2, default=(15, 45), help="Motion Blur")
    p.add_argument("--downsample_factor", type=int, default=8, help="factor")
    p.add_argument("--mask_ratio", type=float, default=0.9, help="Inpaint Mask Ratio")
    p.add_argument("--haze_omega", type=float, default=1, help="haze")
    p.add_argument("--bandmis_ratio", type=float, default=0.3, help="Bandmis Ratio")
    p.add_argument("--select_bands", type=list, default=[27, 15, 9])
    p.add_argument("--output_path", type=str, default="output/")
    p.add_argument("--ckpt_path", type=str, default=None)
    p.add_argument("--rank", type=int, default=31)
    # additions
    p.add_argument("--model", type=str, default="natural_scene", choices=["natural_scene", "remote_sensing"])
    p.add_argument("--size", type=int, default=512, help="synthetic cube height/width (reference test cubes: 512)")
    p.add_argument("--cubes", type=int, default=4)
    p.add_argument("--precision", type=str, default="f32", choices=["bf16", "f32"])
    p.add_argument("--allow_surrogate_clip", type=int, default=0)
    return p


def crop_img(x, base=64):
    """(C,H,W) centre crop to multiples of `base` (utils/image_utils.py:58-70)"""
    h, w = x.shape[-2:]
    ch, cw = h % base, w % base
    return x[..., ch // 2:h - ch + ch // 2, cw // 2:w - cw + cw // 2]


def load_cube(path):
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32)
    import scipy.io as sio
    return np.array(sio.loadmat(path)["data"]).astype(np.float32)


def cube_source(o, bands, dev, gen):
    """yields (name, clean (1,C,H,W) on dev[, real degraded])"""
    if o.test_dir:
        for fn in sorted(os.listdir(o.test_dir)):
            clean = torch.from_numpy(crop_img(load_cube(os.path.join(o.test_dir, fn)))).to(dev)[None]
            real = None
            if o.mode == 12:
                real = torch.from_numpy(crop_img(load_cube(os.path.join(o.test_degrad_dir, fn)))).to(dev)[None]
            yield fn.split(".")[0], clean, real
    else:
        if o.mode == 12:
            raise SystemExit("mode 12 evaluates real degraded/clean pairs: pass --test_dir and --test_degrad_dir")
        for i in range(o.cubes):
            yield "synthetic_%d" % i, torch.rand((1, bands, o.size, o.size), generator=gen, device=dev), None


def degrade_for_mode(o, clean, d, model):
    """-> (degraded, prompt id) for one clean cube (1,C,H,W); d = degrade.Draws"""
    B, C, H, W = clean.shape
    nb = int(np.floor(C / 3))

    def non_iid(x, sigmas=(10, 30, 50, 70)):
        return D.gaussian_noise_non_iid(x, d.choice([s / 255.0 for s in sigmas], B * C).reshape(B, C), d.randn(B, C, H, W))
    m = o.mode
    if m == 0:
        return D.gaussian_noise(clean, torch.full((B,), o.gaussian_noise_sigma / 255.0, device=clean.device), d.randn(B, C, H, W)), 0
    if m == 1:
        return non_iid(clean, o.gaussian_noise_sigmas), 1
    if m == 2:
        lo, hi = o.stripe_nosie_ratio
        bands = d.band_subset(B, C, nb)
        n = d.randint(int(lo * W), max(int(hi * W), int(lo * W) + 1), (B, C))
        cols = d.column_subsets(B, C, W, n)
        return D.stripe_noise(non_iid(clean), bands, (d.rand(B, C, W) * 0.5 - 0.25) * cols), 1
    if m == 3:
        lo, hi = o.deadline_nosie_ratio
        bands = d.band_subset(B, C, nb)
        n = d.randint(int(np.ceil(lo * W)), max(int(np.ceil(hi * W)), int(np.ceil(lo * W)) + 1), (B, C))
        return D.deadline_noise(non_iid(clean), d.column_subsets(B, C, W, n) & bands[:, :, None]), 1
    if m == 4:
        bands = d.band_subset(B, C, nb)
        p = d.choice(list(o.impulse_nosie_ratio), B).reshape(B, 1, 1, 1)
        return D.impulse_noise(non_iid(clean), (d.rand(B, C, H, W) < p) & bands[:, :, None, None], d.rand(B, C, H, W) < 0.5), 1
    if m == 5:
        return D.blur(clean, D.gaussian_kernel2d(o.gaussian_blur_radius)), 2
    if m == 6:
        return D.blur(clean, D.motion_kernel2d(*o.motion_blur_radius)), 0
    if m == 7:
        return D.super_resolution_input(clean, o.downsample_factor), 3
    if m == 8:
        return D.random_mask(clean, d.rand(B, C, H, W), torch.full((B,), o.mask_ratio, device=clean.device)), 4
    if m == 9:
        low = d.rand(B, 1, max(H // 16, 2), max(W // 16, 2))
        cirrus = torch.nn.functional.interpolate(low, size=(H, W), mode="bilinear", align_corners=True)[:, 0]
        return D.haze(clean, cirrus, torch.full((B,), float(o.haze_omega), device=clean.device)), 5
    if m == 10:
        n = int(o.bandmis_ratio * C)
        return D.band_loss(clean, d.rand(B, C).argsort(dim=1).argsort(dim=1) < n), (5 if model == "natural_scene" else 6)
    if m == 11:
        return D.poisson_noise(clean, 10.0, generator=d.gen), 0
    raise SystemExit("unknown mode %d" % m)


MODE_LABEL = {0: "Denoise sigma=%(gaussian_noise_sigma)s", 1: "Denoise sigma=%(gaussian_noise_sigmas)s",
              2: "Destripe stripe ratio=%(stripe_nosie_ratio)s", 3: "Deadline denoise deadline ratio=%(deadline_nosie_ratio)s",
              4: "Impulse denoise impulse ratio=%(impulse_nosie_ratio)s", 5: "Gaussian deblur sigma=%(gaussian_blur_radius)s",
              6: "Motion deblur motion radius=%(motion_blur_radius)s", 7: "Super resolution downsample factor=%(downsample_factor)s",
              8: "Inpaint mask ratio=%(mask_ratio)s", 9: "Dehaze haze omega=%(haze_omega)s", 10: "Bandmiss ratio=%(bandmis_ratio)s",
              11: "Degrad_Id=%(degrad_id)s", 12: "Degrad_Id=%(degrad_id)s"}


def evaluate(o, net, dev):
    """-> (mean psnr, mean ssim, number of cubes); prints one line per cube"""
    cfg_bands = net.patch_embed.proj.weight.shape[1]
    gen = torch.Generator(device=dev).manual_seed(o.seed)
    d = D.Draws(dev, o.seed + 1)
    run = GraphedForward(net)            # cubes of one shape: captured after two eager calls, then replayed
    ps = ss = 0.0
    n = 0
    for name, clean, real in cube_source(o, cfg_bands, dev, gen):
        if o.mode == 12:
            degraded, pid = real, 1
        else:
            degraded, pid = degrade_for_mode(o, clean, d, o.model)
        restored = run(degraded.float().contiguous(), torch.tensor([pid], device=dev))
        clean_c = clean.clamp(0, 1)
        if o.mode == 10:
            p, s, cnt = compute_psnr_ssim2(restored, clean_c, degraded)      # only the completed bands (test.py:523)
        else:
            p, s, cnt = compute_psnr_ssim(restored, clean_c)
        ps, ss, n = ps + p * cnt, ss + s * cnt, n + cnt
        print("%s psnr %.2f ssim %.4f" % (name, p, s))
    return ps / max(n, 1), ss / max(n, 1), n


def main():
    o = build_parser().parse_args()
    torch.manual_seed(o.seed)
    np.random.seed(o.seed)
    dev = torch.device("cuda", o.cuda)
    cfg = dict(in_channel=31, out_channel=31, dim=64, task_classes=6) if o.model == "natural_scene" else \
        dict(in_channel=100, out_channel=100, dim=96, task_classes=7)
    clip_prompt = "surrogate" if o.allow_surrogate_clip else None
    ckpt = torch.load(o.ckpt_path, map_location="cpu") if o.ckpt_path else None
    if ckpt is not None and ckpt.get("mphsir_clip_prompt") is not None:
        clip_prompt = ckpt["mphsir_clip_prompt"]          # the table the checkpoint was trained with
    net = MP_HSIR_Net(**cfg, clip_prompt=clip_prompt, compute_dtype=torch.float32 if o.precision == "f32" else torch.bfloat16).to(dev).eval()
    if ckpt is not None:
        net.load_state_dict({k[4:]: v for k, v in ckpt["state_dict"].items() if k.startswith("net.")}, strict=False)   # test.py:575
        print("CKPT name : {}".format(o.ckpt_path))
    p, s, n = evaluate(o, net, dev)
    print((MODE_LABEL[o.mode] % vars(o)) + ": psnr: %.2f, ssim: %.4f" % (p, s))


if __name__ == "__main__":
    main()
