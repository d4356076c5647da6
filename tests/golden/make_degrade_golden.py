#!/usr/bin/env python3
"""Generate tests/golden/degrade.npz by running the REFERENCE degradation functions
(/root/reference/utils/degradation_utils.py, utils/image_utils.py) on seeded inputs.

The reference draws from numpy's global state inside each function.  To obtain (draws, output) pairs the script seeds
numpy, calls the reference function, then re-seeds and REPLAYS the same numpy calls in the same order to recover exactly
the draws the function consumed.  The fixture therefore holds data only: inputs are re-creatable from seeds, draws and
outputs are stored.  Runs only in the build container (imports /root/reference through tests/golden/refshim stand-ins for
torchvision / PIL / cv2 / matplotlib / skimage, none of which compute anything)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.path.insert(1, "/root/reference")
import utils.degradation_utils as RD  # noqa: E402  (the reference)
import utils.image_utils as RI  # noqa: E402

SHAPE = (9, 32, 32)


def clean(seed=1):
    return np.random.RandomState(seed).rand(*SHAPE).astype(np.float32)


def haze_entries(D):
    """the reference's `_simulate_haze` (degradation_utils.py:235-283) itself: its cirrus maps are .mat DATA files the image does not
    hold, so the folder listing and `sio.loadmat` are answered with a seeded synthetic map of the patch size (then cv2.resize is
    the identity: refshim/cv2.py).  What gets pinned is the function's arithmetic: atmospheric light from the top pixels per band,
    t1 = 1 - omega * cirrus with its 1e-10 floor, the wavelength exponent, the blend."""
    out = {}
    x = clean(3)
    rs = np.random.RandomState(31)
    cirrus = (rs.rand(SHAPE[1], SHAPE[2]) * 1.6).astype(np.float64)          # values > 1 / omega exercise the t1 <= 0 floor
    real_listdir, real_loadmat = RD.os.listdir, RD.sio.loadmat
    RD.os.listdir = lambda folder: ["synthetic.mat"]
    RD.sio.loadmat = lambda path: {"haze": cirrus}
    try:
        for i, (omega, gamma) in enumerate(((0.75, 1.0), (0.2, 1.0), (0.9, 0.5))):
            out["haze%d/out" % i] = D._simulate_haze(x.copy(), omega=omega, gamma=gamma)
            out["haze%d/omega_gamma" % i] = np.array([omega, gamma])
    finally:
        RD.os.listdir, RD.sio.loadmat = real_listdir, real_loadmat
    out["haze/cirrus"] = cirrus
    return out


def main():
    D = RD.Degradation(None)
    out = {}
    C, H, W = SHAPE
    x = clean()
    # gaussian
    np.random.seed(11); y = D._add_gaussian_noise(x.copy(), (30, 70))
    np.random.seed(11); sig = np.random.uniform(30, 70) / 255; noise = np.random.randn(*SHAPE)
    out.update({"gauss/sigma": np.array(sig), "gauss/noise": noise.astype(np.float64), "gauss/out": y})
    # non-iid
    sigmas = (10, 30, 50, 70)
    np.random.seed(12); y = D._add_gaussian_noise_non_iid(x.copy(), sigmas)
    np.random.seed(12); bs = (np.array(sigmas) / 255.)[np.random.randint(0, len(sigmas), C)]; noise = np.random.randn(*SHAPE)
    out.update({"noniid/band_sigma": bs, "noniid/noise": noise, "noniid/out": y})
    # stripe
    np.random.seed(13); y = D._add_stripe_noise(x.copy(), 0.05, 0.15)
    np.random.seed(13)
    allb = np.random.permutation(range(C)); nb = int(np.floor(C / 3)); bands = allb[:nb]
    ns = np.random.randint(np.floor(0.05 * W), np.floor(0.15 * W), len(bands))
    locs, vals = [], []
    for b, n in zip(bands, ns):
        loc = np.random.permutation(range(W))[:n]
        locs.append(loc); vals.append(np.random.uniform(0, 1, size=(len(loc),)) * 0.5 - 0.25)
    out.update({"stripe/bands": bands, "stripe/n": ns, "stripe/out": y})
    for i in range(len(bands)):
        out["stripe/loc%d" % i], out["stripe/val%d" % i] = locs[i], vals[i]
    # deadline
    np.random.seed(14); y = D._add_deadline_noise(x.copy(), 0.05, 0.15)
    np.random.seed(14)
    bands = np.random.permutation(C)[:nb]
    nd = np.random.randint(np.ceil(0.05 * W), np.ceil(0.15 * W), len(bands))
    out.update({"deadline/bands": bands, "deadline/n": nd, "deadline/out": y})
    for i, n in enumerate(nd):
        out["deadline/loc%d" % i] = np.random.permutation(range(W))[:n]
    # impulse
    np.random.seed(15); y = D._add_impulse_noise(x.copy(), 0.3)
    np.random.seed(15)
    bands = np.random.permutation(C)[:nb]
    fl, sa = [], []
    for _ in bands:
        fl.append(np.random.choice([True, False], size=(H, W), p=[0.3, 0.7]))
        sa.append(np.random.choice([True, False], size=(H, W), p=[0.5, 0.5]))
    out.update({"impulse/bands": bands, "impulse/flipped": np.array(fl), "impulse/salted": np.array(sa), "impulse/out": y})
    # deterministic: blurs, sr, resize
    for k in (7, 9, 15):
        out["gblur%d/out" % k] = D._apply_gaussian_blur(x.copy(), k)
    out["cblur9/out"] = D._apply_circle_blur(x.copy(), 9)
    out["sblur5/out"] = D._apply_square_blur(x.copy(), 5)
    for f in (2, 4, 8):
        lo = D._bicubic_downsample(x.copy(), f)
        out["sr%d/low" % f] = lo
        out["sr%d/out" % f] = D._resize(lo, f)
    # mask / band loss
    np.random.seed(16); y = D._apply_random_mask(x.copy(), 0.8)
    np.random.seed(16); u = np.random.rand(*SHAPE)
    out.update({"mask/u": u, "mask/out": y})
    np.random.seed(17); y = D._simulate_band_loss(x.copy(), 0.3)
    np.random.seed(17); lost = np.random.choice(C, int(0.3 * C), replace=False)
    out.update({"bandloss/lost": lost, "bandloss/out": y})
    # augmentation modes, band interpolation
    for m in range(8):
        out["aug%d/out" % m] = np.ascontiguousarray(RI.data_augmentation(x.copy(), m))
    out["interp31/out"] = RI.interpolate_bands(x.copy(), 31)[0]
    out.update(haze_entries(D))
    np.savez_compressed(os.path.join(HERE, "degrade.npz"), **out)
    print("wrote degrade.npz with", len(out), "entries")


def add_haze_only():
    """adds the haze entries to an existing degrade.npz (everything else regenerates bitwise, this just saves the minute)"""
    path = os.path.join(HERE, "degrade.npz")
    out = dict(np.load(path))
    out.update(haze_entries(RD.Degradation(None)))
    np.savez_compressed(path, **out)
    print("degrade.npz: haze entries added,", len(out), "entries")


if __name__ == "__main__":
    add_haze_only() if sys.argv[1:] == ["haze"] else main()
