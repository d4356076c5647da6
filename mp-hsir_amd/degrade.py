"""GPU-side degradation synthesiser: the reference's training / test degradations (utils/degradation_utils.py:25-284,
utils/dataset_utils.py:277-879) as batched tensor programs on the device the model runs on.

The reference degrades one (C,H,W) numpy patch at a time on DataLoader workers, drawing from the process-global
numpy / `random` state.  At thousands of patches per second per GPU that is the bottleneck, so here every
degradation is a *pure function of the clean cube and explicit random draws*:

    stripe_noise(x, bands, locs, vals)      # what the reference computes once np.random has produced bands/locs/vals

and `Draws` produces those draws on the device from a seeded torch.Generator.  The split is what makes parity
checkable: tests/golden/make_degrade_golden.py runs the REFERENCE functions under a seeded numpy state, replays the
same numpy calls to recover the draws they consumed, and stores (draws, output); the oracle restatement
(oracle/degrade_oracle.py) and these functions must reproduce the outputs from the draws (tests/test_degrade.py).

All functions take and return (B,C,H,W) float32 cubes; per-sample draws carry a leading batch axis.  PyTorch device ops
only (this is the loader side of the boundary, not the model hot path).  `file:line` = reference repository.
"""
import math

import torch
import torch.nn.functional as F

# ---- the degradation menu of ImageTransformDataset (utils/dataset_utils.py:112-122) -----------------------------------
DE_DICT = {
    "natural_scene": {"gaussianN": [(30, 70)], "complexN": [(10, 30, 50, 70), (0.05, 0.15), (0.1, 0.3, 0.5, 0.7), (0.05, 0.15)],
                      "blur": [(9, 15, 21)], "sr": [(2, 4, 8)], "inpaint": [(0.7, 0.8, 0.9)], "bandmiss": [(0.1, 0.2, 0.3)],
                      "motion_blur": [((15, 45),)]},
    "remote_sensing": {"gaussianN": [(30, 70)], "complexN": [(10, 30, 50, 70), (0.05, 0.15), (0.1, 0.3, 0.5, 0.7), (0.05, 0.15)],
                       "blur": [(7, 11, 15)], "sr": [(2, 4, 8)], "inpaint": [(0.7, 0.8, 0.9)], "haze": [(0.5, 0.75, 1)],
                       "bandmiss": [(0.1, 0.2, 0.3)], "circle_blur": [(9,)], "poissonN": [(10,)]},
}


# ---- deterministic kernels ------------------------------------------------------------------------------------------
def gaussian_kernel2d(k):
    """(k,k) separable Gaussian of _apply_gaussian_blur (degradation_utils.py:95-102): sigma = 0.3((k-1)/2 - 1) + 0.8."""
    sigma = 0.3 * ((k - 1) * 0.5 - 1) + 0.8
    x = torch.arange(k, dtype=torch.float32)
    k1 = torch.exp(-((x - (k - 1) / 2) ** 2) / (2 * sigma ** 2))
    k1 = k1 / k1.sum()
    return k1.unsqueeze(0) * k1.unsqueeze(1)


def circle_kernel2d(k):
    """_apply_circle_blur (degradation_utils.py:114-123): Gaussian of sigma = radius inside the disc of radius k//2."""
    r = k // 2
    yy, xx = torch.meshgrid(torch.arange(k, dtype=torch.float32), torch.arange(k, dtype=torch.float32), indexing="ij")
    d = torch.sqrt((xx - r) ** 2 + (yy - r) ** 2)
    ker = torch.where(d <= r, torch.exp(-(d ** 2) / (2 * (r ** 2))), torch.zeros(()))
    return ker / ker.sum()


def square_kernel2d(k):
    return torch.full((k, k), 1.0 / (k * k))


def motion_kernel2d(k, angle):
    """_apply_motion_blur (degradation_utils.py:137-143): a horizontal line of 1/k through row (k-1)//2, rotated by `angle`
    degrees about (k/2, k/2) with cv2.getRotationMatrix2D + cv2.warpAffine (bilinear, zero border).  cv2 is not available
    offline, so this restates its published semantics: dst(x,y) = src(M^-1 (x,y)), M = [[a, b, (1-a)cx - b cy],
    [-b, a, b cx + (1-a) cy]], a = cos, b = sin.  Pinned at the angles where the warp maps pixel centres onto pixel centres (0, 90, 180,
    270: tests/test_degrade.py::test_motion_kernel_exact_angles); in between OpenCV rounds source coordinates to 1/32 pixel, which
    this float restatement does not reproduce (differences of the order of 1e-2 of a tap: unpinned, no cv2 to compare with)."""
    src = torch.zeros((k, k), dtype=torch.float64)
    src[int((k - 1) / 2), :] = 1.0 / k
    a, b = math.cos(math.radians(angle)), math.sin(math.radians(angle))
    cx = cy = k / 2
    m = torch.tensor([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], dtype=torch.float64)
    minv = torch.linalg.inv(torch.cat([m, torch.tensor([[0.0, 0.0, 1.0]], dtype=torch.float64)]))
    ys, xs = torch.meshgrid(torch.arange(k, dtype=torch.float64), torch.arange(k, dtype=torch.float64), indexing="ij")
    sx = minv[0, 0] * xs + minv[0, 1] * ys + minv[0, 2]
    sy = minv[1, 0] * xs + minv[1, 1] * ys + minv[1, 2]
    x0, y0 = torch.floor(sx), torch.floor(sy)
    out = torch.zeros((k, k), dtype=torch.float64)
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = (x0 + dx).long(), (y0 + dy).long()
            w = (1 - (sx - x0 - dx).abs()) * (1 - (sy - y0 - dy).abs())
            ok = (xi >= 0) & (xi < k) & (yi >= 0) & (yi < k)
            out += torch.where(ok, src[yi.clamp(0, k - 1), xi.clamp(0, k - 1)] * w, torch.zeros((), dtype=torch.float64))
    return out.float()


def blur(x, kernel2d):
    """depthwise 2-D correlation with zero padding k//2 (F.conv2d, groups = bands), every *_blur of the reference."""
    C = x.shape[1]
    k = kernel2d.shape[-1]
    w = kernel2d.to(x.device, x.dtype).reshape(1, 1, k, k).expand(C, 1, k, k)
    return F.conv2d(x, w, padding=k // 2, groups=C)


def bicubic_downsample(x, factor):
    """_bicubic_downsample (degradation_utils.py:170-181): F.interpolate(bicubic, align_corners=True) to (H//f, W//f)."""
    H, W = x.shape[-2:]
    return F.interpolate(x, size=(H // factor, W // factor), mode="bicubic", align_corners=True)


def resize_nearest(x, factor):
    """_resize (degradation_utils.py:195-206): every low-resolution pixel replicated factor x factor times."""
    return x.repeat_interleave(factor, dim=-2).repeat_interleave(factor, dim=-1)


def super_resolution_input(x, factor):
    """the 'sr' degradation as the loaders emit it (single_degrade :427-428; Super_Resolution_Dataset :697-711)."""
    return resize_nearest(bicubic_downsample(x, factor), factor)


# ---- noise / masking given explicit draws ----------------------------------------------------------------------------
def gaussian_noise(x, sigma, noise):
    """_add_gaussian_noise (:25-31): x + N(0,1) * sigma, sigma (B,) already divided by 255."""
    return x + noise * sigma.reshape(-1, 1, 1, 1)


def gaussian_noise_non_iid(x, band_sigma, noise):
    """_add_gaussian_noise_non_iid (:33-39): one sigma per band, band_sigma (B,C) already divided by 255."""
    return x + noise * band_sigma[:, :, None, None]


def stripe_noise(x, band_mask, col_offset):
    """_add_stripe_noise (:41-55): in the chosen bands, the chosen columns are lowered by a per-column constant.
    band_mask (B,C) bool; col_offset (B,C,W) = the stripe value at the chosen (band, column) pairs, 0 elsewhere."""
    return x - (col_offset * band_mask[:, :, None])[:, :, None, :]


def deadline_noise(x, col_dead):
    """_add_deadline_noise (:57-68): chosen columns of chosen bands are zeroed.  col_dead (B,C,W) bool."""
    return x * (~col_dead)[:, :, None, :]


def impulse_noise(x, flipped, salted):
    """_add_impulse_noise (:70-84): where flipped, the pixel becomes 1 (salted) or 0 (peppered).  (B,C,H,W) bools, False in
    bands that were not chosen."""
    return torch.where(flipped, salted.to(x.dtype), x)


def random_mask(x, u, ratio):
    """_apply_random_mask (:235-241): keep where U[0,1) > ratio.  ratio (B,)."""
    return x * (u > ratio.reshape(-1, 1, 1, 1))


def band_loss(x, lost):
    """_simulate_band_loss (:285-293): lost (B,C) bool -> those bands are zero."""
    return x * (~lost)[:, :, None, None]


def haze(x, cirrus, omega, gamma=1.0, top_percent=0.01):
    """_simulate_haze (:243-283) given the cirrus-band map already resized to (H,W): cirrus (B,H,W), omega (B,).
    atmospheric light = mean of the top max(int(HW*top_percent/100),1) pixels per band; t1 = 1 - omega*cirrus (<= 0 -> 1e-10);
    transmission_c = t1 ** ((lambda_0/lambda_c) ** gamma) with lambda = linspace(400,1000,100) (so C <= 100)."""
    B, C, H, W = x.shape
    top_k = max(int(H * W * top_percent / 100), 1)
    atm = x.reshape(B, C, -1).topk(top_k, dim=-1).values.mean(-1)                       # (B,C)
    t1 = 1 - omega.reshape(B, 1, 1) * cirrus
    t1 = torch.where(t1 <= 0, torch.full_like(t1, 1e-10), t1)
    lam = torch.linspace(400, 1000, 100, device=x.device, dtype=torch.float64)[:C]
    ratio = ((lam[0] / lam) ** gamma).to(x.dtype)                                       # (C,)
    trans = torch.exp(ratio.reshape(1, C, 1, 1) * torch.log(t1).unsqueeze(1))
    return x * trans + atm[:, :, None, None] * (1 - trans)


def poisson_noise(x, scale, generator=None):
    """_apply_poisson (:86-89): Poisson(clip(x,0) * scale) / scale (distribution-level parity only)."""
    return torch.poisson(x.clamp_min(0) * scale, generator=generator) / scale


def augment(x, mode):
    """data_augmentation (utils/image_utils.py:141-176) per sample: mode (B,) in 0..7 = rot90 by mode//2 (counter-clockwise,
    axes (H,W)) then an up-down flip when mode is odd.  Square patches."""
    out = torch.empty_like(x)
    for m in range(8):
        sel = (mode == m).nonzero(as_tuple=True)[0]
        if sel.numel() == 0:
            continue
        t = torch.rot90(x[sel], k=m // 2, dims=(-2, -1))
        out[sel] = t.flip(-2) if m % 2 else t
    return out


def interpolate_bands(x, target_bands):
    """interpolate_bands (utils/image_utils.py:597-619): place the C source bands at round(linspace(0, T-1, C)) and fill the
    gaps linearly between neighbouring source bands."""
    B, C, H, W = x.shape
    idx = torch.round(torch.linspace(0, target_bands - 1, C, dtype=torch.float64)).long().tolist()
    out = torch.zeros((B, target_bands, H, W), dtype=x.dtype, device=x.device)
    out[:, idx] = x
    for i in range(C - 1):
        s, e = idx[i], idx[i + 1]
        n = e - s
        for j in range(1, n):
            pos = j / n
            out[:, s + j] = x[:, i] * (1 - pos) + x[:, i + 1] * pos
    return out


# ---- draws ---------------------------------------------------------------------------------------------------------------
class Draws:
    """Random draws of the degradations on the device, from one seeded torch.Generator (replaces the reference's global
    numpy / random state).  Distributions follow the reference's calls; the streams differ."""

    def __init__(self, device, seed):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(seed)

    def rand(self, *shape):
        return torch.rand(shape, generator=self.gen, device=self.device)

    def randn(self, *shape):
        return torch.randn(shape, generator=self.gen, device=self.device)

    def randint(self, lo, hi, shape):
        return torch.randint(lo, hi, tuple(shape), generator=self.gen, device=self.device)

    def choice(self, values, n):
        v = torch.tensor(values, dtype=torch.float32, device=self.device)
        return v[self.randint(0, len(values), (n,))]

    def band_subset(self, B, C, count):
        """(B,C) bool with exactly `count` bands per sample: np.random.permutation(C)[:count]."""
        order = self.rand(B, C).argsort(dim=1)
        return order.argsort(dim=1) < count

    def column_subsets(self, B, C, W, n):
        """(B,C,W) bool with n[b,c] columns chosen uniformly without replacement per (sample, band)."""
        rank = self.rand(B, C, W).argsort(dim=2).argsort(dim=2)
        return rank < n[:, :, None]


def complex_noise(x, d, sigmas=(10, 30, 50, 70), deadline=(0.05, 0.15), impulse=(0.1, 0.3, 0.5, 0.7), stripe=(0.05, 0.15)):
    """'complexN' (:303-318): non-iid Gaussian noise, then per sample ONE of deadline / impulse / stripe noise on a random
    third of the bands.  Returns (degraded, type_idx (B,))."""
    B, C, H, W = x.shape
    nb = int(math.floor(C / 3))
    y = gaussian_noise_non_iid(x, d.choice([s / 255.0 for s in sigmas], B * C).reshape(B, C), d.randn(B, C, H, W))
    kind = d.randint(0, 3, (B,))
    bands = d.band_subset(B, C, nb)
    # deadline: n ~ randint(ceil(min W), ceil(max W)) columns zeroed
    nd = d.randint(math.ceil(deadline[0] * W), max(math.ceil(deadline[1] * W), math.ceil(deadline[0] * W) + 1), (B, C))
    dead = d.column_subsets(B, C, W, nd) & bands[:, :, None] & (kind == 0).reshape(B, 1, 1)
    y = deadline_noise(y, dead)
    # impulse: amount chosen per sample from the list, salt vs pepper 0.5
    amt = d.choice(list(impulse), B).reshape(B, 1, 1, 1)
    flipped = (d.rand(B, C, H, W) < amt) & bands[:, :, None, None] & (kind == 1).reshape(B, 1, 1, 1)
    y = impulse_noise(y, flipped, d.rand(B, C, H, W) < 0.5)
    # stripe: n ~ randint(floor(min W), floor(max W)) columns lowered by U(0,1)/2 - 1/4
    ns = d.randint(math.floor(stripe[0] * W), max(math.floor(stripe[1] * W), math.floor(stripe[0] * W) + 1), (B, C))
    cols = d.column_subsets(B, C, W, ns)
    off = (d.rand(B, C, W) * 0.5 - 0.25) * cols
    y = stripe_noise(y, bands & (kind == 2).reshape(B, 1), off)
    return y, kind


class DegradationSynthesizer:
    """ImageTransformDataset.__getitem__ (utils/dataset_utils.py:128-146) for a whole batch on the device: per sample a task
    id de_id ~ U{0..T-1}, the degradation of that task with its range from DE_DICT, one of 7 flip/rotation augmentations
    applied to both cubes; emits `degrad_patch, clean_patch, prompt (B,1) int64`.  `cirrus` (a callable (B,H,W) -> maps in
    [0,1]) supplies the haze maps the reference reads from .mat files (:245-257); default: smooth synthetic fields."""

    def __init__(self, data_type, de_types, device, seed=2024, cirrus=None):
        self.table, self.de_types = DE_DICT[data_type], list(de_types)
        for t in self.de_types:
            if t not in self.table:
                raise ValueError("degradation %r is not defined for %s" % (t, data_type))
        self.d = Draws(device, seed)
        self.cirrus = cirrus or self._synthetic_cirrus

    def _synthetic_cirrus(self, B, H, W):
        low = self.d.rand(B, 1, max(H // 16, 2), max(W // 16, 2))
        return F.interpolate(low, size=(H, W), mode="bilinear", align_corners=True)[:, 0]

    def degrade_as(self, x, de_type):
        """one degradation type for the whole batch x (B,C,H,W) -> degraded (B,C,H,W)"""
        d, rng = self.d, self.table[de_type]
        B, C, H, W = x.shape
        if de_type == "gaussianN":
            lo, hi = rng[0]
            return gaussian_noise(x, (lo + (hi - lo) * d.rand(B)) / 255.0, d.randn(B, C, H, W))
        if de_type == "complexN":
            return complex_noise(x, d, *rng)[0]
        if de_type in ("blur", "circle_blur", "motion_blur"):
            out = torch.empty_like(x)
            pick = d.randint(0, len(rng[0]), (B,))
            for i, k in enumerate(rng[0]):
                sel = (pick == i).nonzero(as_tuple=True)[0]
                if sel.numel():
                    ker = gaussian_kernel2d(k) if de_type == "blur" else circle_kernel2d(k) if de_type == "circle_blur" else motion_kernel2d(*k)
                    out[sel] = blur(x[sel], ker)
            return out
        if de_type == "sr":
            out = torch.empty_like(x)
            pick = d.randint(0, len(rng[0]), (B,))
            for i, f in enumerate(rng[0]):
                sel = (pick == i).nonzero(as_tuple=True)[0]
                if sel.numel():
                    out[sel] = super_resolution_input(x[sel], f)
            return out
        if de_type == "inpaint":
            return random_mask(x, d.rand(B, C, H, W), d.choice(list(rng[0]), B))
        if de_type == "bandmiss":
            frac = d.choice(list(rng[0]), B)
            n = (frac * C).long()                                                         # int(loss_percentage * B)
            return band_loss(x, d.rand(B, C).argsort(dim=1).argsort(dim=1) < n[:, None])
        if de_type == "haze":
            return haze(x, self.cirrus(B, H, W), d.choice(list(rng[0]), B))
        if de_type == "poissonN":
            return poisson_noise(x, float(rng[0][0]), generator=d.gen)
        raise ValueError("Invalid degradation type " + de_type)

    def __call__(self, clean):
        """clean (B,C,H,W) on the device -> (degraded, clean_augmented, prompt (B,1) int64)"""
        B = clean.shape[0]
        de_id = self.d.randint(0, len(self.de_types), (B,))
        degraded = torch.empty_like(clean)
        for t, name in enumerate(self.de_types):
            sel = (de_id == t).nonzero(as_tuple=True)[0]
            if sel.numel():
                degraded[sel] = self.degrade_as(clean[sel], name)
        mode = self.d.randint(1, 8, (B,))                                                 # random.randint(1, 7)
        return augment(degraded, mode), augment(clean, mode), de_id.reshape(B, 1)
