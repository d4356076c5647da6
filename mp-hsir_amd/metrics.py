"""Band-wise PSNR / SSIM on the device: compute_psnr_ssim / compute_psnr_ssim2 of the reference
(utils/val_utils.py:49-105) without skimage and without a host round trip per band.

The reference loops over images and bands on the CPU calling skimage's peak_signal_noise_ratio and structural_similarity
(data_range=1, defaults: 7x7 uniform window, sample covariance, mean over the interior).  Here both are batched tensor
programs over (B,C,H,W): the 7x7 window means are one avg_pool2d each ("valid" windows = exactly the interior skimage
averages over).  Accumulation in float64 like skimage.  PyTorch device ops (evaluation side of the boundary)."""
import torch
import torch.nn.functional as F


def _clip01(t):
    return t.detach().double().clamp(0, 1)


def psnr_bands(restored, clean):
    """(B,C) band-wise PSNR, data_range 1."""
    mse = ((_clip01(restored) - _clip01(clean)) ** 2).mean(dim=(-1, -2))
    return 10.0 * torch.log10(1.0 / mse)


def ssim_bands(restored, clean, win=7, k1=0.01, k2=0.03):
    """(B,C) band-wise SSIM with skimage's defaults (uniform window, sample covariance NP/(NP-1), interior mean)."""
    x, y = _clip01(restored), _clip01(clean)
    B, C, H, W = x.shape
    x, y = x.reshape(B * C, 1, H, W), y.reshape(B * C, 1, H, W)
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    pool = lambda t: F.avg_pool2d(t, win, stride=1)
    ux, uy = pool(x), pool(y)
    vx = cov_norm * (pool(x * x) - ux * ux)
    vy = cov_norm * (pool(y * y) - uy * uy)
    vxy = cov_norm * (pool(x * y) - ux * uy)
    c1, c2 = k1 ** 2, k2 ** 2
    s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    return s.mean(dim=(-1, -2, -3)).reshape(B, C)


def compute_psnr_ssim(restored, clean):
    """-> (mean PSNR, mean SSIM, B): mean over bands, then over images (val_utils.py:49-69)."""
    assert restored.shape == clean.shape
    return float(psnr_bands(restored, clean).mean(1).mean()), float(ssim_bands(restored, clean).mean(1).mean()), restored.shape[0]


def compute_psnr_ssim2(restored, clean, degraded=None):
    """compute_psnr_ssim2 (val_utils.py:71-105): with `degraded`, only bands that are entirely zero in it are scored (band
    completion, test mode 10); images without such a band do not count.  -> (psnr, ssim, count)."""
    if degraded is None:
        return compute_psnr_ssim(restored, clean)
    use = (degraded == 0).flatten(2).all(dim=2)                     # (B,C)
    p, s = psnr_bands(restored, clean), ssim_bands(restored, clean)
    n = use.sum(1)
    ok = n > 0
    count = int(ok.sum())
    if count == 0:
        return 0.0, 0.0, 0
    pm = (p * use).sum(1)[ok] / n[ok]
    sm = (s * use).sum(1)[ok] / n[ok]
    return float(pm.mean()), float(sm.mean()), count
