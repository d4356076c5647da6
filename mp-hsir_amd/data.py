"""Synthetic patch source emitting the reference's training batch tuple on the GPU.

The reference's loader (utils/dataset_utils.py:102-146) yields
``([clean_name, de_id], degrad_patch (B,C,64,64) f32, clean_patch (B,C,64,64) f32, prompt (B,1) i64)``
from LMDB patches degraded on CPU workers.  Datasets are not available offline and 16 CPU workers
cannot feed an MI355X, so the bench/tests use this generator: clean ~ U[0,1) min-max normalised per
patch (utils/image_utils.py:437-439), a task id per sample, and the Gaussian-noise degradation
sigma ~ U(30,70)/255 (utils/degradation_utils.py:25-31).  Explicit torch.Generator seeding
(seed 2024 = options.py:7) replaces the reference's global numpy/random state.
"""
import torch


class SyntheticPatchSource:
    def __init__(self, bands=31, patch=64, batch=32, task_classes=6, device="cuda", seed=2024, rank=0):
        self.shape = (batch, bands, patch, patch)
        self.task_classes, self.device = task_classes, device
        self.gen = torch.Generator(device=device).manual_seed(seed + 7919 * rank)

    def next(self):
        B = self.shape[0]
        clean = torch.rand(self.shape, generator=self.gen, device=self.device)
        lo = clean.amin(dim=(1, 2, 3), keepdim=True)
        hi = clean.amax(dim=(1, 2, 3), keepdim=True)
        clean = (clean - lo) / (hi - lo)
        sigma = (30.0 + 40.0 * torch.rand((B, 1, 1, 1), generator=self.gen, device=self.device)) / 255.0
        degraded = clean + sigma * torch.randn(self.shape, generator=self.gen, device=self.device)
        prompt = torch.randint(0, self.task_classes, (B, 1), generator=self.gen, device=self.device)
        names = ["synthetic_%08d" % i for i in range(B)]
        return [names, prompt[:, 0]], degraded, clean, prompt
