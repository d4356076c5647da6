"""Diagnostic: does torch._foreach_copy_ take its multi-tensor kernel for (arena view <- fresh gradient) pairs?"""
import torch
from torch.profiler import ProfilerActivity, profile
dev = "cuda"
shapes = [(128, 128, 1, 1), (384, 128), (128,), (225, 2), (2, 1, 1), (128, 8), (1, 1, 128, 8), (704, 128), (384, 1, 3, 3)] * 60
total = sum((torch.Size(s).numel() + 3) // 4 * 4 for s in shapes)
flat = torch.zeros(total, device=dev)
views, o = [], 0
for s in shapes:
    n = torch.Size(s).numel()
    views.append(flat[o:o + n].view(s))
    o += (n + 3) // 4 * 4
grads = [torch.randn(s, device=dev) for s in shapes]
for name, g in (("contiguous", grads), ("one transposed", grads[:-1] + [torch.randn(3, 3, 1, 384, device=dev).permute(3, 2, 0, 1)])):
    torch._foreach_copy_(views, g)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        torch._foreach_copy_(views, g)
        torch.cuda.synchronize()
    ks = [(e.key[:70], e.count) for e in prof.key_averages() if e.device_time_total > 0]
    print(name, len(shapes), "pairs ->", ks)
