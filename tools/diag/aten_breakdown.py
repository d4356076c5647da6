"""Diagnostic: which framework (aten) kernels does one eager training step launch?  torch.profiler, grouped by op."""
import os
import sys
import torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mp_hsir_amd.data import SyntheticPatchSource
from mp_hsir_amd.engine import DataParallelEngine
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net

dev = torch.device("cuda")
net = MP_HSIR_Net(compute_dtype=torch.bfloat16, clip_prompt="surrogate").to(dev).train()
eng = DataParallelEngine(net, lr=2e-4)
src = SyntheticPatchSource(31, 64, 32, 6, dev, 2024, 0)
for _ in range(3):
    _, x, c, p = src.next(); eng.train_step(x, c, p)
torch.cuda.synchronize()
_, x, c, p = src.next()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    eng.train_step(x, c, p)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda r: -r[2])
tot = 0.0
for k, n, t in rows[:40]:
    print("%-45s calls %5d  device %.3f ms" % (k, n, t / 1e3))
print("total aten device time %.3f ms" % (sum(r[2] for r in rows) / 1e3))
