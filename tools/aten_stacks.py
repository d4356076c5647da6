import os, sys, torch
from torch.profiler import ProfilerActivity, profile
sys.path.insert(0, '/root/repo')
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.train_step(x, c, p)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_stack_n=8):
    if e.key in ("aten::clone", "aten::contiguous", "aten::cat", "aten::_to_copy", "aten::add", "aten::add_") and e.device_time_total > 0:
        st = [s for s in e.stack if "mp_hsir_amd" in s or "mp-hsir_amd" in s][:2]
        rows.append((e.device_time_total, e.key, e.count, st))
rows.sort(key=lambda r: -r[0])
for t, k, n, st in rows[:30]:
    print("%.3f ms %-18s x%d  %s" % (t / 1e3, k, n, " <- ".join(s.split("/")[-1] for s in st)))
