import os, sys, torch
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    eng.train_step(x, c, p)
    torch.cuda.synchronize()
import collections
agg = collections.OrderedDict()
for e in prof.events():
    if not e.name.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    st = [f for f in (e.stack or []) if "hsir" in f or "bench" in f][:3]
    key = (e.name, tuple(f.split("/")[-1] for f in st), tuple(e.input_shapes or ()) if hasattr(e, "input_shapes") else ())
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += e.self_device_time_total
print("glue ops with their own GPU time: %d launches, %.3f ms" % (sum(v[0] for v in agg.values()), sum(v[1] for v in agg.values()) / 1e3))
for (name, st, shp), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:100]:
    print("%.3f ms %-20s x%-3d %s" % (t / 1e3, name, n, " <- ".join(st)))
