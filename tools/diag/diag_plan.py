import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, '/root/repo/tests')
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops, engine
from torch.utils._pytree import tree_flatten
import model_checks as M
from golden.cases import TINY_CFG
from golden.detfill import surrogate_clip_prompt
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
from mp_hsir_amd.data import SyntheticPatchSource
dev = "cuda"
net = MP_HSIR_Net(**TINY_CFG, clip_prompt=surrogate_clip_prompt(6), compute_dtype=torch.bfloat16).to(dev).eval()
eng = engine.DataParallelEngine(net, lr=1e-3, use_pack_plan=False)
src = SyntheticPatchSource(8, 64, 2, 6, dev, 2024, 0)
_, x, c, p = src.next()
eng.train_step(x, c, p); eng.train_step(x, c, p)
flat = eng.flat_p
lo_ptr, hi_ptr = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()
inside = lambda t: lo_ptr <= t.data_ptr() < hi_ptr
caches = ops.WeightCache.live()
print("caches", len(caches), "with last", sum(c.last is not None for c in caches))
n_out = 0
for c in caches:
    if c.last is None: continue
    out = [tuple(p.shape) for p in c.last[0] if not inside(p)]
    if out:
        n_out += 1
        print("cache with params outside the arena:", out[:4], "of", len(c.last[0]))
print("n_out", n_out, "arena elems", flat.numel(), "unused", len(eng.unused))
plan = engine.PackPlan(flat).build()
print("pinned", plan.pinned, "skipped", plan.skipped, {k: v[0].numel() for k, v in plan.groups.items()})
for w in plan.why[:12]:
    print(w)
