import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = "cuda"
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for shape, batched in [((34, 704, 128), False), ((64, 128, 128), False), ((32, 6, 128, 128), True), ((2048, 2, 128), False), ((2048, 225, 2), False), ((1024, 9, 384), False)]:
    p = torch.randn(shape, device=dev)
    a = t_us(lambda: ops.reduce_parts(p, batched=batched))
    b = t_us(lambda: p.sum(1 if batched else 0))
    print(shape, "reduce_parts %.1f us   torch.sum %.1f us   MB %.1f" % (a, b, p.numel() * 4 / 1e6))
ps = [torch.randn(s, device=dev) for s in [(34, 704, 128), (64, 128, 128), (64, 384, 128), (2048, 2, 128), (2048, 225, 2), (1024, 9, 384), (34, 704), (64, 128)]]
def scoped():
    with ops.reduce_scope():
        for p in ps: ops.reduce_parts(p)
print("scope of 8: %.1f us; torch sums: %.1f us" % (t_us(scoped), t_us(lambda: [p.sum(0) for p in ps])))
