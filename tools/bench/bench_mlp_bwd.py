import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = "cuda"; dt = torch.bfloat16
def t_us(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for M, C, hid in [(131072, 128, 340), (32768, 128, 340), (131072, 64, 170), (8192, 256, 680), (65536, 192, 510), (16384, 192, 510), (65536, 96, 255), (4096, 384, 1021)]:
    x, dy = torch.randn(M, C, device=dev, dtype=dt), torch.randn(M, C, device=dev, dtype=dt)
    fc1w, fc1b, fc2w = torch.randn(2 * hid, C, device=dev) * C ** -0.5, torch.randn(2 * hid, device=dev) * 0.1, torch.randn(C, hid, device=dev) * hid ** -0.5
    lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    W1, b1, W2 = ops.pack_gated_mlp(fc1w, fc1b, fc2w, dt)
    W1T, W2T = W1.t().contiguous(), W2.t().contiguous()
    HP = W2.shape[1]
    fl = 12.0 * M * C * HP
    res = []
    for v in (0, 1, 2, 3, 4):
        try:
            us = t_us(lambda: ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T, variant=v))
            res.append("v%d %7.1f us %6.1f TF/s" % (v, us, fl / us / 1e6))
        except Exception as e:
            res.append("v%d failed" % v)
    print("M=%d C=%d: " % (M, C) + " | ".join(res))
