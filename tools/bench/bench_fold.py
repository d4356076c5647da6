import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = torch.device("cuda")
def t_us(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, C, heads, nsplit) in [(32, 64, 2, 16), (32, 64, 2, 1), (32, 128, 2, 16), (32, 128, 4, 4), (32, 256, 8, 1), (32, 256, 8, 2), (1, 64, 2, 64)]:
    hd = C // heads
    gp = torch.randn(B, nsplit, heads, hd, hd, device=dev); sp = torch.rand(B, nsplit, 2, C, device=dev) + 1
    temp = torch.ones(heads, device=dev); wo = torch.randn(C, C, device=dev) * C ** -0.5
    f = lambda: ops.spectral_fold(gp, sp, temp, wo, torch.bfloat16)
    ft = lambda: ops.spectral_fold(gp, sp, temp, wo, torch.bfloat16, transposed=True)
    e = lambda: torch.empty((B, C, C), dtype=torch.bfloat16, device=dev)
    print("B=%d C=%d heads=%d nsplit=%d: fold %.1f us, training form %.1f us, (torch.empty alone %.1f us)" % (B, C, heads, nsplit, t_us(f), t_us(ft), t_us(e)), flush=True)
