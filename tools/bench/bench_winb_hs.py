"""diagnostic (GPU box): win_attn_bwd with the heads of a window dealt to 1, 2, 4, 8 workgroups (head_split)."""
import os, sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
from mp_hsir_amd.net.MP_HSIR import PGSSTB
dev = torch.device("cuda"); dt = torch.bfloat16
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H, C, heads) in [(32, 16, 256, 8), (32, 32, 128, 4), (16, 16, 256, 8), (1, 128, 256, 8), (32, 64, 128, 2), (32, 64, 64, 2)]:
    blk = PGSSTB(C, heads, [64, 64], 8, 4, 0.0, 2.66, 8, 128).to(dev)
    pk = blk.packed(dt)
    x = torch.randn(B, H, H, C, device=dev, dtype=dt); dsa = torch.randn(B, H, H, C, device=dev, dtype=dt)
    dmu = torch.randn(B * H * H // 64, C, device=dev)
    ref = None
    for hs in (1, 2, 4, 8):
        if heads % hs: continue
        f = lambda: ops.win_attn_bwd(x, dsa, dmu, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wprojT"], heads, 4, head_split=hs)
        out = f()
        if ref is None: ref = out
        ok = all(torch.equal(a, b) for a, b in zip(out, ref))
        print("B=%d %dx%d C=%d heads=%d hsplit=%d: %.1f us %s" % (B, H, H, C, heads, hs, t_us(f), "" if ok else "(!=)"), flush=True)
