"""diagnostic (GPU box): fused GDFN against the three-launch chain at the 512x512 forward's two big shapes."""
import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops

dev = torch.device("cuda"); dt = torch.bfloat16


def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (B, H, W, D, hid) in [(1, 512, 512, 128, 340), (1, 256, 256, 256, 680), (1, 512, 512, 64, 170), (16, 64, 64, 128, 340), (1, 512, 512, 192, 510)]:
    HP = ops.round_up(hid, 32)
    x = torch.randn(B * H * W, D, device=dev, dtype=dt)
    w_in = (torch.randn(2 * HP, D, device=dev) * D ** -0.5).to(dt)
    w9 = torch.randn(9, 2 * HP, device=dev) / 3
    w_out = (torch.randn(D, HP, device=dev) * HP ** -0.5).to(dt)
    ln = (torch.ones(D, device=dev), torch.zeros(D, device=dev))
    three = lambda: ops.gemm_tok(ops.dwconv_gate(ops.gemm_tok(x, w_in, ln=ln), w9, B, H, W), w_out, epi=1, res=x)
    print("B=%d %dx%d D=%d HP=%d: three launches %.1f us" % (B, H, W, D, HP, t_us(three)), end="")
    tw = 16 if D <= 128 else 8
    tiles = (H // 8) * (W // tw)
    for ns in (None, 128, 256, 512, 1024):
        if ns is not None and (tiles % max(1, ns // B) or ns // B < 1):
            continue
        f = lambda: ops.gdfn_fused(x, ln, w_in, w9, w_out, B, H, W, nsplit=None if ns is None else ns // B)
        print("  fused[wgs=%s] %.1f" % (ns, t_us(f)), end="")
    print(flush=True)
