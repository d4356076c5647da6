"""gated_mlp forward: four-wave (tiles_per_wave 1, 2) against eight-wave (3: one tile per wave, 4: two) workgroups at the widths /
token counts of the training step.  GPU box only."""
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


for (M, C, hid) in [(131072, 64, 170), (131072, 128, 340), (32768, 128, 340), (8192, 256, 680), (262144, 128, 340), (65536, 96, 255), (65536, 192, 510)]:
    HP = ops.round_up(hid, 32)
    x = torch.randn(M, C, device=dev, dtype=dt)
    W1 = (torch.randn(2 * HP, C, device=dev) * C ** -0.5).to(dt)
    W2 = (torch.randn(C, HP, device=dev) * HP ** -0.5).to(dt)
    b1, b2 = torch.zeros(2 * HP, device=dev), torch.zeros(C, device=dev)
    lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    ref = None
    res = []
    for tpw in (0, 1, 2, 3, 4):
        try:
            y = ops.gated_mlp_fwd(x, lw, lb, W1, b1, W2, b2, tiles_per_wave=tpw)
            if ref is None: ref = y
            ok = torch.equal(y, ref)
            t = t_us(lambda: ops.gated_mlp_fwd(x, lw, lb, W1, b1, W2, b2, tiles_per_wave=tpw))
            res.append("tpw%d=%.1fus%s" % (tpw, t, "" if ok else "(!=)"))
        except Exception as ex:
            res.append("tpw%d=err" % tpw)
    fl = 6.0 * M * C * HP
    best = min(float(r.split("=")[1].split("us")[0]) for r in res if "us" in r)
    print("M=%d C=%d hid=%d: %s  best %.0f TFLOP/s" % (M, C, hid, " ".join(res), fl / best / 1e6), flush=True)
