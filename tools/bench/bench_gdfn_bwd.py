"""mphsir_gdfn_dw_bwd (gate backward + depthwise backward of the GDFN in one launch) against the two launches it replaces, on the
four GDFN shapes of the natural-scene training step (batch 32).  Usage: python tools/bench/bench_gdfn_bwd.py"""
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


for (B, H, hid) in [(32, 64, 340), (32, 64, 170), (32, 32, 680), (32, 32, 340), (16, 64, 510), (16, 64, 255)]:
    HP = ops.round_up(hid, 32)
    M = B * H * H
    NS = max(1, min(3, int(600e6 // (M * HP * 2 * 8))))
    sets = [dict(t=torch.randn(M, 2 * HP, device=dev, dtype=dt), du=torch.randn(M, HP, device=dev, dtype=dt)) for _ in range(NS)]
    w9 = torch.randn(9, 2 * HP, device=dev) / 3
    k = [0]

    def two():
        s = sets[k[0] % NS]; k[0] += 1
        u, dtdw = ops.dwconv_gate_bwd(s["t"], w9, s["du"], B, H, H)
        with ops.reduce_scope():
            ops.dwconv3x3_bwd(s["t"].reshape(B, H, H, 2 * HP), dtdw.reshape(B, H, H, 2 * HP), w9, col_ranges=[(0, hid), (HP, hid)])

    def fused(nblk):
        def f():
            s = sets[k[0] % NS]; k[0] += 1
            ops.gdfn_dw_bwd(s["t"], w9, s["du"], B, H, H, nblk=nblk)
        return f
    t2 = t_us(two)
    nslab = HP // 16
    tiles = B * (H // 8) * (H // 16)
    res = []
    for wgs in (256, 512, 768, 1024, 2048):
        nb = max(1, min(tiles, wgs // nslab))
        nb = nb // 8 * 8 if nb >= 8 else nb
        if res and res[-1][0] == nb:
            continue
        res.append((nb, t_us(fused(nb))))
    best = min(res, key=lambda r: r[1])
    byt = (2.0 + 1.0 + 1.0 + 2.0) * M * HP * 2
    print("B=%d %dx%d hid=%d (HP %d, %d slabs, %d tiles): two launches %.1f us | fused best nblk=%d %.1f us (%.2f TB/s of 6 HP per token) x%.2f | %s" % (
        B, H, H, hid, HP, nslab, tiles, t2, best[0], best[1], byt / best[1] / 1e6, t2 / best[1], " ".join("nblk%d=%.1f" % r for r in res)), flush=True)
