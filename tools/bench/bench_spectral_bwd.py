"""mphsir_spectral_dqkv_bwd (dv, [dq | dk] and the depthwise backward in one launch) against the three launches it replaces, on the
shapes of the natural-scene training step (batch 32) and of the remote-sensing one (batch 16), swept over tile ranges.
Usage: python tools/bench/bench_spectral_bwd.py"""
import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops, _lib
dev = torch.device("cuda"); dt = torch.bfloat16
lib = _lib.load()


def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


CASES = [(32, 64, 128, 2), (32, 64, 64, 2), (32, 32, 128, 4), (32, 16, 256, 8), (16, 64, 192, 2), (16, 64, 96, 2), (16, 32, 192, 4), (16, 16, 384, 8)]
for (B, H, C, heads) in CASES:
    M = B * H * H
    # rotating operand sets (> the 256 MB Infinity Cache at the big shapes): what the kernels see inside a step
    NS = max(1, min(4, int(600e6 // (M * C * 2 * 9))))
    sets = []
    for i in range(NS):
        sets.append(dict(qk=torch.randn(M, 2 * C, device=dev, dtype=dt), d_out=torch.randn(M, C, device=dev, dtype=dt), t=torch.randn(M, 3 * C, device=dev, dtype=dt)))
    W2 = (torch.randn(B, 2 * C, 2 * C, device=dev) * (2 * C) ** -0.5).to(dt)
    MbT = (torch.randn(B, C, C, device=dev) * C ** -0.5).to(dt)
    w9 = torch.randn(9, 3 * C, device=dev) / 3
    dall = torch.empty(M, 3 * C, device=dev, dtype=dt)
    k = [0]

    def three():
        s = sets[k[0] % NS]; k[0] += 1
        ops.gemm_tok(s["d_out"], MbT, out=dall[:, 2 * C:])
        ops.gemm_tok(s["qk"], W2, out=dall[:, :2 * C])
        with ops.reduce_scope():
            ops.dwconv3x3_bwd(s["t"].reshape(B, H, H, 3 * C), dall.reshape(B, H, H, 3 * C), w9, col_ranges=[(0, 3 * C)])

    def fused(nblk):
        def f():
            s = sets[k[0] % NS]; k[0] += 1
            with ops.reduce_scope():
                ops.spectral_dqkv_bwd(s["qk"], s["d_out"], s["t"], W2, MbT, w9, B, H, H, C, heads, nblk=nblk)
        return f
    t3 = t_us(three)
    nslab = lib.mphsir_spectral_dqkv_bwd_slabs(C, heads)
    tiles = B * (H // 8) * (H // 16)
    res = []
    for wgs in (256, 384, 512, 768, 1024, 2048):
        nb = max(1, min(tiles, wgs // nslab))
        nb = nb // 8 * 8 if nb >= 8 else nb
        if res and res[-1][0] == nb:
            continue
        res.append((nb, t_us(fused(nb))))
    best = min(res, key=lambda r: r[1])
    byt = 10.0 * M * C * 2
    print("B=%d %dx%d C=%d heads=%d (%d slabs, %d tiles): three launches %.1f us | fused best nblk=%d %.1f us (%.2f TB/s of 10 C per token) x%.2f | %s" % (
        B, H, H, C, heads, nslab, tiles, t3, best[0], best[1], byt / best[1] / 1e6, t3 / best[1], " ".join("nblk%d=%.1f" % r for r in res)), flush=True)
