"""Fused pass A: row-walking form (spectral_rows.hip) against the tile form (spectral_fused.hip) on the shapes of the
512x512x31 forward, the batch-16/32 64x64 steps and the RS widths.  Usage: python tools/bench_rows.py [keep]"""
import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"


def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


CASES = [(1, 512, 64, 2, False), (1, 512, 128, 2, False), (1, 512, 128, 4, False), (1, 256, 128, 4, False), (32, 64, 64, 2, False), (32, 64, 128, 2, False),
         (32, 32, 128, 4, False), (16, 64, 96, 2, False), (16, 32, 192, 4, False), (1, 512, 96, 2, False), (1, 256, 192, 4, False)]
for (B, H, C, heads, ln) in CASES:
    M = B * H * H
    x = torch.randn(M, C, device=dev, dtype=dt)
    w = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(dt)
    w9 = torch.randn(9, 3 * C, device=dev) / 3
    lnp = (torch.ones(C, device=dev), torch.zeros(C, device=dev)) if ln else None
    tile = t_us(lambda: ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, ln=lnp, keep=keep, row_segments=0))
    res = []
    s = 1
    while H % s == 0 and H // s >= 4 and s <= 64:
        res.append((s, t_us(lambda: ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, ln=lnp, keep=keep, row_segments=s))))
        s *= 2
    best = min(res, key=lambda r: r[1])
    auto = ops.choose_row_segments(B, H, H, C, heads)
    byt = (7.0 if keep else 2.0) * M * C * 2
    print("B=%d %dx%d C=%d heads=%d ln=%d keep=%d: tile %.1f us | rows best s=%d %.1f us (%.2f TB/s algorithmic, %.0f TFLOP/s useful; auto s=%d) x%.2f | %s" % (
        B, H, H, C, heads, ln, keep, tile, best[0], best[1], byt / best[1] / 1e6, M * (6.0 * C * C + 54.0 * C + 2.0 * C * C / heads) / best[1] / 1e6, auto,
        tile / best[1], " ".join("s%d=%.1f" % r for r in res)), flush=True)

# per-phase shader-clock stamps (100 MHz ticks) of workgroup 0 / wave 0 at walk steps 9 (plain) and 8 (with the edge block)
import ctypes
from mp_hsir_amd import _lib
lib = _lib.load()
names = ["DMA issue + LDS reads + MFMA issue + Gram + stores", "depthwise taps", "row image writes", "edge block + vmcnt", "barrier"]
for (B, H, C, heads, ln) in [(1, 512, 64, 2, False), (1, 512, 128, 2, False), (1, 512, 128, 4, False), (1, 512, 96, 2, False)]:
    M = B * H * H
    x = torch.randn(M, C, device=dev, dtype=dt)
    w = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(dt)
    w9 = torch.randn(9, 3 * C, device=dev) / 3
    lnp = (torch.ones(C, device=dev), torch.zeros(C, device=dev)) if ln else None
    stamps = torch.zeros(64, dtype=torch.int64, device=dev)
    lib.mphsir_debug(2, ctypes.c_void_p(stamps.data_ptr()))
    f = lambda: ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, ln=lnp, keep=keep)
    f(); f(); stamps.zero_(); f(); torch.cuda.synchronize()
    t = stamps.cpu().tolist()
    print("C=%d heads=%d step 9 (shader clocks):" % (C, heads), {n: t[k + 1] - t[k] for k, n in enumerate(names)}, "whole walk of workgroup 0: %d" % (t[9] - t[0]), flush=True)
    lib.mphsir_debug(2, None)
