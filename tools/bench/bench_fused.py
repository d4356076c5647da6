"""Fused pass A (qkv_dwconv_gram) against the two-kernel path (gemm_tok -> dwconv_gram) on the shapes of the 512x512x31
forward and of the batch-16 64x64 forward.  MPHSIR_FUSED_OCC=1 forces one workgroup per CU."""
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


CASES = [(1, 512, 128, 2, False), (1, 512, 64, 2, False), (1, 256, 128, 4, False), (1, 128, 256, 8, False), (1, 512, 128, 4, True), (1, 256, 256, 8, True),
         (16, 64, 64, 2, False), (16, 32, 128, 4, False), (16, 16, 256, 8, False), (1, 256, 96, 2, False), (1, 128, 192, 4, False),
         (1, 64, 384, 8, False)]
for (B, H, C, heads, ln) in CASES:
    M = B * H * H
    x = torch.randn(M, C, device=dev, dtype=dt)
    w = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(dt)
    w9 = torch.randn(9, 3 * C, device=dev) / 3
    lnp = (torch.ones(C, device=dev), torch.zeros(C, device=dev)) if ln else None
    fused = lambda: ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, ln=lnp)

    def two():
        t = ops.gemm_tok(x, w, ln=lnp)
        return ops.dwconv_gram(t[:, :C], t[:, C:2 * C], t[:, 2 * C:], w9[:, :C], w9[:, C:2 * C], w9[:, 2 * C:], 3 * C, B, H, H, C, heads)
    tf, t2 = t_us(fused), t_us(two)
    tf1 = t_us(lambda: ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, ln=lnp, head_groups=1))
    tg = t_us(lambda: ops.gemm_tok(x, w, ln=lnp))
    byt = 2.0 * M * C * 2
    print("B=%d %dx%d C=%d heads=%d ln=%d: fused %.1f us (%.2f TB/s of x+v, %.0f TFLOP/s useful)   [1 head group: %.1f us]   two-kernel %.1f us (gemm_tok %.1f)  x%.2f" % (
        B, H, H, C, heads, ln, tf, byt / tf / 1e6, 6.0 * M * C * C / tf / 1e6, tf1, t2, tg, t2 / tf), flush=True)

# per-phase shader-clock stamps (100 MHz) of workgroup 0, first tile, first head (mphsir_debug)
import ctypes
from mp_hsir_amd import _lib
lib = _lib.load()
names = ["x load+LN", "W/taps -> LDS", "MFMA", "t -> LDS + barrier", "depthwise", "barrier", "Gram+store"]
for (B, H, C, heads, ln) in [(1, 512, 64, 2, False), (1, 256, 128, 4, False), (1, 128, 256, 8, False)]:
    M = B * H * H
    x = torch.randn(M, C, device=dev, dtype=dt)
    w = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(dt)
    w9 = torch.randn(9, 3 * C, device=dev) / 3
    stamps = torch.zeros(16, dtype=torch.int64, device=dev)
    lib.mphsir_debug(2, ctypes.c_void_p(stamps.data_ptr()))
    for hg in (1, None):
        f = lambda: ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, head_groups=hg)
        f(); f(); stamps.zero_(); f(); torch.cuda.synchronize()
        t = stamps.cpu().tolist()
        print("C=%d hg=%s phases (10 ns ticks):" % (C, hg), {n: t[i + 1] - t[i] for i, n in enumerate(names)}, "kernel of wg0:", t[8] - t[0], flush=True)
    lib.mphsir_debug(2, None)
