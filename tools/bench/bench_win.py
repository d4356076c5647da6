import sys, warnings
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
for (B, H, C, heads, cr) in [(32, 64, 128, 2, 8), (32, 64, 64, 2, 8), (32, 32, 128, 4, 16), (32, 16, 256, 8, 32)]:
    blk = PGSSTB(C, heads, [64, 64], 8, 4, 0.0, 2.66, cr, 128).to(dev)
    pk = blk.packed(dt)
    x = torch.randn(B, H, H, C, device=dev, dtype=dt)
    for shift in (0, 4):
        f = lambda: ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"], pk["bproj"], pk["pg"], heads, shift, save=True)
        print("C=%d shift=%d win_attn_fwd %.1f us" % (C, shift, t_us(f)))

    # backward core (set MPHSIR_WINB_XL=0/1 to force the X-tile placement)
    M = B * H * H
    dsa = torch.randn(B, H, H, C, device=dev, dtype=dt)
    dmu = torch.randn(M // 64, C, device=dev)
    fb = lambda: ops.win_attn_bwd(x, dsa, dmu, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wprojT"], heads, 4)
    print("C=%d win_attn_bwd %.1f us" % (C, t_us(fb)))

# per-phase shader-clock stamps of workgroup 0 of the forward kernel (mphsir_debug)
import ctypes
from mp_hsir_amd import _lib
lib = _lib.load()
names = ["LN -> LDS", "qkv (head 0)", "QK^T + softmax", "PV", "O store", "proj MFMAs", "heads 1.., barrier", "out stage + store + mean"]
for (B, H, C, heads, cr) in [(32, 64, 64, 2, 8), (32, 64, 128, 2, 8), (32, 32, 128, 4, 16), (32, 16, 256, 8, 32)]:
    blk = PGSSTB(C, heads, [64, 64], 8, 4, 0.0, 2.66, cr, 128).to(dev)
    pk = blk.packed(dt)
    x = torch.randn(B, H, H, C, device=dev, dtype=dt)
    stamps = torch.zeros(16, dtype=torch.int64, device=dev)
    lib.mphsir_debug(1, ctypes.c_void_p(stamps.data_ptr()))
    f = lambda: ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"], pk["bproj"], pk["pg"], heads, 4, save=True)
    f(); f(); stamps.zero_(); f(); torch.cuda.synchronize()
    t = stamps.cpu().tolist()
    print("C=%d heads=%d phases (shader clocks):" % (C, heads), {n: t[i + 1] - t[i] for i, n in enumerate(names[:7])}, "workgroup 0 total:", t[7] - t[0], flush=True)
    lib.mphsir_debug(1, None)
