"""window attention (forward / backward per training shape, phase stamps of the forward kernel) and -- `bench_win.py pg` -- the local
spectral-prompt gate (pg_gate_fwd / pg_gate_bwd at the window counts of the training step: launch times with rotating inputs and the
phase stamps of workgroup 0 of the backward kernel, mphsir_debug kind 0, s_memtime ticks)."""
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
def t_us_i(fn, n=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def bench_pg():
    import ctypes
    from mp_hsir_amd import _lib
    lib = _lib.load()
    t_us = t_us_i
    for (nW, C, heads, cr) in [(2048, 64, 2, 8), (512, 128, 4, 16), (128, 256, 8, 32), (2048, 128, 2, 8), (4096, 64, 2, 8)]:
        blk = PGSSTB(C, heads, [64, 64], 8, 4, 0.0, 2.66, cr, 128).to(dev)
        pg = blk.packed(torch.bfloat16)["pg"]
        r = pg["linear_down.weight"].shape[0]
        mus = [torch.randn(nW, C, device=dev) for _ in range(4)]
        dgs = [torch.randn(nW, C, device=dev) for _ in range(4)]
        KL, KR = ops.round_up(C + 5 * r + 256, 8), ops.round_up(5 * r + 1 + C, 8)
        L = torch.randn(nW, KL, device=dev).bfloat16(); R = torch.randn(nW, KR, device=dev).bfloat16()
        tf = t_us(lambda i: ops.pg_gate_fwd(mus[i % 4], pg))
        tb = {}
        for fd in (torch.bfloat16, torch.float32):
            tb[fd] = t_us(lambda i: ops.pg_gate_bwd(mus[i % 4], dgs[i % 4], pg, factor_dtype=fd))
        tg = t_us(lambda i: ops.gemm_tn(L, R, reduce=False))
        stamps = torch.zeros(32, dtype=torch.int64, device=dev)
        lib.mphsir_debug(0, ctypes.c_void_p(stamps.data_ptr()))
        ops.pg_gate_bwd(mus[0], dgs[0], pg, factor_dtype=torch.bfloat16); stamps.zero_()
        ops.pg_gate_bwd(mus[0], dgs[0], pg, factor_dtype=torch.bfloat16); torch.cuda.synchronize()
        t = stamps.cpu().tolist()
        lib.mphsir_debug(0, None)
        ph = [t[i + 1] - t[i] if t[i + 1] and t[i] else None for i in range(16)]
        print("nW=%d C=%d r=%d: fwd %.1f us   bwd (+factor GEMM launch) bf16 rows %.1f us, fp32 rows %.1f us   the GEMM alone %.1f us" % (nW, C, r, tf, tb[torch.bfloat16], tb[torch.float32], tg))
        print("   stamps (ticks between marks k, k+1):", ph, "first..last", max(t) - min(x for x in t if x), flush=True)


if "pg" in sys.argv[1:]:
    bench_pg()
    sys.exit(0)
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
