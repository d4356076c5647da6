"""Diagnostic: which gemm_tok / gemm_tn shapes does one training step launch, and how fast is each?
Records every call of one eager step of the bench workload, then times each distinct shape in isolation."""
import collections
import sys
import os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mp_hsir_amd import ops
from mp_hsir_amd.data import SyntheticPatchSource
from mp_hsir_amd.engine import DataParallelEngine
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net

dev = torch.device("cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net = MP_HSIR_Net(compute_dtype=torch.bfloat16, clip_prompt="surrogate").to(dev).train()
eng = DataParallelEngine(net, lr=2e-4)
src = SyntheticPatchSource(31, 64, B, 6, dev, 2024, 0)
for _ in range(2):
    _, x, c, p = src.next(); eng.train_step(x, c, p)
calls = collections.Counter()
orig = ops.gemm_tok
def rec(x, w, bias=None, ln=None, epi=0, res=None, sa=None, gate=None, keep=None, geom=None, out=None):
    calls[(x.shape[0], w.shape[-2], x.shape[1], epi, ln is not None, w.dim() == 3, x.stride(0), (out.stride(0) if out is not None else w.shape[-2]))] += 1
    return orig(x, w, bias=bias, ln=ln, epi=epi, res=res, sa=sa, gate=gate, keep=keep, geom=geom, out=out)
ops.gemm_tok = rec
import mp_hsir_amd.autograd_ops as AG
_, x, c, p = src.next(); eng.train_step(x, c, p)
ops.gemm_tok = orig
torch.cuda.synchronize()
tot = 0.0
rows = []
for (M, N, K, epi, ln, ps, ldx, ldy), n in sorted(calls.items(), key=lambda kv: -kv[0][0] * (kv[0][1] + kv[0][2]) * kv[1]):
    xw = torch.randn((M, ldx), device=dev, dtype=torch.bfloat16)
    xv = xw[:, :K]
    Bt = 32 if ps else 1
    w = torch.randn((Bt, N, K) if ps else (N, K), device=dev, dtype=torch.bfloat16)
    lnp = (torch.ones(K, device=dev), torch.zeros(K, device=dev)) if ln else None
    res = torch.randn((M, N), device=dev, dtype=torch.bfloat16) if epi else None
    sa = torch.randn((M, N), device=dev, dtype=torch.bfloat16) if epi == 2 else None
    gate = torch.randn((M // 64, N), device=dev) if epi == 2 else None
    outw = torch.empty((M, ldy), device=dev, dtype=torch.bfloat16)
    f = lambda: ops.gemm_tok(xv, w, ln=lnp, epi=epi, res=res, sa=sa, gate=gate, geom=(64, 64, 4) if epi == 2 else None, out=outw[:, :N])
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    nbytes = 2.0 * (M * K + M * N * (1 + (epi > 0) + (epi == 2)))
    tot += ms * n
    rows.append((ms * n, "M=%6d N=%4d K=%4d epi=%d ln=%d ps=%d ldx=%4d ldy=%4d  x%2d  %.1f us  %.2f TB/s  %.0f TF" % (M, N, K, epi, ln, ps, ldx, ldy, n, ms * 1e3, nbytes / ms / 1e9, 2.0 * M * N * K / ms / 1e9)))
for t, r in sorted(rows, reverse=True):
    print("%.3f ms  %s" % (t, r))
print("total %.3f ms / step" % tot)
