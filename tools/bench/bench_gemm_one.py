"""Diagnostic: a few big gemm_tok shapes in isolation (MPHSIR_GEMM_NW selects the column tiles per workgroup)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mp_hsir_amd import ops
dev = "cuda"
for (M, N, K, epi) in [(131072, 384, 128, 0), (131072, 128, 384, 0), (131072, 128, 384, 1), (131072, 192, 64, 0), (131072, 256, 256, 0), (131072, 704, 128, 0)]:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    res = torch.randn(M, N, device=dev, dtype=torch.bfloat16) if epi else None
    f = lambda: ops.gemm_tok(x, w, epi=epi, res=res)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    print("M=%d N=%d K=%d epi=%d: %.1f us  %.2f TB/s" % (M, N, K, epi, ms * 1e3, 2.0 * M * (K + N * (1 + (epi > 0))) / ms / 1e9), flush=True)
