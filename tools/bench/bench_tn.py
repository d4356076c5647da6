"""gemm_tn forms at the shapes of the training step (bf16): transposed-read kernel (form 1) vs ring form (form 2), the GEMM launch
alone (reduce=False) and with its partial reduction."""
import sys, warnings, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = "cuda"
def t_us(fn, n=12):
    """fn(i) runs on input set i (the sets rotate: together they exceed the 256 MB Infinity Cache, so every call reads CLEAN cold
    data -- a zero_() flush leaves the cache full of dirty lines whose write-back then competes with the reads)"""
    for i in range(2): fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
cases = [("dW1 C128", 131072, 704, 128, 0), ("dW2 C128", 131072, 128, 352, 0), ("dWqkv C128", 131072, 384, 128, 0), ("dWproj C128", 131072, 128, 128, 0),
         ("dM C128 b32", 4096, 128, 128, 32), ("dW1 C64", 131072, 384, 64, 0), ("dWqkv C64", 131072, 192, 64, 0), ("dWproj C64", 131072, 64, 64, 0),
         ("dW1 C256", 8192, 1408, 256, 0), ("dWqkv C256", 8192, 768, 256, 0), ("dW1 C128 r32", 32768, 704, 128, 0), ("dM C128 r32 b32", 1024, 128, 128, 32),
         ("pgLR", 2048, 440, 200, 0)]
for name, M, N1, N2, bt in cases:
    nb = (bt or 1) * M * (N1 + N2) * 2
    K = max(2, int(700e6 // nb) + 1)
    As = [torch.randn(((bt, M, N1) if bt else (M, N1)), device=dev, dtype=torch.bfloat16) for _ in range(K)]
    Bs = [torch.randn(((bt, M, N2) if bt else (M, N2)), device=dev, dtype=torch.bfloat16) for _ in range(K)]
    a, b = As[0], Bs[0]
    out = []
    for form, wgs in ((1, 0), (2, 256), (2, 512)):
        ops.TN_FORM, ops.TN_RING_WGS = form, wgs or 256
        g = t_us(lambda i: ops.gemm_tn(As[i % K], Bs[i % K], reduce=False))
        gr = t_us(lambda i: ops.gemm_tn(As[i % K], Bs[i % K]))
        p = ops.gemm_tn(a, b, reduce=False)
        out.append("form%d%s: %6.1f us (%.2f TB/s) +reduce %6.1f us, partials %5.1f MB" % (form, "/%d" % wgs if wgs else "", g, nb / g / 1e6, gr, p.numel() * 4 / 1e6))
    print("%-16s %6.1f MB | %s" % (name, nb / 1e6, " | ".join(out)), flush=True)
