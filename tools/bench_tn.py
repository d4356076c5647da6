"""gemm_tn variants at the shapes of the training step (bf16)."""
import sys, warnings
sys.path.insert(0, '/root/repo')
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = "cuda"
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
cases = [("dW1 C128", 131072, 704, 128, 0), ("dW2 C128", 131072, 128, 352, 0), ("dWqkv C128", 131072, 384, 128, 0), ("dWproj C128", 131072, 128, 128, 0),
         ("dM C128 b32", 4096, 128, 128, 32), ("dW1 C64", 131072, 384, 64, 0), ("dWqkv C64", 131072, 192, 64, 0), ("dWproj C64", 131072, 64, 64, 0),
         ("dW1 C256", 8192, 1408, 256, 0), ("dWqkv C256", 8192, 768, 256, 0), ("dW1 C128 r32", 32768, 704, 128, 0), ("dM C128 r32 b32", 1024, 128, 128, 32),
         ("pgLR", 2048, 440, 200, 0)]
for name, M, N1, N2, bt in cases:
    a = torch.randn(((bt, M, N1) if bt else (M, N1)), device=dev, dtype=torch.bfloat16)
    b = torch.randn(((bt, M, N2) if bt else (M, N2)), device=dev, dtype=torch.bfloat16)
    nb = (bt or 1) * M * (N1 + N2) * 2
    old = t_us(lambda: ops.gemm_tn(a, b, tile128=False))
    res = []
    for wgs in (0.5, 1.0, 2.0):
        ops.TN_BIG_ROUNDS = wgs
        res.append(t_us(lambda: ops.gemm_tn(a, b, tile128=True)))
    print("%-18s old %6.1f us (%.2f TB/s) | tr kernel @0.5/1/2 rounds: %s us  (best %.2f TB/s)" % (
        name, old, nb / old / 1e6, " ".join("%6.1f" % r for r in res), nb / min(res) / 1e6))
