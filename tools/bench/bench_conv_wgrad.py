"""diagnostic (GPU box): dense-conv weight gradient, gathered inside the token-reduction GEMM vs im2col + gemm_tn."""
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


for (B, H, Cin, N) in [(32, 64, 32, 64), (32, 64, 64, 32), (32, 32, 128, 64), (32, 32, 128, 256), (32, 16, 256, 512), (32, 64, 64, 64), (32, 32, 128, 128), (32, 64, 64, 32)]:
    x = torch.randn(B, H, H, Cin, device=dev, dtype=dt)
    dy = torch.randn(B * H * H, N, device=dev, dtype=dt)
    a = t_us(lambda: ops.conv3x3_wgrad(dy, x))
    b = t_us(lambda: ops.gemm_tn(dy, ops.im2col3x3(x), immediate=True))
    print("B=%d %dx%d Cin=%d N=%d: gathered %.1f us, im2col + gemm_tn %.1f us" % (B, H, H, Cin, N, a, b), flush=True)
