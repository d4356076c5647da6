"""diagnostic (GPU box): the dense 3x3 conv at the shapes of the 512x512 forward and of the training step."""
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


tot = 0.0
for (B, H, Cin, N) in [(1, 512, 128, 32), (1, 256, 128, 256), (1, 128, 256, 512), (1, 512, 64, 64), (1, 256, 128, 128), (1, 512, 64, 32),
                       (1, 256, 128, 64), (1, 512, 32, 64), (32, 64, 64, 32), (32, 32, 128, 256), (32, 16, 256, 512), (32, 64, 32, 64)]:
    x = torch.randn(B, H, H, Cin, device=dev, dtype=dt)
    w = (torch.randn(N, 9 * Cin, device=dev) * (9 * Cin) ** -0.5).to(dt)
    t = t_us(lambda: ops.conv3x3_tok(x, w))
    tot += t
    print("B=%d %dx%d Cin=%d N=%d: %.1f us  %.0f TFLOP/s" % (B, H, H, Cin, N, t, 18.0 * B * H * H * Cin * N / t / 1e6), flush=True)
print("sum %.1f us" % tot)
