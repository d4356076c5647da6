import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = "cuda"; dt = torch.bfloat16
B, H, W, C = 32, 64, 64, 384
x = torch.randn(B, H, W, C, device=dev, dtype=dt); dy = torch.randn(B, H, W, C, device=dev, dtype=dt)
w9 = torch.randn(9, C, device=dev)
for _ in range(3):
    ops.dwconv3x3(x, w9)
    ops.dwconv3x3_wgrad(x, dy)
t = torch.randn(B * H * W, 704, device=dev, dtype=dt); w9g = torch.randn(9, 704, device=dev)
for _ in range(3):
    ops.dwconv_gate(t, w9g, B, H, W)
torch.cuda.synchronize()
print("algorithmic MB: dwconv3x3 %.0f (r %.0f + w %.0f), wgrad %.0f, gate r %.0f w %.0f" % (2 * x.numel() * 2 / 1e6, x.numel() * 2 / 1e6, x.numel() * 2 / 1e6, 2 * x.numel() * 2 / 1e6, t.numel() * 2 / 1e6, t.numel() / 1e6))
