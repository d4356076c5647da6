import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = "cuda"; dt = torch.bfloat16
M, C, hid = 131072, 128, 340
x, dy = torch.randn(M, C, device=dev, dtype=dt), torch.randn(M, C, device=dev, dtype=dt)
fc1w, fc1b, fc2w = torch.randn(2 * hid, C, device=dev) * C ** -0.5, torch.randn(2 * hid, device=dev) * 0.1, torch.randn(C, hid, device=dev) * hid ** -0.5
lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
W1, b1, W2 = ops.pack_gated_mlp(fc1w, fc1b, fc2w, dt)
W1T, W2T = W1.t().contiguous(), W2.t().contiguous()
for v in (1, 2, 3):
    for _ in range(3):
        ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T, variant=v)
torch.cuda.synchronize()
