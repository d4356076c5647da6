"""one shape of gated_mlp forward and backward (default forms), a few launches (for rocprofv3 --pmc): python tools/run_mlp_once.py C hid M"""
import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
C, hid, M = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = "cuda"; dt = torch.bfloat16
x, dy = torch.randn(M, C, device=dev, dtype=dt), torch.randn(M, C, device=dev, dtype=dt)
fc1w, fc1b, fc2w = torch.randn(2 * hid, C, device=dev) * C ** -0.5, torch.randn(2 * hid, device=dev) * 0.1, torch.randn(C, hid, device=dev) * hid ** -0.5
lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
W1, b1, W2 = ops.pack_gated_mlp(fc1w, fc1b, fc2w, dt)
W1T, W2T = W1.t().contiguous(), W2.t().contiguous()
b2 = torch.zeros(C, device=dev)
for _ in range(4):
    ops.gated_mlp_fwd(x, lnw, lnb, W1, b1, W2, b2)
    ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T)
torch.cuda.synchronize()
