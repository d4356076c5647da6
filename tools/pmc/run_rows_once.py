"""one shape of the fused pass A, row-walking form, a few launches (for rocprofv3 --pmc): python tools/run_rows_once.py C heads H [B] [keep]"""
import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
C, heads, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
keep = len(sys.argv) > 5 and sys.argv[5] == "keep"
dev = torch.device("cuda"); dt = torch.bfloat16
x = torch.randn(B * H * H, C, device=dev, dtype=dt)
w = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(dt)
w9 = torch.randn(9, 3 * C, device=dev) / 3
for _ in range(5):
    ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, keep=keep)
torch.cuda.synchronize()
