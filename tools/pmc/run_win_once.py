"""one shape of win_attn_fwd, a few launches (for rocprofv3 --pmc): python tools/run_win_once.py C heads H [B]"""
import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
from mp_hsir_amd.net.MP_HSIR import PGSSTB
C, heads, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda"); dt = torch.bfloat16
blk = PGSSTB(C, heads, [64, 64], 8, 4, 0.0, 2.66, 8, 128).to(dev)
pk = blk.packed(dt)
x = torch.randn(B, H, H, C, device=dev, dtype=dt)
for _ in range(5):
    ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"], pk["bproj"], pk["pg"], heads, 4, save=False)
torch.cuda.synchronize()
