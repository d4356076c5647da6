import sys, warnings
sys.path.insert(0, '/root/repo')
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
M, N1, N2 = 131072, 704, 128
a = torch.randn((M, N1), device="cuda", dtype=torch.bfloat16)
b = torch.randn((M, N2), device="cuda", dtype=torch.bfloat16)
ops.TN_BIG_ROUNDS = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
for _ in range(5):
    ops.gemm_tn(a, b, tile128=True)
    ops.gemm_tn(a, b, tile128=False)
torch.cuda.synchronize()
