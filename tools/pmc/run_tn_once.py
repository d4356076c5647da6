"""one token-reduction GEMM shape a few times (for tools/pmc_kernel.sh): python tools/run_tn_once.py M N1 N2 form [ring_wgs]"""
import sys, os, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
M, N1, N2, form = (int(v) for v in sys.argv[1:5])
ops.TN_FORM = form
if len(sys.argv) > 5:
    ops.TN_RING_WGS = int(sys.argv[5])
K = 6
As = [torch.randn((M, N1), device="cuda", dtype=torch.bfloat16) for _ in range(K)]
Bs = [torch.randn((M, N2), device="cuda", dtype=torch.bfloat16) for _ in range(K)]
for i in range(K):
    ops.gemm_tn(As[i], Bs[i], reduce=False)
torch.cuda.synchronize()
