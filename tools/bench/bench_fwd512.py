"""diagnostic: replayed-hipGraph time of one 512x512x31 bf16 forward (test.py shape).  (GPU box only)"""
import sys, time, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
from mp_hsir_amd.engine import GraphedForward

dev = torch.device("cuda")
torch.manual_seed(0)
net = MP_HSIR_Net(compute_dtype=torch.bfloat16, clip_prompt="surrogate").to(dev).eval()
x = torch.rand(1, 31, 512, 512, device=dev)
p = torch.tensor([0], device=dev)
with torch.no_grad():
    for _ in range(2):
        net(x, p)
    run = GraphedForward(net, warmup=0)
    run(x, p)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            run(x, p)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 20)
print("512x512x31 forward: %.3f ms" % (best * 1e3))
