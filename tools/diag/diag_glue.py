"""where the non-library time of a training step goes: torch.profiler table of GPU kernels grouped by the torch op that launched them."""
import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from torch.profiler import profile, ProfilerActivity
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
from mp_hsir_amd.engine import DataParallelEngine
from mp_hsir_amd.data import SyntheticPatchSource

dev = torch.device("cuda")
torch.manual_seed(0)
T = 6
net = MP_HSIR_Net(31, 31, 64, task_classes=T, clip_prompt=torch.randn(T, 512)).to(dev).set_compute_dtype(torch.bfloat16)
eng = DataParallelEngine(net, lr=2e-4)
src = SyntheticPatchSource(31, 64, 32, T, dev, 1)
_, x, c, p = src.next()
batch = (x, c, p)
for _ in range(3):
    eng.train_step(*batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(3):
        eng.train_step(*batch)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=70))
