"""count aten::copy_/fill_/sum launches: forward with warm weight cache vs forward after a weight-epoch bump (= repack) vs train step."""
import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from torch.profiler import profile, ProfilerActivity
from mp_hsir_amd import ops
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

def run(tag, fn):
    fn(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        fn(); torch.cuda.synchronize()
    rows = {e.key: e for e in prof.key_averages()}
    out = []
    for k in ("aten::copy_", "aten::fill_", "aten::sum", "aten::cat", "aten::add", "aten::add_", "aten::mul", "aten::zero_", "aten::_foreach_copy_", "aten::mm", "aten::addmm_", "aten::bmm"):
        if k in rows:
            out.append("%s %d calls %.2f ms" % (k[6:], rows[k].count, rows[k].self_device_time_total / 1e3))
    tot = sum(e.self_device_time_total for e in prof.key_averages()) / 1e3
    print("%-28s total %.2f ms | " % (tag, tot) + " | ".join(out))

def fwd_warm():
    with torch.no_grad():
        net(x, p)
def fwd_repack():
    ops.bump_weight_epoch()
    with torch.no_grad():
        net(x, p)
def fwd_bwd():
    loss = (net(x, p).clamp(0, 1) - c).abs().mean()
    loss.backward()
run("forward, warm cache", fwd_warm)
run("forward, repack", fwd_repack)
run("fwd+bwd, warm cache", fwd_bwd)
run("train step (eager)", lambda: eng.train_step(x, c, p))
