"""diagnostic: where does a training step go?  (GPU box only)"""
import sys, time, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
from mp_hsir_amd.data import SyntheticPatchSource
from mp_hsir_amd import ops
import mp_hsir_amd.autograd_ops as AG

dev = torch.device("cuda")
torch.manual_seed(0)
net = MP_HSIR_Net(compute_dtype=torch.bfloat16, clip_prompt="surrogate").to(dev).train()
src = SyntheticPatchSource(31, 64, 32, 6, dev, 2024, 0)
_, x, c, p = src.next()

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3

def fwd_bwd():
    net.zero_grad(set_to_none=True)
    y = net(x, p); (y.clamp(0, 1) - c).abs().mean().backward()
print("fwd+bwd (no optimizer, cached packs): %.1f ms" % timeit(fwd_bwd))
def fwd_bwd_repack():
    ops.bump_weight_epoch(); fwd_bwd()
print("fwd+bwd with repack every step:       %.1f ms" % timeit(fwd_bwd_repack))
def fwd_only():
    with torch.no_grad(): net(x, p)
print("forward only:                         %.1f ms" % timeit(fwd_only))
# prompt modules detached
orig_tvsp, orig_fus = AG.tvsp, AG.prompt_fusion
AG.prompt_fusion = lambda m, a, b: orig_fus(m, a.detach(), b.detach()).detach()
print("fwd+bwd, prompt modules w/o backward: %.1f ms" % timeit(fwd_bwd))
AG.prompt_fusion = orig_fus
# one stage alone
for name in ["encoder_level1", "encoder_level2", "latent", "refinement"]:
    st = getattr(net, name)
    Cc = st.blocks[0].dim
    res = {"encoder_level1": 64, "encoder_level2": 32, "latent": 16, "refinement": 64}[name]
    xs = torch.randn(32, res, res, Cc, device=dev, dtype=torch.bfloat16, requires_grad=True)
    def f():
        st.zero_grad(set_to_none=True)
        y = st(xs); y.float().mean().backward()
    def ff():
        with torch.no_grad(): st(xs)
    print("%-16s depth %d: fwd %.2f ms, fwd+bwd %.2f ms" % (name, len(st.blocks), timeit(ff), timeit(f)))
