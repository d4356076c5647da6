"""diagnostic (GPU box): every gemm_tok / gemm_tn / reduce launch of one eager training step with its shape and its own HIP-event time,
grouped by shape -- which shapes the per-kernel totals of bench.py are made of.  python tools/diag_shapes.py [gemm_tok|gemm_tn|...]"""
import sys, warnings, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
from mp_hsir_amd.data import SyntheticPatchSource
from mp_hsir_amd import ops

which = sys.argv[1] if len(sys.argv) > 1 else "gemm_tok"
dev = torch.device("cuda")
torch.manual_seed(0)
net = MP_HSIR_Net(compute_dtype=torch.bfloat16, clip_prompt="surrogate").to(dev).train()
src = SyntheticPatchSource(31, 64, 32, 6, dev, 2024, 0)
_, x, c, p = src.next()


def fwd_bwd():
    net.zero_grad(set_to_none=True)
    (net(x, p).clamp(0, 1) - c).abs().mean().backward()


for _ in range(3):
    fwd_bwd()
torch.cuda.synchronize()
rec = []
orig = getattr(ops, which)


def wrapped(*a, **k):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = orig(*a, **k)
    e.record()
    if which == "gemm_tok":
        xx, w = a[0], a[1]
        key = (xx.shape[0], w.shape[-2], xx.shape[1], k.get("epi", 0), k.get("ln") is not None, w.dim() == 3)
    elif which == "gemm_tn":
        aa, bb = a[0], a[1]
        key = (tuple(aa.shape), tuple(bb.shape), k.get("immediate", False))
    else:
        key = tuple(tuple(t.shape) for t in a if torch.is_tensor(t))
    rec.append((key, s, e))
    return r


setattr(ops, which, wrapped)
import mp_hsir_amd.autograd_ops as AG
fwd_bwd()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, s, e in rec:
    t = s.elapsed_time(e) * 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += t
tot = sum(v[1] for v in agg.values())
print("%s: %d launches, %.0f us (event time around each call, eager)" % (which, len(rec), tot))
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    extra = ""
    if which == "gemm_tok":
        M, N, K = key[:3]
        byt = (M * K + M * N) * 2
        extra = "  %.2f TB/s (x + y only)" % (byt * n / t / 1e6)
    print("  %-60s x%-3d %8.1f us  avg %6.1f%s" % (key, n, t, t / n, extra))
