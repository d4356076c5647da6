"""diagnostic (GPU box): every library launch of one eager 512x512x31 bf16 forward with its shape and its own HIP-event time,
grouped by (op, shape).  python tools/diag_fwd512.py"""
import sys, warnings, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
from mp_hsir_amd import ops

dev = torch.device("cuda")
torch.manual_seed(0)
net = MP_HSIR_Net(compute_dtype=torch.bfloat16, clip_prompt="surrogate").to(dev).eval()
x = torch.rand(1, 31, 512, 512, device=dev)
p = torch.tensor([0], device=dev)
NAMES = ["qkv_dwconv_gram", "win_attn_fwd", "gated_mlp", "gemm_tok", "pg_gate_fwd", "spectral_fold", "conv3x3_tok", "dwconv_gate",
         "dwconv_gram", "reduce_parts", "resize_bilinear", "layernorm_tok"]
rec = []


def wrap(name):
    orig = getattr(ops, name, None)
    if orig is None:
        return
    def f(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(*a, **k)
        e.record()
        shapes = tuple(tuple(t.shape) for t in a[:3] if torch.is_tensor(t))
        extra = tuple((kk, vv) for kk, vv in k.items() if isinstance(vv, (int, bool)))
        rec.append(((name, shapes, extra), s, e))
        return r
    setattr(ops, name, f)


with torch.no_grad():
    for _ in range(2):
        net(x, p)
    for n in NAMES:
        wrap(n)
    net(x, p)
    torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, s, e in rec:
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += s.elapsed_time(e) * 1e3
tot = sum(v[1] for v in agg.values())
print("%d launches, %.0f us (event time around each call, eager)" % (len(rec), tot))
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%7.0f us  %3d x %6.1f  %s" % (t, n, t / n, key))
