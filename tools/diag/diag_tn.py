"""diagnostic (GPU box): every token-reduction GEMM launch (single, batched, grouped) of one eager training step with its problems and
its own HIP-event time, largest first.  python tools/diag_tn.py [form]"""
import sys, os, warnings, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
from mp_hsir_amd.data import SyntheticPatchSource
from mp_hsir_amd import ops, _lib

if len(sys.argv) > 1:
    ops.TN_FORM = int(sys.argv[1])
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
lib = _lib.load()
rec = []
g0, s0, c0 = lib.mphsir_gemm_tn_group, lib.mphsir_gemm_tn, lib.mphsir_conv3x3_wgrad


def timed(desc, fn, *a):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn(*a)
    e.record()
    rec.append((desc, s, e))
    return r


def group(arr, n, *rest):
    probs = tuple((int(arr[k].M), int(arr[k].N1), int(arr[k].N2), int(arr[k].nsplit)) for k in range(n))
    return timed(("group",) + probs, g0, arr, n, *rest)


def single(A, lda, abs_, B, ldb, bbs, Cp, cs, M, N1, N2, nsplit, Bt, *rest):
    return timed(("single", (M, N1, N2, nsplit, Bt)), s0, A, lda, abs_, B, ldb, bbs, Cp, cs, M, N1, N2, nsplit, Bt, *rest)


def conv(dY, lddy, X, ldx, Cp, B, H, W, Cout, Cin, nsplit, *rest):
    return timed(("conv_wgrad", (B * H * W, Cout, 9 * Cin, nsplit)), c0, dY, lddy, X, ldx, Cp, B, H, W, Cout, Cin, nsplit, *rest)


lib.mphsir_gemm_tn_group, lib.mphsir_gemm_tn, lib.mphsir_conv3x3_wgrad = group, single, conv
fwd_bwd()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, s, e in rec:
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += s.elapsed_time(e) * 1e3
tot = sum(v[1] for v in agg.values())
print("gemm_tn form %d: %d launches, %.0f us (event time around each launch, eager)" % (ops.TN_FORM, len(rec), tot))
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    byt = sum(q[0] * (q[1] + q[2]) * 2 * (q[4] if len(q) > 4 else 1) for q in key[1:])
    par = sum(q[1] * q[2] * q[3] * 4 * (q[4] if len(q) > 4 else 1) for q in key[1:])
    print("  x%-3d %8.1f us  avg %6.1f  %.2f TB/s inputs, partials %.1f MB  %s" % (n, t, t / n, byt * n / t / 1e6, par / 1e6, key))
