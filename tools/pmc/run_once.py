"""ONE kernel family, one shape, a few launches -- the driver `tools/pmc_kernel.sh` profiles (rocprofv3 --pmc):
    python tools/pmc/run_once.py mlp      C hid M          gated_mlp forward + backward (default forms)
    python tools/pmc/run_once.py wgrad    C hid M [nch]    gated_mlp_wgrad (parameter gradients by recomputation)
    python tools/pmc/run_once.py rows     C heads H [B] [keep]   fused pass A, row-walking form
    python tools/pmc/run_once.py tn       M N1 N2 form     token-reduction GEMM (1 transposed-read kernel, 2 ring form)
    python tools/pmc/run_once.py win      C heads H [B]    win_attn forward
    python tools/pmc/run_once.py win_bwd  C heads H [B]    win_attn backward
    python tools/pmc/run_once.py dw                        depthwise 3x3 / wgrad / gate at the widest shapes"""
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops

dev, dt = torch.device("cuda"), torch.bfloat16
kind, av = sys.argv[1], sys.argv[2:]


def mlp_weights(C, hid):
    fc1w, fc1b, fc2w = torch.randn(2 * hid, C, device=dev) * C ** -0.5, torch.randn(2 * hid, device=dev) * 0.1, torch.randn(C, hid, device=dev) * hid ** -0.5
    W1, b1, W2 = ops.pack_gated_mlp(fc1w, fc1b, fc2w, dt)
    return W1, b1, W2, W1.t().contiguous(), W2.t().contiguous(), torch.ones(C, device=dev), torch.zeros(C, device=dev)


def block(C, heads):
    from mp_hsir_amd.net.MP_HSIR import PGSSTB
    return PGSSTB(C, heads, [64, 64], 8, 4, 0.0, 2.66, 8, 128).to(dev).packed(dt)


if kind in ("mlp", "wgrad"):
    C, hid, M = int(av[0]), int(av[1]), int(av[2])
    W1, b1, W2, W1T, W2T, lnw, lnb = mlp_weights(C, hid)
    x, dy = torch.randn(M, C, device=dev, dtype=dt), torch.randn(M, C, device=dev, dtype=dt)
    for _ in range(4):
        if kind == "mlp":
            ops.gated_mlp_fwd(x, lnw, lnb, W1, b1, W2, torch.zeros(C, device=dev))
            ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T)
        else:
            xn = ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T, operands=False)[1]
            ops.gated_mlp_wgrad(xn, dy, W1, b1, W2T, hid, nch=int(av[3]) if len(av) > 3 else None)
elif kind == "rows":
    C, heads, H = int(av[0]), int(av[1]), int(av[2])
    B, keep = (int(av[3]) if len(av) > 3 else 1), (len(av) > 4 and av[4] == "keep")
    x = torch.randn(B * H * H, C, device=dev, dtype=dt)
    w, w9 = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(dt), torch.randn(9, 3 * C, device=dev) / 3
    for _ in range(5):
        ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, keep=keep)
elif kind == "tn":
    M, N1, N2, ops.TN_FORM = (int(v) for v in av[:4])
    As = [torch.randn((M, N1), device=dev, dtype=dt) for _ in range(6)]
    Bs = [torch.randn((M, N2), device=dev, dtype=dt) for _ in range(6)]
    for a, b in zip(As, Bs):
        ops.gemm_tn(a, b, reduce=False)
elif kind in ("win", "win_bwd"):
    C, heads, H = int(av[0]), int(av[1]), int(av[2])
    B = int(av[3]) if len(av) > 3 else 1
    pk = block(C, heads)
    x, dsa = torch.randn(B, H, H, C, device=dev, dtype=dt), torch.randn(B, H, H, C, device=dev, dtype=dt)
    dmu = torch.randn(B * H * H // 64, C, device=dev)
    for _ in range(5):
        if kind == "win":
            ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"], pk["bproj"], pk["pg"], heads, 4, save=False)
        else:
            ops.win_attn_bwd(x, dsa, dmu, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wprojT"], heads, 4)
elif kind == "dw":
    B, H, W, C = 32, 64, 64, 384
    x, dy, w9 = torch.randn(B, H, W, C, device=dev, dtype=dt), torch.randn(B, H, W, C, device=dev, dtype=dt), torch.randn(9, C, device=dev)
    t, w9g = torch.randn(B * H * W, 704, device=dev, dtype=dt), torch.randn(9, 704, device=dev)
    for _ in range(3):
        ops.dwconv3x3(x, w9)
        ops.dwconv3x3_wgrad(x, dy)
        ops.dwconv_gate(t, w9g, B, H, W)
else:
    raise SystemExit(__doc__)
torch.cuda.synchronize()
