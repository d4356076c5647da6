"""gated-MLP parameter gradients: the operand path (gated_mlp_bwd writes h / dpre, gemm_tn reads them) against the recomputing
kernel (gated_mlp_bwd without operands + gated_mlp_wgrad), per training shape; cold inputs by rotating over 6 input sets."""
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops

dev, dt = "cuda", torch.bfloat16


def t_us(fn, n=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


shapes = [(131072, 128, 340), (32768, 128, 340), (131072, 64, 170), (65536, 192, 510), (16384, 192, 510), (65536, 96, 255)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for M, C, hid in shapes:
    NS = 6
    xs = [torch.randn(M, C, device=dev, dtype=dt) for _ in range(NS)]
    dys = [torch.randn(M, C, device=dev, dtype=dt) for _ in range(NS)]
    fc1w, fc1b, fc2w = torch.randn(2 * hid, C, device=dev) * C ** -0.5, torch.randn(2 * hid, device=dev) * 0.1, torch.randn(C, hid, device=dev) * hid ** -0.5
    lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    W1, b1, W2 = ops.pack_gated_mlp(fc1w, fc1b, fc2w, dt)
    W1T, W2T = W1.t().contiguous(), W2.t().contiguous()
    HP = W2.shape[1]

    def old(i):
        x, dy = xs[i % NS], dys[i % NS]
        dx, xn, h, dpre, part = ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T)
        with ops.reduce_scope():
            ops.gemm_tn_blocks(dy, h, [(0, C)], ncols=hid, colsum=True)
            ops.gemm_tn_blocks(dpre, xn, [(0, hid), (HP, hid)], colsum=True)

    def k1_old(i):
        x, dy = xs[i % NS], dys[i % NS]
        ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T)

    def k1_new(i):
        x, dy = xs[i % NS], dys[i % NS]
        ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T, operands=False)

    xn0 = ops.gated_mlp_bwd(xs[0], dys[0], dys[0], lnw, lnb, W1, b1, W1T, W2T, operands=False)[1]
    xns = [xn0.clone() for _ in range(NS)]
    out = ["M=%6d C=%3d  old k1 %6.1f  new k1 %6.1f  old total %6.1f us |" % (M, C, t_us(k1_old), t_us(k1_new), t_us(old))]
    for nch in (1, 2):
        if not ops._lib.load().mphsir_gated_mlp_wgrad_fits(C, nch, 1):
            continue
        S = (HP // 32 + nch - 1) // nch
        slots = 512 // nch
        cands = sorted({max(8, (slots * f // 4) // S // 8 * 8) for f in (1, 2, 3, 4, 6, 8)})
        for R in cands:
            def k2(i):
                with ops.reduce_scope():
                    ops.gated_mlp_wgrad(xns[i % NS], dys[i % NS], W1, b1, W2T, hid, nch=nch, ranges=R)
            us = t_us(k2)
            out.append("nch%d R%3d(%4d wg) %6.1f" % (nch, R, R * S, us))
    print(" ".join(out), flush=True)
