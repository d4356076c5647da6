"""gated-MLP kernels per training shape (GPU box only):
    python tools/bench/bench_mlp.py fwd            forward: four-wave (tiles_per_wave 1, 2) against eight-wave (3, 4) workgroups
    python tools/bench/bench_mlp.py bwd            data-gradient kernel forms (variant 0..4)
    python tools/bench/bench_mlp.py wgrad [n]      parameter gradients: the operand path (gated_mlp_bwd writes h / dpre, gemm_tn reads them)
                                                   against the recomputing kernel (gated_mlp_bwd without operands + gated_mlp_wgrad),
                                                   swept over token ranges and chunks per workgroup; cold inputs by rotation"""
import os
import sys
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops


def bench_fwd():
    dev = torch.device("cuda"); dt = torch.bfloat16


    def t_us(fn, n=20):
        for _ in range(3): fn()
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3


    for (M, C, hid) in [(131072, 64, 170), (131072, 128, 340), (32768, 128, 340), (8192, 256, 680), (262144, 128, 340), (65536, 96, 255), (65536, 192, 510)]:
        HP = ops.round_up(hid, 32)
        x = torch.randn(M, C, device=dev, dtype=dt)
        W1 = (torch.randn(2 * HP, C, device=dev) * C ** -0.5).to(dt)
        W2 = (torch.randn(C, HP, device=dev) * HP ** -0.5).to(dt)
        b1, b2 = torch.zeros(2 * HP, device=dev), torch.zeros(C, device=dev)
        lw, lb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        ref = None
        res = []
        for tpw in (0, 1, 2, 3, 4):
            try:
                y = ops.gated_mlp_fwd(x, lw, lb, W1, b1, W2, b2, tiles_per_wave=tpw)
                if ref is None: ref = y
                ok = torch.equal(y, ref)
                t = t_us(lambda: ops.gated_mlp_fwd(x, lw, lb, W1, b1, W2, b2, tiles_per_wave=tpw))
                res.append("tpw%d=%.1fus%s" % (tpw, t, "" if ok else "(!=)"))
            except Exception as ex:
                res.append("tpw%d=err" % tpw)
        fl = 6.0 * M * C * HP
        best = min(float(r.split("=")[1].split("us")[0]) for r in res if "us" in r)
        print("M=%d C=%d hid=%d: %s  best %.0f TFLOP/s" % (M, C, hid, " ".join(res), fl / best / 1e6), flush=True)


def bench_bwd():
    dev = "cuda"; dt = torch.bfloat16
    def t_us(fn, n=10):
        for _ in range(2): fn()
        torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    for M, C, hid in [(131072, 128, 340), (32768, 128, 340), (131072, 64, 170), (8192, 256, 680), (65536, 192, 510), (16384, 192, 510), (65536, 96, 255), (4096, 384, 1021)]:
        x, dy = torch.randn(M, C, device=dev, dtype=dt), torch.randn(M, C, device=dev, dtype=dt)
        fc1w, fc1b, fc2w = torch.randn(2 * hid, C, device=dev) * C ** -0.5, torch.randn(2 * hid, device=dev) * 0.1, torch.randn(C, hid, device=dev) * hid ** -0.5
        lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        W1, b1, W2 = ops.pack_gated_mlp(fc1w, fc1b, fc2w, dt)
        W1T, W2T = W1.t().contiguous(), W2.t().contiguous()
        HP = W2.shape[1]
        fl = 12.0 * M * C * HP
        res = []
        for v in (0, 1, 2, 3, 4):
            try:
                us = t_us(lambda: ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T, variant=v))
                res.append("v%d %7.1f us %6.1f TF/s" % (v, us, fl / us / 1e6))
            except Exception as e:
                res.append("v%d failed" % v)
        print("M=%d C=%d: " % (M, C) + " | ".join(res))


def bench_wgrad(argv):
    sys.argv = [sys.argv[0]] + argv

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


if __name__ == "__main__":
    kind = sys.argv[1] if len(sys.argv) > 1 else ""
    if kind == "fwd":
        bench_fwd()
    elif kind == "bwd":
        bench_bwd()
    elif kind == "wgrad":
        bench_wgrad(sys.argv[2:])
    else:
        raise SystemExit(__doc__)
