import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mp_hsir_amd import ops
dev = "cuda"; dt = torch.bfloat16
for (M, C, hid) in [(131072, 128, 340), (32768, 128, 340), (8192, 256, 680), (131072, 64, 170)]:
    g = torch.Generator().manual_seed(1)
    fc1w, fc1b, fc2w = torch.randn(2 * hid, C, generator=g) * C ** -0.5, torch.randn(2 * hid, generator=g) * 0.1, torch.randn(C, hid, generator=g) * hid ** -0.5
    W1, b1, W2 = ops.pack_gated_mlp(fc1w.to(dev), fc1b.to(dev), fc2w.to(dev), dt)
    x, dy = torch.randn(M, C, device=dev, dtype=dt), torch.randn(M, C, device=dev, dtype=dt)
    lnw, lnb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    W1T, W2T = W1.t().contiguous(), W2.t().contiguous()
    f = lambda: ops.gated_mlp_bwd(x, dy, dy, lnw, lnb, W1, b1, W1T, W2T)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print("M=%d C=%d: %.1f us" % (M, C, e0.elapsed_time(e1) / 20 * 1e3), flush=True)
