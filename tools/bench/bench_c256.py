import sys, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
def t_us(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H, C, heads, ln) in [(1, 128, 256, 8, False), (1, 256, 256, 8, True), (32, 16, 256, 8, False)]:
    x = torch.randn(B * H * H, C, device=dev, dtype=dt)
    w = (torch.randn(3 * C, C, device=dev) * C ** -0.5).to(dt)
    w9 = torch.randn(9, 3 * C, device=dev) / 3
    lnp = (torch.ones(C, device=dev), torch.zeros(C, device=dev)) if ln else None
    tiles = (H // 8) * (H // 16)
    print("B=%d %dx%d C=%d ln=%s: default %.1f us" % (B, H, H, C, ln, t_us(lambda: ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, ln=lnp))), flush=True)
    for hg in (1, 2, 4, 8):
        res = []
        for ns in (tiles, tiles // 2, tiles // 4, tiles // 8):
            if ns < 1 or tiles % ns: continue
            try:
                res.append("ns=%d:%.1f" % (ns, t_us(lambda: ops.qkv_dwconv_gram(x, w, w9, B, H, H, C, heads, ln=lnp, nsplit=ns, head_groups=hg))))
            except Exception as ex:
                res.append("ns=%d:err" % ns)
        print("   hg=%d  %s  (WGs = B*ns*hg)" % (hg, "  ".join(res)), flush=True)
