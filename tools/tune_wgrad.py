import sys, warnings
sys.path.insert(0, '/root/repo'); warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = torch.device("cuda"); dt = torch.bfloat16
def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (B, H, C) in [(32, 64, 384), (32, 64, 704), (32, 32, 384), (32, 16, 768)]:
    x = torch.randn(B, H, H, C, device=dev, dtype=dt); dy = torch.randn_like(x)
    for nblk in (256, 512, 1024, 2048, 4096, 8192):
        us = t_us(lambda: ops.dwconv3x3_wgrad(x, dy, nblk=nblk))
        print("wgrad B=%d H=%d C=%d nblk=%5d: %8.1f us  %.2f TB/s" % (B, H, C, nblk, us, 2 * x.numel() * 2 / us / 1e6))
    a = torch.randn(B * H * H, 704, device=dev, dtype=dt); b = torch.randn(B * H * H, 128, device=dev, dtype=dt)
    for ns in (8, 16, 32, 64):
        us = t_us(lambda: ops.gemm_tn(a, b, nsplit=ns))
        print("gemm_tn M=%d 704x128 nsplit=%3d: %8.1f us  %.2f TB/s" % (a.shape[0], ns, us, (a.numel() + b.numel()) * 2 / us / 1e6))
