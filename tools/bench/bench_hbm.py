"""What the box's HBM actually delivers to simple streaming kernels (torch copy / fill / sum, 1-2 GB buffers): the practical
ceiling the roofline fractions in DESIGN.md should be read against (peak in MI355X_MICROARCH.md: 8 TB/s)."""
import torch
dev = torch.device("cuda")


def t_ms(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for mb in (64, 256, 1024, 2048):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(x)
    tc = t_ms(lambda: y.copy_(x))
    tf = t_ms(lambda: y.zero_())
    ts = t_ms(lambda: x.float().sum()) if mb <= 256 else None
    tr = t_ms(lambda: torch.max(x))
    print("%5d MB: copy %.3f ms = %.2f TB/s (read+write)   fill %.2f TB/s   max-reduce (read only) %.2f TB/s" % (
        mb, tc, 2 * mb / 1024 / 1024 / tc * 1e3 * 1.048576, mb / 1024 / 1024 / tf * 1e3 * 1.048576, mb / 1024 / 1024 / tr * 1e3 * 1.048576), flush=True)
