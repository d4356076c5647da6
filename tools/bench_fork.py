import sys, warnings
sys.path.insert(0, '/root/repo')
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = "cuda"; dt = torch.bfloat16
B, H, W, C, heads = 32, 64, 64, 128, 2
hd = C // heads
gp = torch.randn(B, 1, heads, hd, hd, device=dev); sp = torch.rand(B, 1, 2, C, device=dev) + 1
temp = torch.ones(heads, device=dev); wo = torch.randn(C, C, device=dev) * 0.1; dM = torch.randn(B, C, C, device=dev)
x = torch.randn(B, H, W, 2 * C, device=dev, dtype=dt); w9 = torch.randn(9, 2 * C, device=dev)
def work(fork_on):
    ops.USE_SIDE_STREAM = fork_on
    with ops.side_stream(dM) as f:
        W2, a, b = ops.spectral_fold_bwd(gp, sp, temp, wo, dM, dt, reduce=False)
    y = ops.dwconv3x3(x, w9)
    y2 = ops.dwconv3x3(y, w9)
    f.join(W2, a, b)
    return W2, y2
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for fork_on in (False, True):
    eager = timeit(lambda: work(fork_on))
    g = torch.cuda.CUDAGraph()
    work(fork_on); torch.cuda.synchronize()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        for _ in range(5): work(fork_on)
    graph = timeit(g.replay) / 5
    print("fork=%s: eager %.1f us, graph %.1f us per (fold_bwd + 2 dwconv)" % (fork_on, eager, graph))
