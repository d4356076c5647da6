"""lab: the ordered-sum launch behind the attention backward of a level-1 block (C = 128, batch 32): its 18 segments together and one by one"""
import sys, os, warnings
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/mp-hsir_amd") else os.environ.get("GRAFT_REPO_ROOT", "."))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
dev = "cuda"
def t_us(fn, n=20):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(n): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
# (n, rows, nsplit, nbatch, transposed)
SEGS = [(16, 128, 2, 1, 0), (16, 16, 2, 1, 0), (128, 128, 2, 1, 0), (16384, 1, 32, 1, 0), (384, 9, 64, 1, 1), (49152, 1, 128, 1, 0), (49152, 1, 128, 1, 0),
        (384, 1, 128, 1, 0), (16384, 1, 128, 1, 0), (128, 1, 128, 1, 0), (256, 1, 2048, 1, 0), (450, 1, 2048, 1, 0)]
K = 4
def make(seg):
    n, rows, nsplit, nb, tr = seg
    if tr:      # tap gradients: partials (nsplit, 9, n) summed into (n, 9)
        return [torch.randn(nsplit, rows, n, device=dev) for _ in range(K)]
    if rows > 1:
        return [torch.randn(nsplit, rows, n, device=dev) for _ in range(K)]
    return [torch.randn(nsplit, n, device=dev) for _ in range(K)]
bufs = [make(s) for s in SEGS]
def run(idx, i):
    with ops.reduce_scope():
        for j in idx:
            n, rows, nsplit, nb, tr = SEGS[j]
            p = bufs[j][i % K]
            if tr:
                out = torch.empty((n, rows), dtype=torch.float32, device=dev)
                ops.reduce_block(p, 0, rows, 0, n, out, transpose=True)
            else:
                ops.reduce_parts(p)
allidx = list(range(len(SEGS)))
mb = sum(s[0] * s[1] * s[2] * s[3] * 4 for s in SEGS) / 1e6
print("all %d segments (%.1f MB): %.1f us" % (len(SEGS), mb, t_us(lambda i: run(allidx, i))))
for j, s in enumerate(SEGS):
    print("  segment %s (%.2f MB): %.1f us" % (s, s[0] * s[1] * s[2] * s[3] * 4 / 1e6, t_us(lambda i: run([j], i))))
print("without the two 2048-split ones: %.1f us" % t_us(lambda i: run(allidx[:-2], i)))
print("only the two 49152 x 128: %.1f us" % t_us(lambda i: run([5, 6], i)))
