"""spectral_fold_bwd per training shape in isolation (HIP events, rotating operand sets): the four shapes of a natural-scene step.
    python tools/bench/bench_fold_bwd.py            (MPHSIR_LIB_AB=ab/<variant>.so selects a variant build)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from mp_hsir_amd import ops

dev, dt = "cuda", torch.bfloat16
g = torch.Generator(device=dev).manual_seed(1)


def rnd(*s, dtype=torch.float32, scale=1.0):
    return (torch.randn(*s, generator=g, device=dev) * scale).to(dtype)


def run(B, C, heads, N, splits, blocks, sets=4, iters=40):
    hd = C // heads
    ins = []
    for _ in range(sets):
        gp, sp = rnd(B, 1, heads, hd, hd), rnd(B, 1, 2, C).abs() + 0.5
        temp, wo = (1 + 0.3 * rnd(heads)).contiguous(), rnd(C, C, scale=C ** -0.5)
        if splits == 0:
            ins.append((gp, sp, temp, wo, None, rnd(B * N, C, dtype=dt), rnd(B * N, C, dtype=dt)))
        else:
            ins.append((gp, sp, temp, wo, rnd(B, splits, C, C), None, None))

    def once(i):
        gp, sp, temp, wo, dm, do, v = ins[i % sets]
        return ops.spectral_fold_bwd(gp, sp, temp, wo, dm, dt, reduce=False, d_out=do, v=v, w2_blocks=blocks)
    for i in range(5):
        once(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()          # replayed from a captured graph: the eager call path costs ~13 us per launch, as much as the kernel
    with torch.cuda.graph(gr):
        keep = [once(i) for i in range(iters)]
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


# (B, C, heads, tokens per sample, dM split partials [0: formed in the kernel])
for shape in [(32, 64, 2, 4096, 12), (32, 128, 2, 4096, 12), (32, 128, 4, 1024, 0), (32, 256, 8, 256, 0)]:
    print(shape, "dense W2 %.1f us, the head blocks only (what the fused backward reads) %.1f us" % (run(*shape, False), run(*shape, True)))
