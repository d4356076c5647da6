"""lab: does a second chain of dependent launches hide behind the first one?  Two streams, each a chain of N dependent kernels
(small: latency-bound; medium: ~20 us of work on a quarter of the chip), eager and as branches of ONE captured graph."""
import sys, time, torch

def chain(x, n):
    for _ in range(n):
        x.mul_(1.0001)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def main():
    dev = torch.device("cuda:0")
    n = 400
    for numel in (1 << 14, 1 << 22, 1 << 25):
        a = torch.ones(numel, device=dev, dtype=torch.float32); b = torch.ones(numel, device=dev, dtype=torch.float32)
        s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
        def one():
            chain(a, n)
        def two_serial():
            chain(a, n); chain(b, n)
        def two_streams():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur); s2.wait_stream(cur)
            with torch.cuda.stream(s1): chain(a, n)
            with torch.cuda.stream(s2): chain(b, n)
            cur.wait_stream(s1); cur.wait_stream(s2)
        res = {}
        for name, fn in (("one", one), ("two_serial", two_serial), ("two_streams", two_streams)):
            res["eager_" + name] = timed(fn)
            g = torch.cuda.CUDAGraph()
            cs = torch.cuda.Stream()
            with torch.cuda.stream(cs):
                fn(); torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=cs):
                    fn()
            res["graph_" + name] = timed(g.replay)
        print(f"numel {numel}: " + "  ".join(f"{k} {v:.3f} ms" for k, v in res.items()), flush=True)

main()
