"""Diagnostic: duration of the PG-gate kernels per launch size (run under rocprofv3 --kernel-trace for pure kernel times)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mp_hsir_amd import ops

dev = "cuda"
for C, cr, nW in [(128, 8, 2048), (64, 8, 2048), (128, 16, 512), (256, 32, 128), (128, 8, 128)]:
    r = C // cr
    g = torch.Generator().manual_seed(1)
    pg = {"linear_down.weight": torch.randn(r, C, generator=g) * 0.1, "linear_up.weight": torch.randn(C, r, generator=g) * 0.1,
          "linear_prompt.weight": torch.randn(128, C, generator=g) * 0.1, "prompt_param": torch.rand(128, r, generator=g),
          "q.weight": torch.randn(r, r, generator=g) * 0.2, "kv.weight": torch.randn(2 * r, r, generator=g) * 0.2,
          "proj.weight": torch.randn(r, r, generator=g) * 0.2, "proj.bias": torch.randn(r, generator=g) * 0.1}
    pg = {k: v.to(dev).contiguous() for k, v in pg.items()}
    mu, dg = torch.randn(nW, C, device=dev), torch.randn(nW, C, device=dev)
    for name, f in (("fwd", lambda: ops.pg_gate_fwd(mu, pg)), ("bwd", lambda: ops.pg_gate_bwd(mu, dg, pg, factor_dtype=torch.bfloat16))):
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            f()
        e1.record()
        torch.cuda.synchronize()
        print("C=%d r=%d nW=%d %s: %.1f us per call (back-to-back, bwd includes its gemm_tn)" % (C, r, nW, name, e0.elapsed_time(e1) * 1e3 / 50), flush=True)

# per-phase shader-clock stamps of workgroup 0 (mphsir_debug)
import ctypes
from mp_hsir_amd import _lib
lib = _lib.load()
for C, cr, nW in [(128, 8, 2048), (256, 32, 128)]:
    r = C // cr
    g = torch.Generator().manual_seed(1)
    pg = {"linear_down.weight": torch.randn(r, C, generator=g) * 0.1, "linear_up.weight": torch.randn(C, r, generator=g) * 0.1,
          "linear_prompt.weight": torch.randn(128, C, generator=g) * 0.1, "prompt_param": torch.rand(128, r, generator=g),
          "q.weight": torch.randn(r, r, generator=g) * 0.2, "kv.weight": torch.randn(2 * r, r, generator=g) * 0.2,
          "proj.weight": torch.randn(r, r, generator=g) * 0.2, "proj.bias": torch.randn(r, generator=g) * 0.1}
    pg = {k: v.to(dev).contiguous() for k, v in pg.items()}
    mu, dg = torch.randn(nW, C, device=dev), torch.randn(nW, C, device=dev)
    stamps = torch.zeros(32, dtype=torch.int64, device=dev)
    lib.mphsir_debug(0, ctypes.c_void_p(stamps.data_ptr()))
    for name, f, n in (("fwd", lambda: ops.pg_gate_fwd(mu, pg), 9), ("bwd", lambda: ops.pg_gate_bwd(mu, dg, pg, factor_dtype=torch.bfloat16), 17)):
        f(); f()
        stamps.zero_()
        f()
        torch.cuda.synchronize()
        t = stamps.cpu().tolist()[:n]
        print("C=%d r=%d %s phase cycles (100 MHz ticks?):" % (C, r, name), [t[i + 1] - t[i] for i in range(n - 1)], "total", t[n - 1] - t[0], flush=True)
    lib.mphsir_debug(0, None)
