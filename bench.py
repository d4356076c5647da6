#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native MP-HSIR hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json metric "HSI patches/sec (fwd+bwd) 64x64x31 bf16", configs[2]): natural-scene
MP_HSIR_Net(31,31,64,T=6), random init, synthetic 64x64x31 patches generated on the GPU, batch 32 per GPU,
bf16 compute, one *full* training step per "step": forward, L1-after-clamp loss, backward, gradient
all-reduce (RCCL, N>1), fused AdamW.  Weak scaling: per-GPU batch fixed.  Rank 0 prints one JSON line.

Extra objects on the line:
  roofline     the dominant HIP kernel of the step: algorithmic FLOPs (or bytes) per launch / its mean
               launch duration, measured live with HIP events on the launch stream (mphsir_prof_*).
  roofline_spectral  the kernel north_star names: the spectral attention at 512x512x31 bf16 (forward of one test cube):
               cubes/s through the replayed graph, and the depthwise+Gram pass against the HBM roof / its QK^T FLOPs against
               the MFMA peak (N=1, rank 0).
  cpu_baseline the CPU oracle (a port of the reference's op sequence, oracle/mp_hsir_oracle.py) timed
               on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time
import warnings

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md)
PEAK_MFMA_TF = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}
MODELS = {"natural_scene": dict(in_channel=31, out_channel=31, dim=64, task_classes=6),      # test.py:39
          "remote_sensing": dict(in_channel=100, out_channel=100, dim=96, task_classes=7),   # train.py:45
          "rs172": dict(in_channel=172, out_channel=172, dim=96, task_classes=7)}            # BASELINE configs[3]
DTYPES = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}
RIDGE = 312.0                  # FLOP/B, bf16 dense MFMA peak / HBM peak

def kernel_ids(lib):
    """name -> id of every kernel the library can time (mphsir_kernel_name)."""
    lib.mphsir_kernel_name.restype = ctypes.c_char_p
    out = {}
    for kid in range(32):
        n = lib.mphsir_kernel_name(kid).decode()
        if n != "?":
            out[n] = kid
    return out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="patches per GPU per step")
    ap.add_argument("--patch", type=int, default=64, help="patch height = width (diagnostic: 512 = the test cubes of test.py)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16"], help="f16 = the reference's 16-mixed (dynamic loss scaling)")
    ap.add_argument("--model", default="natural_scene", choices=sorted(MODELS), help="headline = natural_scene; the others are BASELINE configs[3]/[4]")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra BASELINE configurations of the N=1 line")
    ap.add_argument("--forward-only", action="store_true", help="diagnostic: time inference only (not the headline metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-spectral", action="store_true", help="skip the 512x512 spectral-attention roofline leg")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of replaying the captured step (hipGraph)")
    return ap.parse_args()


def cube_forward_worker(out_path, threads):
    """child process: ONE 512x512x31 forward of the CPU oracle (SURVEY 8d's fourth shape; ~2 min on 32 threads), started at
    the beginning of the run so that it overlaps the GPU legs; writes seconds to out_path"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    torch.set_num_threads(threads)
    from oracle import mp_hsir_oracle as O
    from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
    torch.manual_seed(2024)
    net = MP_HSIR_Net(clip_prompt="surrogate")
    P = {k: v.detach().clone() for k, v in net.state_dict().items() if not k.endswith("attn_mask")}
    x = torch.rand(1, 31, 512, 512)
    t0 = time.perf_counter()
    with torch.no_grad():
        O.mp_hsir_forward(P, O.make_cfg(), x, torch.tensor([[0]]), net.clip_prompts)
    json.dump({"seconds": time.perf_counter() - t0, "threads": threads}, open(out_path, "w"))


def cpu_baseline(budget_s=30.0):
    """The CPU oracle (oracle/mp_hsir_oracle.py: the reference's op sequence in eager PyTorch, pinned to the reference by
    tests/golden) timed on this box's host cores on a bounded sample of the workload: fwd+bwd of batch 4 (SURVEY 8d shape c),
    1 warm-up + 3 timed iterations, median, at a few thread counts (all cores capped at 32: oversubscribing a 128-core
    host made round 1's number 4x too low) -- `value` is the best of them, `cores` its thread count.  `detail` also holds
    the forward-only shapes (a) B=1 and (b) B=16 and the reference's own setting of one thread (train.py:34)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import mp_hsir_oracle as O
    from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
    torch.manual_seed(2024)
    net = MP_HSIR_Net(clip_prompt="surrogate")
    P = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()
         if not k.endswith("attn_mask")}
    cfg = O.make_cfg()
    clip = net.clip_prompts
    ncpu = os.cpu_count() or 1
    t_start = time.perf_counter()

    def timed(fn, iters=3):
        fn()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > budget_s:
                break
        return sorted(ts)[len(ts) // 2]

    def fwd(batch):
        x, task = torch.rand(batch, 31, 64, 64), torch.randint(0, 6, (batch, 1))
        def run():
            with torch.no_grad():
                O.mp_hsir_forward(P, cfg, x, task, clip)
        return run

    def fwd_bwd(batch):
        x, c, task = torch.rand(batch, 31, 64, 64), torch.rand(batch, 31, 64, 64), torch.randint(0, 6, (batch, 1))
        def run():
            for v in P.values():
                v.grad = None
            O.l1_after_clamp(O.mp_hsir_forward(P, cfg, x, task, clip), c).backward()
        return run
    prev = torch.get_num_threads()
    detail = {}
    best = None
    for nt in sorted({min(ncpu, 8), min(ncpu, 32)}, reverse=True):
        torch.set_num_threads(nt)
        t = timed(fwd_bwd(4))
        detail["fwd_bwd_b4_threads%d" % nt] = round(4 / t, 4)
        if best is None or 4 / t > best[0]:
            best = (4 / t, nt)
    torch.set_num_threads(best[1])
    dropped = []
    detail["fwd_b1_threads%d" % best[1]] = round(1 / timed(fwd(1)), 4)
    if time.perf_counter() - t_start < budget_s:
        detail["fwd_b16_threads%d" % best[1]] = round(16 / timed(fwd(16), iters=1), 4)
    else:
        dropped.append("fwd_b16")
    if time.perf_counter() - t_start < budget_s:
        torch.set_num_threads(1)
        detail["fwd_b1_threads1"] = round(1 / timed(fwd(1), iters=1), 4)
    else:
        dropped.append("fwd_b1_threads1")
    torch.set_num_threads(prev)
    return {"value": round(best[0], 4), "unit": "patches/s", "cores": best[1], "kind": "port", "host_cpus": ncpu, "detail": detail,
            "dropped_legs": dropped,
            "sample": "natural 64x64x31 fp32, torch CPU oracle: fwd+bwd of batch 4, median of <=3 after 1 warm-up, best of the thread "
                      "counts in detail (patches/s each); bounded to ~%ds" % int(budget_s)}


def spectral_roofline(net, dev, lib, steps=20):
    """What north_star names: the spectral attention at 512x512x31 bf16 (forward, batch 1 = one test cube of test.py).
    Times the forward (eager warm-up, then the replayed hipGraph) and, with HIP events on the launch stream, the kernels
    of the global spectral branch; reports them against the roof that bounds each:
      qkv_dwconv_gram   fused pass A (1x1 qkv + depthwise 3x3 + Gram + norms; t, q, k never in HBM): algorithmic bytes
                        (x in, v out) / time vs 8 TB/s, and its useful FLOPs / time vs the dense bf16 MFMA peak
      dwconv_gram       the two-kernel form's depthwise + Gram kernel where it still runs (cross attention)
      qk_*              the QK^T FLOPs inside either (2*C*hd per pixel) / the same time vs the MFMA peak"""
    from mp_hsir_amd import ops
    from mp_hsir_amd.engine import GraphedForward
    net.eval()
    g = torch.Generator(device=dev).manual_seed(7)
    x = torch.rand((1, 31, 512, 512), generator=g, device=dev) + 0.1 * torch.randn((1, 31, 512, 512), generator=g, device=dev)
    p = torch.tensor([0], device=dev)
    with torch.no_grad():
        for _ in range(2):
            net(x, p)
        # a kernel's own duration: the parallel branches (prompt gate, prompt modules) are issued in line for the per-kernel table, as
        # in the training leg -- beside each other their launches stretch; cubes_per_s below is measured WITH the branches
        side_state = (ops.SIDE_BRANCH, ops.PROMPT_SIDE)
        ops.SIDE_BRANCH, ops.PROMPT_SIDE = False, False
        ops.ACCOUNT = {}
        net(x, p)
        torch.cuda.synchronize()
        acct, ops.ACCOUNT = ops.ACCOUNT, None
        per = {}
        for name, kid in kernel_ids(lib).items():
            if name not in acct:
                continue
            lib.mphsir_prof_enable(kid)
            net(x, p)
            n, ms = ctypes.c_int(0), ctypes.c_float(0)
            lib.mphsir_prof_read(ctypes.byref(n), ctypes.byref(ms))
            lib.mphsir_prof_enable(-1)
            if n.value:
                per[name] = (n.value, ms.value)
        ops.SIDE_BRANCH, ops.PROMPT_SIDE = side_state
        run = GraphedForward(net, warmup=0)
        run(x, p)
        torch.cuda.synchronize()
        t_fwd = None
        for _ in range(3):                      # best of three runs of `steps` replays
            t0 = time.perf_counter()
            for _ in range(steps):
                run(x, p)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t0) / steps
            t_fwd = t if t_fwd is None else min(t_fwd, t)
    out = {"workload": "natural-scene net, one 512x512x31 cube, bf16 forward (test.py shape), hipGraph replay",
           "cubes_per_s": round(1.0 / t_fwd, 2), "ms_per_cube": round(t_fwd * 1e3, 3),
           "kernel_table": {k: [int(v[0]), round(v[1], 3), round(acct[k][2] / 1e9, 3), round(acct[k][2] / v[1] / 1e9, 2),
                                round(acct[k][1] / v[1] / 1e9, 1)] for k, v in sorted(per.items())}}
    # pass A of the global spectral attention: the fused kernel (LN + 1x1 qkv + depthwise 3x3 + Gram, t on-chip) on the no-grad
    # path, the depthwise+Gram kernel wherever the two-kernel path still runs (cross attention).  AI of the fused pass =
    # (6 C^2 + ...) FLOP per 4 C bytes ~ 1.5 C FLOP/B: HBM side of the ridge (312 FLOP/B) at C <= 128 -> priced against HBM,
    # with the MFMA fraction of its useful FLOPs beside it.
    for kname in ("qkv_dwconv_gram", "dwconv_gram"):
        if kname in per:
            n, ms = per[kname]
            nbytes, flops = acct[kname][2], acct[kname][1]
            qk = acct.get(kname + ":qk", [0, 0.0, 0])[1]              # the QK^T (Gram) FLOPs alone: 2*C*hd per pixel
            out[kname] = {"bound": "hbm", "achieved": round(nbytes / ms / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": round(nbytes / ms / 1e6 / PEAK_HBM_GBS, 4), "launches": int(n), "avg_launch_us": round(ms * 1e3 / n, 2),
                          "algorithmic_bytes_per_launch": round(nbytes / n),
                          "tflops": round(flops / ms / 1e9, 1), "mfma_frac": round(flops / ms / 1e9 / PEAK_MFMA_TF["bf16"], 4),
                          "qk_tflops": round(qk / ms / 1e9, 1),
                          "qk_mfma_frac": round(qk / ms / 1e9 / PEAK_MFMA_TF["bf16"], 4),
                          "qk_ceiling_at_hbm_peak_tflops": round(qk / (nbytes / (PEAK_HBM_GBS * 1e9)) / 1e12, 1)}
    return out


def timed_leg(step, steps, warmup, fence, repeats=2):
    """seconds per step: the faster of `repeats` timed runs of `steps` steps (the CPU oracle's 512x512 forward runs on 32 host
    threads beside these legs; a 4 ms replayed step is short enough for that to show in a single run)"""
    for _ in range(warmup):
        step()
    best = None
    for _ in range(repeats):
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        t = (time.perf_counter() - t0) / steps
        best = t if best is None else min(best, t)
    return best


def extra_configs(dev):
    """The other BASELINE.json configurations on the same code (N = 1 line only; the headline stays configs[2]):
    configs[1] natural net forward batch 16 bf16, configs[3] RS-width 172-band 512x512 cube forward bf16, configs[4]
    remote-sensing training batch 16 fp16 + dynamic loss scaling.  Each: hipGraph replay, a few steps after the capture."""
    from mp_hsir_amd.data import SyntheticPatchSource
    from mp_hsir_amd.engine import DataParallelEngine, GraphedForward
    from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
    out = {}

    def sync():
        torch.cuda.synchronize()

    def fwd_leg(model, dtype, batch, patch, steps):
        cfg = MODELS[model]
        torch.manual_seed(2024)
        net = MP_HSIR_Net(**cfg, compute_dtype=DTYPES[dtype], clip_prompt="surrogate").to(dev).eval()
        src = SyntheticPatchSource(cfg["in_channel"], patch, batch, cfg["task_classes"], dev, 2024, 0, pool=4).prefill()
        run = GraphedForward(net)
        def step():
            _, x, c, p = src.next()
            with torch.no_grad():
                return run(x, p)
        t = timed_leg(step, steps, 3, sync)
        del run, net
        torch.cuda.empty_cache()
        return t

    def train_leg(model, dtype, batch, steps):
        cfg = MODELS[model]
        torch.manual_seed(2024)
        net = MP_HSIR_Net(**cfg, compute_dtype=DTYPES[dtype], clip_prompt="surrogate").to(dev).train()
        src = SyntheticPatchSource(cfg["in_channel"], 64, batch, cfg["task_classes"], dev, 2024, 0, pool=8).prefill()
        eng = DataParallelEngine(net, lr=2e-4, use_graph=True)
        def step():
            _, x, c, p = src.next()
            return eng.train_step(x, c, p)
        t = timed_leg(step, steps, 4, sync)
        scale = None if eng.scaler is None else float(eng.scaler[0])
        del eng, net
        torch.cuda.empty_cache()
        return t, scale

    t = fwd_leg("natural_scene", "bf16", 16, 64, 20)
    out["configs[1] natural forward b16 bf16"] = {"patches_per_s": round(16 / t, 1), "ms_per_step": round(t * 1e3, 3)}
    t = fwd_leg("rs172", "bf16", 1, 512, 5)
    out["configs[3] rs172 512x512x172 forward b1 bf16"] = {"cubes_per_s": round(1 / t, 2), "ms_per_cube": round(t * 1e3, 3)}
    t, scale = train_leg("remote_sensing", "f16", 16, 10)
    out["configs[4] remote_sensing training b16 fp16+loss scaling"] = {"patches_per_s": round(16 / t, 1), "ms_per_step": round(t * 1e3, 3),
                                                                        "loss_scale_after": scale}
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks here, one fresh process per GPU (the reference's
    `devices=opt.num_gpus`, train.py:118), relay rank 0's JSON line and fail if any rank fails.  Runs before this process
    touches the GPU (device_count() does not initialise it) and never re-execs: the children are ordinary subprocesses.
    MPHSIR_SHARE_GPU=1 (test hook, with MPHSIR_DIST_BACKEND=gloo) lets the ranks share the GPUs that exist."""
    import socket
    import subprocess
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n and os.environ.get("MPHSIR_SHARE_GPU", "0") != "1":
        print("bench.py: --gpus %d asked for, %d GPU(s) visible on this node (set MPHSIR_SHARE_GPU=1 MPHSIR_DIST_BACKEND=gloo to "
              "let ranks share a GPU: a test hook, not a benchmark)" % (n, have), file=sys.stderr, flush=True)
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's line is read on a thread while EVERY child is polled: a rank that dies early (bad device, out of memory, RCCL
    # initialisation) would otherwise leave rank 0 in init_process_group / its first all-reduce and this process blocked in read()
    # until the outer harness gives up.  One failed rank, or the overall limit, ends all of them.
    import threading
    import time
    buf = []
    reader = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    limit = float(os.environ.get("MPHSIR_BENCH_SPAWN_TIMEOUT", "3000"))
    t0, failed = time.time(), False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() - t0 > limit:
            failed = True
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t1 = time.time()
            while any(p.poll() is None for p in procs) and time.time() - t1 < 10:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    codes = [p.wait() for p in procs]
    reader.join(timeout=5)
    sys.stdout.write(b"".join(buf).decode())
    sys.stdout.flush()
    if failed and all(c == 0 for c in codes):
        codes[0] = 124                                   # the overall limit
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print("bench.py: rank(s) failed: %s" % ", ".join("rank %d exit %d" % rc for rc in bad), file=sys.stderr, flush=True)
        return 1
    return 0


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    cube_proc, cube_out = None, None
    if world > 1 and torch.cuda.device_count() < world and os.environ.get("MPHSIR_SHARE_GPU", "0") != "1" \
            and os.environ.get("MPHSIR_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit("bench.py: %d ranks but %d GPU(s) visible" % (world, torch.cuda.device_count()))
    dev = torch.device("cuda", local % max(1, torch.cuda.device_count()))
    torch.cuda.set_device(dev)
    backend = None
    if world > 1:
        # RCCL over xGMI.  MPHSIR_DIST_BACKEND=gloo is a TEST hook: it lets the whole N>1 path (broadcast, gradient
        # all-reduce outside the captured step, AdamW + repack after it) run with several ranks sharing one GPU box.
        backend = os.environ.get("MPHSIR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from mp_hsir_amd import _lib, ops
    from mp_hsir_amd.data import SyntheticPatchSource
    from mp_hsir_amd.engine import DataParallelEngine
    from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net
    lib = _lib.load()
    dt = DTYPES[args.dtype]
    cfg = MODELS[args.model]
    bands = cfg["in_channel"]
    torch.manual_seed(2024)
    net = MP_HSIR_Net(**cfg, compute_dtype=dt, clip_prompt="surrogate").to(dev)      # no CLIP weights offline: seeded stand-in (timing only)
    # a pool of 8 pre-generated batches, resident in HBM before the timed region, handed out round-robin (the timed step is the
    # model's step: the reference's loader workers run beside the GPU)
    src = SyntheticPatchSource(bands, args.patch, args.batch, cfg["task_classes"], dev, 2024, rank, pool=8).prefill()
    if args.forward_only:
        net.eval()
        from mp_hsir_amd.engine import GraphedForward
        fwd = net if args.no_graph else GraphedForward(net)
        def step():
            _, x, c, p = src.next()
            with torch.no_grad():
                return fwd(x, p)
    else:
        net.train()
        eng = DataParallelEngine(net, lr=2e-4, use_graph=not args.no_graph)
        def step():
            _, x, c, p = src.next()
            return eng.train_step(x, c, p)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 0 if args.forward_only or args.no_graph else 3)):   # graph mode: 2 eager steps + capture
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt_s = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt_s], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_s = float(t)
    value = world * args.batch * args.steps / dt_s
    comm = None
    if world > 1 and not args.forward_only:
        # what the collective costs on its own: the gradient arena reduced bucket by bucket, as in a step, nothing else running
        fence()
        t0 = time.perf_counter()
        for _ in range(5):
            for st, en, _ in eng.buckets:
                dist.all_reduce(eng.flat_g[st:en], group=eng.pg)
        fence()
        comm = {"backend": "rccl" if backend == "nccl" else backend, "ranks": world, "gradient_bytes": int(eng.flat_g.numel() * 4),
                "buckets": len(eng.buckets), "allreduce_ms_per_step_alone": round((time.perf_counter() - t0) / 5 * 1e3, 3),
                "overlap": ("graph: bucket all-reduces start from external event nodes inside the replay" if getattr(eng, "_overlap", False)
                            else ("graph: one arena-wide all-reduce after the replay" if not args.no_graph else "eager: bucketed all-reduces from gradient hooks"))}

    roofline = None
    if not args.no_roofline:
        # algorithmic work per step and per kernel (one accounted step), then per-kernel HIP-event timing (eager launches)
        if not args.forward_only:
            eng.force_eager = True          # same data flow (incl. the all-reduce for N > 1), individual launches
        else:
            fwd = net
        # a kernel's own duration: the parallel branches of the step (weight-gradient branch, prompt gate, prompt modules) are issued
        # in line for this leg -- beside each other their launches stretch (the headline above is measured WITH them)
        side_state = (ops.DW_SIDE, ops.SIDE_BRANCH, ops.PROMPT_SIDE)
        ops.DW_SIDE, ops.SIDE_BRANCH, ops.PROMPT_SIDE = 0, False, False
        ops.ACCOUNT = {}
        step()
        torch.cuda.synchronize()
        acct, ops.ACCOUNT = ops.ACCOUNT, None
        per = {}
        for name, kid in kernel_ids(lib).items():
            if name not in acct and name != "flat_adamw":
                continue
            lib.mphsir_prof_enable(kid)
            step()
            n, ms = ctypes.c_int(0), ctypes.c_float(0)
            lib.mphsir_prof_read(ctypes.byref(n), ctypes.byref(ms))
            lib.mphsir_prof_enable(-1)
            if n.value:
                per[name] = (float(n.value), ms.value)              # launches / step, ms / step
        ops.DW_SIDE, ops.SIDE_BRANCH, ops.PROMPT_SIDE = side_state
        dom = max((k for k in per if k in acct), key=lambda k: per[k][1])
        launches, ms = per[dom]
        _, flops, nbytes = acct[dom]
        ai = flops / max(nbytes, 1.0)
        if ai >= RIDGE:
            ach, peak, unit, bound = flops / (ms * 1e-3) / 1e12, PEAK_MFMA_TF[args.dtype], "TFLOP/s", "mfma"
        else:
            ach, peak, unit, bound = nbytes / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
        # HBM bytes per launch from the COMMITTED rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE) of the newest training profile
        # under profiles/ -- a file, not this run (PMC collection needs rocprofv3 around the process): `traffic_source` names it
        # ... and the same kernel's average launch duration in that profile (rocprofv3 --kernel-trace of the REPLAYED step, where the
        # parallel branches of the captured graph run beside each other): `frac_profile` prices the algorithmic work against it,
        # `frac_inline` (= `frac`) against this run's HIP-event timing with the branches issued in line
        traffic, traffic_source, prof_us = None, None, None
        try:
            import glob
            pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_train_b32_bf16_graph_pmc_summary.json")))
            if pm and not args.forward_only and args.model == "natural_scene" and args.dtype == "bf16" and args.batch == 32:
                e = json.load(open(pm[-1])).get(dom, {})
                if "hbm_read_bytes_per_launch" in e:
                    traffic = e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]
                    traffic_source = "committed profile " + os.path.relpath(pm[-1], ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per MI355X_MICROARCH.md)"
                prof_us = e.get("avg_us")
        except Exception:
            traffic, traffic_source, prof_us = None, None, None
        per_launch = (flops if bound == "mfma" else nbytes) / launches
        frac_profile = round(per_launch / (prof_us * 1e-6) / (1e12 if bound == "mfma" else 1e9) / peak, 4) if prof_us else None
        roofline = {"kernel": dom, "bound": bound, "achieved": round(ach, 3), "peak": peak, "unit": unit,
                    "frac": round(ach / peak, 4), "frac_inline": round(ach / peak, 4), "frac_profile": frac_profile, "profile_avg_launch_us": prof_us,
                    "traffic": traffic, "traffic_source": traffic_source, "algorithmic_per_launch": round(per_launch),
                    "launches_per_step": launches,
                    # bytes the kernel moves beyond its inputs (split-K partial sums): overhead, NOT part of `achieved`
                    "partials_bytes_per_step": acct.get(dom + ":partials", [0, 0.0, 0.0])[2],
                    "avg_launch_us": round(ms * 1e3 / launches, 2), "flops_per_step": flops, "bytes_per_step": nbytes,
                    "tflops_equiv": round(flops / (ms * 1e-3) / 1e12, 2),
                    "timing": "HIP events around eager launches of one kernel id at a time, parallel branches issued in line",
                    "kernel_ms_per_step": {k: round(v[1], 3) for k, v in sorted(per.items())},
                    # per kernel: [launches, ms, algorithmic GB, achieved TB/s, achieved TFLOP/s] per step
                    "kernel_table": {k: [int(v[0]), round(v[1], 3), round(acct[k][2] / 1e9, 3), round(acct[k][2] / v[1] / 1e9, 2),
                                         round(acct[k][1] / v[1] / 1e9, 1)] for k, v in sorted(per.items()) if k in acct}}

    spectral = None
    if rank == 0 and world == 1 and not args.no_roofline and not args.forward_only and args.dtype == "bf16" and not args.no_spectral \
            and args.model == "natural_scene":
        eng.finish()
        torch.cuda.empty_cache()
        spectral = spectral_roofline(net, dev, lib)

    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "natural_scene" and not args.forward_only:
        # the oracle's 512x512 forward (20-30 s on 32 host threads) is started only now -- after the headline, the per-kernel and the
        # spectral legs -- as an ordinary child process: it overlaps the extra configurations below (each the best of two runs)
        import subprocess
        import tempfile
        cube_out = os.path.join(tempfile.gettempdir(), "mphsir_cpu_cube_%d.json" % os.getpid())
        cube_proc = subprocess.Popen([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench; bench.cube_forward_worker(%r, %d)"
                                      % (ROOT, cube_out, min(32, os.cpu_count() or 1))], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    extra = None
    if rank == 0 and world == 1 and not args.no_extra and not args.forward_only and args.model == "natural_scene" and args.dtype == "bf16":
        if spectral is None:
            eng.finish()
        del eng
        torch.cuda.empty_cache()
        extra = extra_configs(dev)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cube_done = False
        if cube_proc is not None:                 # let it finish first: the timed legs below get the host to themselves
            try:
                cube_proc.wait(timeout=150)
                cube_done = True
            except Exception:
                cube_proc.kill()
        cpu = cpu_baseline()
        if cube_proc is not None:
            try:
                if not cube_done:
                    raise RuntimeError("timeout")
                r = json.load(open(cube_out))
                cpu["detail"]["fwd_512x512_b1_threads%d_cubes_per_s" % r["threads"]] = round(1.0 / r["seconds"], 5)
                cpu["detail"]["fwd_512x512_b1_seconds"] = round(r["seconds"], 1)
            except Exception:
                cpu["dropped_legs"].append("fwd_512x512_b1 (not finished 150 s after the GPU legs)")

    if rank == 0:
        line = {
            "metric": "hsi_patches_per_sec_fwd" if args.forward_only else "hsi_patches_per_sec_fwd_bwd",
            "value": round(value, 2), "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_s / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic (8 pre-generated batches resident in HBM, round-robin)",
            "config": {"workload": "%s MP_HSIR_Net(%d,%d,%d,T=%d) %s, %dx%dx%d patches, batch %d/GPU, %s"
                                   % (args.model.replace("_", "-"), bands, bands, cfg["dim"], cfg["task_classes"],
                                      "forward" if args.forward_only else "training step fwd+bwd+allreduce+AdamW",
                                      args.patch, args.patch, bands, args.batch, "dp%d" % world),
                       "global_batch": world * args.batch, "patch": "%dx%dx%d" % (args.patch, args.patch, bands), "parallelism": "dp%d" % world, "launch": "eager" if args.no_graph else "hipGraph replay",
                       "backward": "HIP kernels for every module (fused block / prompt-module backward, token-reduction GEMMs, gemm_tok data gradients); no library GEMM on the step"},
            "roofline": roofline, "roofline_spectral": spectral, "cpu_baseline": cpu, "extra_configs": extra, "comm": comm,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
