"""Host-side logic without a GPU: flags, LR schedule, PSNR, synthetic source, data-parallel engine
(world_size 2 over gloo; the AdamW kernel runs through the CPU-emulated build of the same source)."""
import math
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flags_match_the_reference_defaults():
    from mp_hsir_amd.options import build_parser
    o = build_parser().parse_args([])
    want = dict(cuda=0, seed=2024, epochs=100, batch_size=32, lr=2e-4, init="xu", mode=0, patch_size=64, num_workers=16,
                data_type="remote_sensing", classifier=False, db_path="", classifier_path="", output_path="output/",
                ckpt_path=None, ckpt_dir="", num_gpus=[0], repeat=1)
    for k, v in want.items():
        assert getattr(o, k) == v, k
    assert o.natural_scene_single_de_type == ["gaussianN", "complexN", "blur", "sr", "inpaint", "bandmiss"]
    assert o.remote_sensing_single_de_type == ["gaussianN", "complexN", "blur", "sr", "inpaint", "haze", "bandmiss"]
    o = build_parser().parse_args(["--num_gpus", "01", "--epochs", "300", "--lr", "1e-4"])     # README.md:38 style
    assert o.num_gpus == ["0", "1"] and o.epochs == 300 and o.lr == 1e-4


def test_lr_schedule_matches_oracle_and_q19():
    from mp_hsir_amd.engine import warmup_cosine_lr
    from oracle import mp_hsir_oracle as O
    for epochs, base in ((100, 2e-4), (300, 1e-4)):
        for e in range(epochs + 1):
            assert warmup_cosine_lr(e, base, epochs) == O.warmup_cosine_lr(e, base, epochs)
    assert warmup_cosine_lr(0, 2e-4, 100) == 0.0


def test_psnr_matches_oracle():
    sys.path.insert(0, os.path.join(ROOT, "mp-hsir_amd"))
    import importlib
    T = importlib.import_module("mp_hsir_amd.test")
    from oracle import mp_hsir_oracle as O
    a, b = torch.rand(2, 5, 16, 16) * 1.4 - 0.2, torch.rand(2, 5, 16, 16)
    assert abs(T.psnr_bandwise(a, b) - O.psnr_bandwise(a, b)) < 1e-12


def test_synthetic_source_contract():
    from mp_hsir_amd.data import SyntheticPatchSource
    s = SyntheticPatchSource(31, 64, 4, 6, "cpu", 2024, 0)
    (names, de), degraded, clean, prompt = s.next()
    assert degraded.shape == clean.shape == (4, 31, 64, 64) and degraded.dtype == torch.float32
    assert prompt.shape == (4, 1) and prompt.dtype == torch.int64 and int(prompt.max()) < 6
    assert float(clean.amin()) == 0.0 and float(clean.amax()) == 1.0 and len(names) == 4
    s2 = SyntheticPatchSource(31, 64, 4, 6, "cpu", 2024, 0)
    assert torch.equal(s2.next()[1], degraded)                       # reproducible
    assert not torch.equal(SyntheticPatchSource(31, 64, 4, 6, "cpu", 2024, 1).next()[1], degraded)   # per-rank shard


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from emu import bind_emulator
bind_emulator()
from mp_hsir_amd.engine import DataParallelEngine
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(100 + rank)           # different init per rank: the engine must broadcast rank 0's
class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Conv2d(3, 8, 3, padding=1); self.b = torch.nn.Conv2d(8, 3, 3, padding=1)
        self.unused = torch.nn.Linear(5, 5)          # never receives a gradient (SURVEY Q3)
    def forward(self, x, prompt):
        return self.b(torch.nn.functional.gelu(self.a(x))) + x
net = Net()
eng = DataParallelEngine(net, lr=1e-2, bucket_mb=0.0005)     # tiny buckets -> several all-reduces
g = torch.Generator().manual_seed(7)
xs = torch.rand(3, 4, 3, 8, 8, generator=g); cs = torch.rand(3, 4, 3, 8, 8, generator=g)
half = slice(rank * 2, rank * 2 + 2)
losses = [float(eng.train_step(xs[i][half], cs[i][half], None)) for i in range(3)]
out = {k: v.detach().clone() for k, v in net.state_dict().items()}
torch.save({"losses": losses, "state": out, "nbuckets": len(eng.buckets), "unused": len(eng.unused)}, %(out)r + str(rank))
dist.destroy_process_group()
'''


@pytest.mark.timeout(300)
def test_data_parallel_engine_world2_gloo(tmp_path):
    """2 ranks x batch 2 over gloo == 1 process x batch 4: same parameters after 3 AdamW steps, ranks identical,
    the parameter without gradient is left untouched (no weight decay), several buckets were reduced."""
    script = tmp_path / "worker.py"
    out = str(tmp_path / "res")
    script.write_text(WORKER % dict(root=ROOT, out=out))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2", OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=280) == 0
    r0, r1 = torch.load(out + "0"), torch.load(out + "1")
    for k in r0["state"]:
        assert torch.equal(r0["state"][k], r1["state"][k]), k
    assert r0["nbuckets"] >= 2 and r0["unused"] == 2

    # single-process reference on the full batch with torch.optim.AdamW (the reference's optimizer)
    torch.manual_seed(100)
    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.Conv2d(3, 8, 3, padding=1); self.b = torch.nn.Conv2d(8, 3, 3, padding=1)
            self.unused = torch.nn.Linear(5, 5)
        def forward(self, x, prompt):
            return self.b(torch.nn.functional.gelu(self.a(x))) + x
    net = Net()
    unused0 = net.unused.weight.detach().clone()
    opt = torch.optim.AdamW(net.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(7)
    xs = torch.rand(3, 4, 3, 8, 8, generator=g); cs = torch.rand(3, 4, 3, 8, 8, generator=g)
    for i in range(3):
        opt.zero_grad(set_to_none=True)
        ((net(xs[i], None).clamp(0, 1) - cs[i]).abs().mean()).backward()
        opt.step()
    for k, v in net.state_dict().items():
        assert torch.allclose(r0["state"][k], v, rtol=2e-5, atol=2e-6), k
    assert torch.equal(r0["state"]["unused.weight"], unused0)


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus 8` on a node with fewer GPUs exits non-zero with a message (it used to benchmark one GPU and print
    n_gpus: 1); the check runs before anything touches a GPU, so it holds on the CPU-only build box too."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MPHSIR_SHARE_GPU")}
    if torch.cuda.device_count() >= 8:
        pytest.skip("this node has 8 GPUs")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
