#!/usr/bin/env python3
"""Training script: the reference's train.py (PromptIRModel + pl.Trainer, train.py:37-120) re-expressed as
one process per GPU with RCCL gradient all-reduce (engine.DataParallelEngine).

    python mp-hsir_amd/train.py --data_type natural_scene --epochs 100 --lr 2e-4 --batch_size 32
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 mp-hsir_amd/train.py ...

Same loss (clamp + L1, :58-61), optimizer (AdamW(lr) defaults, :69), schedule (linear warm-up over
0.1*epochs then cosine to 1e-6, stepped per epoch, :71-85 -- including lr(0) = 0), seed handling (:88-92),
warm start from a Lightning checkpoint by key+shape filtering with the `net.` prefix (:109-116) and a
checkpoint every 50 epochs (:104).  Datasets are not available offline: --synthetic 1 (default) feeds
data.SyntheticPatchSource, which emits the reference's batch tuple on the GPU.
"""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mp_hsir_amd.data import SyntheticPatchSource  # noqa: E402
from mp_hsir_amd.engine import DataParallelEngine, warmup_cosine_lr  # noqa: E402
from mp_hsir_amd.net.MP_HSIR import MP_HSIR_Net  # noqa: E402
from mp_hsir_amd.options import options as opt  # noqa: E402

MODELS = {"natural_scene": dict(in_channel=31, out_channel=31, dim=64, task_classes=6),     # train.py:44
          "remote_sensing": dict(in_channel=100, out_channel=100, dim=96, task_classes=7)}  # train.py:45


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def load_warm_start(net, path, device):
    """keep checkpoint entries whose key AND shape match (train.py:109-116); keys carry the `net.` prefix."""
    state = torch.load(path, map_location=device)["state_dict"]
    own = {"net." + k: v for k, v in net.state_dict().items()}
    kept = {k[4:]: v for k, v in state.items() if k in own and own[k].shape == v.shape}
    net.load_state_dict(kept, strict=False)
    return len(kept)


def main():
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    if rank == 0:
        print("Options\n", opt)
    set_seed(opt.seed)
    cfg = MODELS[opt.model or opt.data_type]
    net = MP_HSIR_Net(**cfg, compute_dtype=torch.bfloat16 if opt.precision == "bf16" else torch.float32).to(dev).train()
    if opt.ckpt_path is not None:
        n = load_warm_start(net, opt.ckpt_path, dev)
        if rank == 0:
            print("warm start: %d tensors from %s" % (n, opt.ckpt_path))
    eng = DataParallelEngine(net, lr=opt.lr, use_graph=bool(opt.graph))
    src = SyntheticPatchSource(cfg["in_channel"], opt.patch_size, opt.batch_size, cfg["task_classes"], dev, opt.seed, rank)
    for epoch in range(opt.epochs):
        lr = warmup_cosine_lr(epoch, opt.lr, opt.epochs)
        running = 0.0
        for it in range(opt.steps_per_epoch):
            _, degraded, clean, prompt = src.next()
            loss = eng.train_step(degraded, clean, prompt, lr=lr)
            if (it + 1) % opt.log_every == 0:
                if world > 1:
                    dist.all_reduce(loss, op=dist.ReduceOp.AVG)       # self.log(..., sync_dist=True), train.py:65
                running = float(loss)
                if rank == 0:
                    print("epoch %d it %d lr %.3e train_loss %.5f" % (epoch, it + 1, lr, running), flush=True)
        if rank == 0 and opt.ckpt_dir and (epoch + 1) % 50 == 0:      # ModelCheckpoint(every_n_epochs=50), train.py:104
            os.makedirs(opt.ckpt_dir, exist_ok=True)
            torch.save({"state_dict": {"net." + k: v.detach().cpu().clone() for k, v in net.state_dict().items()},
                        "epoch": epoch}, os.path.join(opt.ckpt_dir, "epoch=%d.ckpt" % epoch))
    eng.finish()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
