#!/usr/bin/env python3
"""Training script: the reference's train.py (PromptIRModel + pl.Trainer, train.py:37-120) re-expressed as
one process per GPU with RCCL gradient all-reduce (engine.DataParallelEngine).

    python mp-hsir_amd/train.py --data_type natural_scene --epochs 100 --lr 2e-4 --batch_size 32
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 mp-hsir_amd/train.py ...

Same loss (clamp + L1, :58-61), optimizer (AdamW(lr) defaults, :69), schedule (linear warm-up over
0.1*epochs then cosine to 1e-6, stepped per epoch, :71-85 -- including lr(0) = 0), seed handling (:88-92),
warm start from a Lightning checkpoint by key+shape filtering with the `net.` prefix (:109-116) and a
checkpoint every 50 epochs (:104).  Data: --synthetic 1 (default; datasets are not available offline) feeds
data.SyntheticPatchSource; --synthetic 0 --db_path <dir> reads the reference's patch records (data.PatchDB, the LMDB
record format in a flat file).  Either way the per-task degradations of --*_single_de_type are synthesised on the GPU
(degrade.DegradationSynthesizer).  Checkpoints carry, besides the `net.`-prefixed state_dict, AdamW's moments and step
count and the CLIP text-embedding table the model was built with.  --ckpt_path is the reference's warm start (weights whose
key and shape match, epoch 0); --resume 1 additionally restores the optimizer state and continues at the saved epoch + 1.
"""
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from mp_hsir_amd.data import REMOTE_SENSING_SOURCES, PatchDB, PatchDBSource, SyntheticPatchSource  # noqa: E402
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


def saved_clip_table(ckpt, task_classes):
    """the CLIP text-embedding table a checkpoint of this script carries, if it fits a model with `task_classes` tasks
    (a natural-scene checkpoint, 6 x 512, warm-starting the remote-sensing model, 7 x 512, brings no table).
    ckpt: the loaded checkpoint dict, or a path."""
    if not isinstance(ckpt, dict):
        ckpt = torch.load(ckpt, map_location="cpu")
    table = ckpt.get("mphsir_clip_prompt")
    return table if table is not None and tuple(table.shape) == (task_classes, 512) else None


def load_warm_start(net, path, device, engine=None, resume=False):
    """The reference's warm start (train.py:109-116): keep checkpoint entries whose key AND shape match (keys carry the
    `net.` prefix), nothing else -- training starts at epoch 0 whatever wrote the checkpoint.  resume=True (--resume 1, a
    checkpoint written by save_checkpoint) also restores the optimizer state into `engine` and returns the epoch to
    continue at.  A saved CLIP table of the model's own shape must match the model's (a model built on other text
    embeddings would silently mis-evaluate); a table of another shape belongs to the other configuration and is ignored."""
    ckpt = path if isinstance(path, dict) else torch.load(path, map_location=device)      # main() loads the file once for both helpers
    state = ckpt["state_dict"]
    own = {"net." + k: v for k, v in net.state_dict().items()}
    kept = {k[4:]: v for k, v in state.items() if k in own and own[k].shape == v.shape}
    net.load_state_dict(kept, strict=False)
    table = ckpt.get("mphsir_clip_prompt")
    if table is not None and tuple(table.shape) == tuple(net.clip_prompts.shape) and \
            not torch.allclose(table.float().cpu(), net.clip_prompts.float().cpu(), atol=1e-5):
        raise RuntimeError("%s was trained with other CLIP text embeddings than this model was built with; build the model "
                           "with clip_prompt=ckpt['mphsir_clip_prompt']" % (path if not isinstance(path, dict) else "the checkpoint"))
    resume_epoch = 0
    if resume:
        if engine is None or "mphsir_optimizer" not in ckpt:
            raise RuntimeError("--resume 1 needs a checkpoint written by this script (optimizer state); %s has none"
                               % (path if not isinstance(path, dict) else "this one"))
        engine.load_optimizer_state(ckpt["mphsir_optimizer"])
        resume_epoch = int(ckpt.get("epoch", -1)) + 1
    return len(kept), resume_epoch


def save_checkpoint(path, net, engine, epoch):
    """Lightning-compatible layout (`state_dict` with the `net.` prefix, `epoch`) + what resuming needs."""
    torch.save({"state_dict": {"net." + k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, "epoch": epoch,
                "mphsir_optimizer": engine.optimizer_state(), "mphsir_clip_prompt": net.clip_prompts.detach().cpu().clone(),
                "mphsir_clip_source": net.text_prompt.clip_source}, path)


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
    clip_prompt = "surrogate" if opt.allow_surrogate_clip else None      # None: encode with OpenAI clip, or raise
    ckpt = None
    if opt.ckpt_path is not None:
        ckpt = torch.load(opt.ckpt_path, map_location="cpu")          # once: the CLIP table now, weights (+ optimizer) below
        saved = saved_clip_table(ckpt, cfg["task_classes"])
        clip_prompt = saved if saved is not None else clip_prompt
    dtypes = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}
    net = MP_HSIR_Net(**cfg, clip_prompt=clip_prompt, compute_dtype=dtypes[opt.precision]).to(dev).train()
    eng = DataParallelEngine(net, lr=opt.lr, use_graph=bool(opt.graph))
    start_epoch = 0
    if opt.ckpt_path is not None:
        n, start_epoch = load_warm_start(net, ckpt, dev, eng, resume=bool(opt.resume))
        if rank == 0:
            print("warm start: %d tensors from %s, %s at epoch %d" % (n, opt.ckpt_path, "resuming" if opt.resume else "starting", start_epoch))
            if not opt.resume and "mphsir_optimizer" in ckpt:
                print("note: %s also holds optimizer state of epoch %s; this run starts at epoch 0 with a fresh optimizer (the reference's "
                      "warm start, train.py:109-116) -- pass --resume 1 to continue that run instead" % (opt.ckpt_path, ckpt.get("epoch")))
        ckpt = None
        if start_epoch >= opt.epochs:
            raise SystemExit("--resume 1: %s already holds epoch %d of --epochs %d: nothing left to train" % (opt.ckpt_path, start_epoch - 1, opt.epochs))
    data_type = opt.model or opt.data_type
    de_types = opt.natural_scene_single_de_type if data_type == "natural_scene" else opt.remote_sensing_single_de_type
    if opt.synthetic:
        src = SyntheticPatchSource(cfg["in_channel"], opt.patch_size, opt.batch_size, cfg["task_classes"], dev, opt.seed, rank,
                                   de_types=de_types, data_type=data_type)
        steps_per_epoch = opt.steps_per_epoch
    else:
        if not opt.db_path:
            raise SystemExit("--synthetic 0 needs --db_path <directory with data.bin + meta_info.txt> (data.write_patch_db)")
        keep_all = data_type == "natural_scene" or opt.all_sources          # dataset_utils.py:56 filters remote-sensing sources
        db = PatchDB(opt.db_path, dataset_names=None if keep_all else REMOTE_SENSING_SOURCES)
        src = PatchDBSource(db, opt.batch_size, de_types, data_type, dev, opt.seed, rank, world, opt.repeat)
        steps_per_epoch = src.steps_per_epoch()
    for epoch in range(start_epoch, opt.epochs):
        lr = warmup_cosine_lr(epoch, opt.lr, opt.epochs)
        running = 0.0
        for it in range(steps_per_epoch):
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
            save_checkpoint(os.path.join(opt.ckpt_dir, "epoch=%d.ckpt" % epoch), net, eng, epoch)
    eng.finish()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
