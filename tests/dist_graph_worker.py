"""TEST INFRASTRUCTURE: one rank of a world_size-N data-parallel run of the tiny net that shares ONE GPU with the other
ranks (gloo process group over 127.0.0.1; the collectives stage through the host) -- or, with a 4th argument "nccl" on a box
with >= world GPUs, one GPU per rank over RCCL.  Exercises exactly the path
`bench.py --gpus N` / train.py take on a multi-GPU node -- broadcast at start, captured step (hipGraph), arena-wide
gradient all-reduce after the replay, fused AdamW + weight repack outside the graph -- so it can be checked on the
1-GPU test box.  Launched by tests/test_gpu_model.py; writes rank r's final parameters and losses to <out><r>."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out, mode, steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
    backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import model_checks as M
    from golden.cases import TINY_CFG
    from golden.detfill import seeded_input
    from mp_hsir_amd.engine import DataParallelEngine
    dev = torch.device("cuda", rank if backend == "nccl" else 0)
    net = M.build_net(TINY_CFG, dev, torch.float32)
    if rank != 0:                      # the engine must broadcast rank 0's parameters
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(1.5)
    eng = DataParallelEngine(net, lr=2e-3, use_graph=(mode == "graph"), graph_warmup=2, bucket_mb=0.25)
    losses = []
    for step in range(steps):
        xs = seeded_input("dp_x%d_%d" % (step, rank), (2, 8, 32, 32)).to(dev)
        cs = seeded_input("dp_c%d_%d" % (step, rank), (2, 8, 32, 32)).to(dev)
        task = torch.tensor([[rank + 1], [3]]).to(dev)
        losses.append(float(eng.train_step(xs, cs, task)))
    if mode == "graph":
        assert eng._graph is not None, "the captured step never ran"
        eng.finish()
    torch.cuda.synchronize()
    torch.save({"losses": losses, "state": {k: v.detach().cpu().clone() for k, v in net.state_dict().items()},
                "overlap": bool(getattr(eng, "_overlap", False)), "nbuckets": len(eng.buckets),
                "bucket_order": list(getattr(eng, "_bucket_order", []))}, out + str(rank))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
