"""Data-parallel training engine: one process per GPU, RCCL all-reduce of gradients over xGMI.

Re-expresses what the reference gets implicitly from pytorch-lightning (train.py:118
``pl.Trainer(devices=..., strategy="auto")`` -> DDP + AdamW, train.py:69) in an MI355X-first way:

  * parameters, gradients and the two Adam moments live in four flat fp32 arenas (nn.Parameters are
    views, so state_dict is untouched); 58 MB / 128 MB for the two shipped models -- trivial next to
    288 GB of HBM, so there is exactly one copy of each and no re-bucketing copies;
  * gradients are summed with `torch.distributed.all_reduce` (backend "nccl" = RCCL) directly on
    slices of the gradient arena, a few large buckets (default 32 MiB: a ring over 7 xGMI links is
    per-link bound, so few large messages beat many small ones), each issued from a
    post-accumulate-grad hook as soon as its last gradient is written, i.e. overlapped with the
    rest of backward; the arena is laid out in reverse registration order so buckets fill
    front-to-back during backward;
  * the 1/world mean, weight decay and the Adam update are one HIP kernel over the arena
    (mphsir_flat_adamw);
  * parameters that never receive a gradient (the reference has 8: TVSP.text_linear/clip_linear,
    SURVEY Q3) are detected on the first step and kept out of the arenas -- the reference's AdamW
    skips them too (grad is None), so no weight decay is applied to them.
"""
import torch
import torch.distributed as dist

from . import ops


def l1_after_clamp(restored, clean):
    """reference training_step loss (train.py:58-61): clamp to [0,1] then nn.L1Loss."""
    return (restored.clamp(0, 1) - clean).abs().mean()


class DataParallelEngine:
    def __init__(self, net, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, bucket_mb=32, process_group=None,
                 loss_fn=l1_after_clamp, use_graph=False, graph_warmup=2):
        self.net, self.lr, self.betas, self.eps, self.wd = net, lr, betas, eps, weight_decay
        self.loss_fn = loss_fn
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = int(bucket_mb * (1 << 20) // 4)
        self.step_count = 0
        self.arena = None
        self._pending = []
        # hipGraph mode: after `graph_warmup` eager steps the whole step (weight repack, forward, loss, backward,
        # gradient gather and -- on one GPU -- the AdamW kernel) is captured once and replayed: ~2000 launches per
        # step make the eager loop host-bound.  With world > 1 the all-reduce stays outside the graph (one
        # arena-wide reduction after the replay: 58 MB over xGMI is <1 ms next to a ~45 ms step).
        self.use_graph, self.graph_warmup, self._graph = use_graph, graph_warmup, None
        if self.world > 1:      # DDP's initial parameter broadcast (rank 0 -> all), one flat message
            ps = [p for p in net.parameters()]
            flat = torch.cat([p.data.reshape(-1).float() for p in ps])
            dist.broadcast(flat, 0, group=self.pg)
            o = 0
            for p in ps:
                p.data.copy_(flat[o:o + p.numel()].view(p.shape))
                o += p.numel()

    # ---- arenas ---------------------------------------------------------------------------------
    def _build_arenas(self):
        params = [p for p in self.net.parameters() if p.requires_grad]
        used = [p for p in params if p.grad is not None]
        self.unused = [p for p in params if p.grad is None]
        used = used[::-1]                                   # backward produces gradients roughly in reverse order
        offs, total = [], 0
        for p in used:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4               # keep every tensor 16-byte aligned
        dev = used[0].device
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.buckets = []                                   # (start, end, number of tensors)
        start, count = 0, 0
        for p, o in zip(used, offs):
            n = p.numel()
            self.flat_p[o:o + n].copy_(p.data.reshape(-1))
            self.flat_g[o:o + n].copy_(p.grad.reshape(-1))
            p.data = self.flat_p[o:o + n].view(p.shape)
            p.grad = None
            count += 1
            end = o + (n + 3) // 4 * 4
            if end - start >= self.bucket_elems:
                self.buckets.append([start, end, count])
                start, count = end, 0
        if count:
            self.buckets.append([start, total, count])
        self.arena = (used, offs, total)
        self._gviews = [self.flat_g[o:o + p.numel()].view(p.shape) for p, o in zip(used, offs)]
        self._bucket_members = []            # per bucket: (params, arena views)
        bi = 0
        cur_p, cur_v = [], []
        for p, o, gv in zip(used, offs, self._gviews):
            while o >= self.buckets[bi][1]:
                self._bucket_members.append((cur_p, cur_v))
                cur_p, cur_v = [], []
                bi += 1
            cur_p.append(p)
            cur_v.append(gv)
        self._bucket_members.append((cur_p, cur_v))
        if self.world > 1:
            bucket_of = {}
            bi = 0
            for p, o in zip(used, offs):
                while o >= self.buckets[bi][1]:
                    bi += 1
                bucket_of[p] = bi
            self._remaining = [b[2] for b in self.buckets]
            if not self.use_graph:
                for p in used:
                    p.register_post_accumulate_grad_hook(self._make_hook(bucket_of[p]))

    def _gather_bucket(self, bi):
        """autograd hands every gradient over as a fresh tensor (p.grad was None): move the bucket's gradients into
        the arena with one multi-tensor copy instead of one accumulate kernel per parameter."""
        ps, views = self._bucket_members[bi]
        torch._foreach_copy_(views, [p.grad for p in ps])
        for p in ps:
            p.grad = None

    def _make_hook(self, bi):
        def hook(_p):
            self._remaining[bi] -= 1
            if self._remaining[bi] == 0:
                self._gather_bucket(bi)
                s, e, _ = self.buckets[bi]
                self._pending.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))
        return hook

    # ---- one optimisation step ---------------------------------------------------------------------
    def _hyper_values(self, lr, step):
        import math
        return [lr, 1.0 - self.betas[0] ** step, math.sqrt(1.0 - self.betas[1] ** step)]

    def _train_step_graph(self, degraded, clean, prompt, lr):
        dev = degraded.device
        if self._graph is None:
            self._sx, self._sc, self._sp = degraded.clone(), clean.clone(), prompt.clone()
            self._hyper = torch.zeros(3, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                restored = self.net(self._sx, self._sp)
                loss = self.loss_fn(restored, self._sc)
                loss.backward()
                for bi in range(len(self.buckets)):
                    self._gather_bucket(bi)
                if self.world == 1:
                    ops.flat_adamw(self.flat_p, self.flat_g, self.flat_m, self.flat_v, 0.0, 0, self.betas[0], self.betas[1],
                                   self.eps, self.wd, 1.0, hyper=self._hyper)
                self._sloss = loss.detach()
            self._graph = g
        self._sx.copy_(degraded)
        self._sc.copy_(clean)
        self._sp.copy_(prompt)
        self.step_count += 1
        self._hyper.copy_(torch.tensor(self._hyper_values(self.lr if lr is None else lr, self.step_count)), non_blocking=True)
        self._graph.replay()
        if self.world > 1:
            for s, e, _ in self.buckets:
                dist.all_reduce(self.flat_g[s:e], group=self.pg)
            ops.flat_adamw(self.flat_p, self.flat_g, self.flat_m, self.flat_v, 0.0, 0, self.betas[0], self.betas[1], self.eps,
                           self.wd, 1.0 / self.world, hyper=self._hyper)
        return self._sloss

    def finish(self):
        """call before using the network eagerly again after graph-mode training (packed-weight caches were last
        refreshed inside the captured step, i.e. before its optimizer update)."""
        ops.bump_weight_epoch()

    def train_step(self, degraded, clean, prompt, lr=None):
        if self.use_graph and self.arena is not None and self.step_count >= self.graph_warmup and degraded.is_cuda:
            return self._train_step_graph(degraded, clean, prompt, lr)
        first = self.arena is None
        if not first and self.world > 1 and not self.use_graph:
            self._remaining = [b[2] for b in self.buckets]
        restored = self.net(degraded, prompt)
        loss = self.loss_fn(restored, clean)
        loss.backward()
        if not first and (self.world == 1 or self.use_graph):
            for bi in range(len(self.buckets)):
                self._gather_bucket(bi)
            if self.world > 1:
                for s, e, _ in self.buckets:
                    self._pending.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))
        if first:
            self._build_arenas()
            if self.world > 1:
                for s, e, _ in self.buckets:
                    self._pending.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))
        for h in self._pending:
            h.wait()
        self._pending = []
        self.step_count += 1
        ops.flat_adamw(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.lr if lr is None else lr, self.step_count,
                       self.betas[0], self.betas[1], self.eps, self.wd, 1.0 / self.world)
        return loss.detach()


def warmup_cosine_lr(epoch, base_lr, epochs, eta_min=1e-6):
    """LinearWarmupCosineAnnealingLR(warmup=int(0.1*epochs), max=epochs, eta_min=1e-6) stepped per epoch,
    as train.py:71-85 configures it (closed form, utils/schedulers.py:332-346).  lr(0) = 0."""
    import math
    wu = int(0.1 * epochs)
    if epoch < wu:
        return epoch * base_lr / (wu - 1) if wu > 1 else 0.0
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * (epoch - wu) / (epochs - wu)))
