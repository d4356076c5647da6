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
import os
import torch
import torch.distributed as dist

from . import ops


def l1_after_clamp(restored, clean):
    """reference training_step loss (train.py:58-61): clamp to [0,1] then nn.L1Loss -- one kernel for the loss and its gradient
    (the torch expression is 4 forward + ~10 backward elementwise launches)."""
    return ops.l1_clamp_loss(restored.float(), clean.float())


class PackPlan:
    """All kernel-layout weights of the model as ONE gather from the flat fp32 parameter arena.

    The per-module packers (net/MP_HSIR.py `packed()`, ops.pack_*) are pure data movement: cast, zero padding,
    transposes, value|gate row splits.  Run once with the arena holding its own indices (two fp32 passes: low 12 bits
    + 1, high bits) they yield, for every element of every packed tensor, the arena element it copies (or "padding").
    After that a step's ~500 cast/copy/fill launches are one `mphsir_pack_gather` launch per dtype into persistent
    buffers the module caches are pinned to.  Every map is verified bitwise against the packer's real output when the
    plan is built; a cache whose packer does anything else (or reads parameters outside the arena) is left unpinned."""

    def __init__(self, flat_p):
        self.flat = flat_p
        self.epoch = -1
        self.groups = {}            # dtype -> (index int32, persistent buffer)
        self.pinned = self.skipped = 0
        self.why = []               # (parameter shapes, reason) of every cache left unpinned

    def build(self):
        from torch.utils._pytree import tree_flatten, tree_map, tree_unflatten
        flat = self.flat
        lo_ptr, hi_ptr = flat.data_ptr(), flat.data_ptr() + 4 * flat.numel()

        def inside(t):
            return lo_ptr <= t.data_ptr() < hi_ptr
        caches = [c for c in ops.WeightCache.live() if c.last is not None and all(inside(p) for p in c.last[0])]
        with torch.no_grad():
            real = [c.last[2]() for c in caches]
            saved = flat.clone()
            ar = torch.arange(flat.numel(), device=flat.device)
            traces = []
            ops._TRACE[0] = True
            try:
                for code in ((ar & 0xFFF) + 1, ar >> 12):
                    flat.copy_(code.float())
                    # a no-op cast leaves a VIEW of the arena in the trace: detach it from the next overwrite
                    traces.append([tree_map(lambda t: t.clone() if isinstance(t, torch.Tensor) and inside(t) else t, c.last[2]())
                                   for c in caches])
            finally:
                ops._TRACE[0] = False
                flat.copy_(saved)
            todo = []                   # (cache, spec, leaves with placeholders, [(leaf slot, dtype, idx, shape)])
            for c, r, tlo, thi in zip(caches, real, traces[0], traces[1]):
                lr_, spec = tree_flatten(r)
                llo, lhi = tree_flatten(tlo)[0], tree_flatten(thi)[0]
                leaves, maps = list(lr_), []
                bad = None if len(llo) == len(lr_) == len(lhi) else "leaf count"
                for j, t in enumerate(lr_):
                    if bad:
                        break
                    if not isinstance(t, torch.Tensor) or inside(t):
                        continue                                    # parameter views need no copy at all
                    a, b = llo[j], lhi[j]
                    if not (t.is_contiguous() and a.shape == t.shape and b.shape == t.shape and t.dtype in ops._DT):
                        bad = "leaf %d: layout %s %s %s" % (j, tuple(t.shape), t.dtype, t.is_contiguous())
                        break
                    a, b = a.reshape(-1).double(), b.reshape(-1).double()
                    idx = torch.where(a > 0, b * 4096 + a - 1, -torch.ones_like(a)).long()
                    got = torch.where(idx >= 0, saved[idx.clamp(min=0)], torch.zeros((), device=flat.device)).to(t.dtype)
                    if not torch.equal(got, t.reshape(-1)):
                        bad = "leaf %d %s: not a gather of the arena (%d of %d elements differ)" % (
                            j, tuple(t.shape), int((got != t.reshape(-1)).sum()), got.numel())
                        break
                    maps.append((j, t.dtype, idx.int(), t.shape))
                if bad is None:
                    todo.append((c, spec, leaves, maps))
                    self.pinned += 1
                else:
                    self.skipped += 1
                    self.why.append(([tuple(p.shape) for p in c.last[0]][:3], bad))
            per = {}
            for _, _, _, maps in todo:
                for j, dt, idx, shape in maps:
                    lst, off = per.setdefault(dt, ([], [0]))
                    n = idx.numel()
                    pad = (-n) % 8
                    lst.append(idx if pad == 0 else torch.cat([idx, torch.full((pad,), -1, dtype=torch.int32, device=idx.device)]))
                    off.append(off[-1] + n + pad)
            self.groups = {dt: (torch.cat(lst), torch.empty(off[-1], dtype=dt, device=flat.device)) for dt, (lst, off) in per.items()}
            cursor = {dt: 0 for dt in per}
            for c, spec, leaves, maps in todo:
                for j, dt, idx, shape in maps:
                    n = idx.numel()
                    o = cursor[dt]
                    leaves[j] = self.groups[dt][1][o:o + n].view(shape)
                    cursor[dt] = o + n + (-n) % 8
                c.pinned = (c.last[1], self, tuple(p._version for p in c.last[0]), tree_unflatten(leaves, spec))
        self.refresh()
        return self

    def refresh(self):
        """one gather launch per dtype; call after every optimizer update of the arena (capturable)."""
        for dt, (idx, buf) in self.groups.items():
            ops.pack_gather(self.flat, idx, dt, out=buf)
        self.epoch = ops.weight_epoch()

    def release(self):
        for c in ops.WeightCache.live():
            if c.pinned is not None and c.pinned[1] is self:
                c.pinned = None


_EXT_EVENTS = {}


def _external_events_work(dev):
    """Does this PyTorch / HIP stack support external event record nodes in a captured graph, with a later
    stream.wait_event() honouring the record of the replay?  Probed once per device with a tiny graph whose event node
    follows a kernel writing a new value each replay; anything unexpected -> False (the engine then falls back to one
    all-reduce after the replay)."""
    key = str(dev)
    if key in _EXT_EVENTS:
        return _EXT_EVENTS[key]
    ok = False
    try:
        ev = torch.cuda.Event(external=True)
        src = torch.zeros(1 << 22, device=dev)
        val = torch.zeros((), device=dev)
        out = torch.zeros(8, device=dev)
        side = torch.cuda.Stream(dev)
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        stale = torch.zeros(8, device=dev)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            torch.cuda._sleep(20_000_000)       # ~10 ms in front of the record node: a wait that is NOT honoured reads the old value
            for _ in range(8):
                src.add_(1.0)
            val.copy_(src[0])
            ev.record()
            src.mul_(1.0)
        ok = True
        for i in range(6):
            g.replay()
            with torch.cuda.stream(side):
                stale[i].copy_(val)             # unsynchronised: must still see the previous replay's value (the probe can tell)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                out[i].copy_(val)
            torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        ok = out[:6].tolist() == [8.0 * (i + 1) for i in range(6)] and stale[:6].tolist() == [8.0 * i for i in range(6)]
    except Exception:
        ok = False
    _EXT_EVENTS[key] = ok
    return ok


class DataParallelEngine:
    def __init__(self, net, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, bucket_mb=32, process_group=None,
                 loss_fn=l1_after_clamp, use_graph=False, graph_warmup=2, use_pack_plan=True, loss_scaling="auto",
                 init_scale=65536.0, growth_interval=2000):
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
        self.use_graph, self.graph_warmup, self._graph, self._graph_key = use_graph, graph_warmup, None, None
        self.use_pack_plan, self.plan = use_pack_plan, None
        import os
        # graph mode, world > 1: bucket all-reduces start while the replay is still running (MPHSIR_GRAPH_OVERLAP=0: off)
        self.graph_overlap = os.environ.get("MPHSIR_GRAPH_OVERLAP", "1") != "0"
        self.force_eager = False        # diagnostics: run a graph-mode engine's step with eager launches (same data flow)
        # fp16 compute (the reference's precision="16-mixed", train.py:118) needs dynamic loss scaling: the loss is
        # multiplied by a device-resident scale before backward, the optimizer kernel divides it out again and skips
        # the step when a gradient overflowed, and the scale adapts (ops.scaled_adamw_step = torch GradScaler).
        # "auto": on exactly when the network computes in float16.
        self.loss_scaling, self.init_scale, self.growth_interval = loss_scaling, init_scale, growth_interval
        self.scaler = None
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
        # built here, outside any capture: its pinned tables cannot be allocated while a stream is capturing (a capture at the
        # very first hand-over -- graph_warmup=1 -- would otherwise be invalidated)
        self._multi_copy = ops.MultiCopy(self.flat_g.device, len(used)) if used else None
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
            self._capturing, self._bucket_order = False, []
            self._overlap = self.use_graph and self.graph_overlap and _external_events_work(dev)
            if not self.use_graph:
                for p in used:
                    p.register_post_accumulate_grad_hook(self._make_hook(bucket_of[p]))
            elif self._overlap:
                # graph mode: the all-reduce cannot live inside the captured step, but it can start while the replay is
                # still running.  While capturing, each bucket's gather is followed by an EXTERNAL event record node; after
                # graph.replay() the communication stream waits for bucket k's event and reduces it while the rest of
                # backward is still executing (DDP's overlap, re-expressed for a replayed hipGraph).
                self._bucket_events = [torch.cuda.Event(external=True) for _ in self.buckets]
                self._comm = torch.cuda.Stream(dev)
                for p in used:
                    p.register_post_accumulate_grad_hook(self._make_capture_hook(bucket_of[p]))

    def _gather_bucket(self, bi):
        """autograd hands every gradient over as a fresh tensor (p.grad was None): move the bucket's gradients into
        the arena with one multi-tensor copy instead of one accumulate kernel per parameter."""
        ops.flush_deferred()      # (only non-empty when a bucket hook fires in the middle of the backward pass)
        ps, views = self._bucket_members[bi]
        # _foreach_copy_ takes its multi-tensor kernel only when EVERY pair has identical dense strides; one transposed /
        # strided gradient view in the list silently turns the whole bucket into one copy launch per parameter (measured:
        # ~530 tiny copies, 1.6 ms of a 28.8 ms step).  So the list is split: contiguous gradients (all but a handful) go
        # through the multi-tensor kernel, the rest are copied one by one; a gradient that already is the arena view
        # needs no copy at all.
        fv, fg = [], []
        for p, v in zip(ps, views):
            g = p.grad
            if g is None or g.data_ptr() == v.data_ptr():
                continue
            if g.is_contiguous() and g.dtype == v.dtype:
                fv.append(v)
                fg.append(g)
            else:
                v.copy_(g)
        if fv:
            # ONE launch for the whole bucket (torch._foreach_copy_: 11 multi-tensor launches, 0.2 ms per step)
            self._multi_copy(fv, fg)
        for p in ps:
            p.grad = None

    def _make_capture_hook(self, bi):
        def hook(_p):
            if not self._capturing:
                return
            self._remaining[bi] -= 1
            if self._remaining[bi] == 0:
                self._gather_bucket(bi)
                self._bucket_events[bi].record()          # external event record node of the graph being captured
                self._bucket_order.append(bi)
        return hook

    def _make_hook(self, bi):
        def hook(_p):
            self._remaining[bi] -= 1
            if self._remaining[bi] == 0:
                self._gather_bucket(bi)
                s, e, _ = self.buckets[bi]
                self._pending.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))
        return hook

    # ---- one optimisation step ---------------------------------------------------------------------
    def _use_scaler(self, device):
        want = self.loss_scaling is True or (self.loss_scaling == "auto" and getattr(self.net, "_dtype", lambda: None)() == torch.float16)
        if want and self.scaler is None:
            self.scaler = ops.new_loss_scaler(device, self.init_scale)
        return want

    def _backward(self, loss):
        """loss.backward(), through the loss scale when the fp16 path is on"""
        with ops.deferred_reductions():      # the parameter-gradient sums of all backward functions in a few big launches at the end
            if self._use_scaler(loss.device):
                (loss * self.scaler[0]).backward()
            else:
                loss.backward()

    def _loss_backward(self, restored, clean):
        """loss + backward pass -> the (unscaled) loss.  With the reference's own loss (l1_after_clamp, train.py:58-61) the loss kernel's
        gradient starts the backward pass at `restored` directly: `loss.backward()` would first build a ones_like tensor and scale the
        gradient by it (two launches; with the fp16 loss scale three)."""
        if (self.loss_fn is l1_after_clamp and restored.dtype == torch.float32 and clean.dtype == torch.float32
                and restored.shape == clean.shape and restored.requires_grad):
            loss, g = ops.l1_clamp_loss_grad(restored.detach(), clean)
            if self._use_scaler(restored.device):
                g.mul_(self.scaler[0])
            with ops.deferred_reductions():
                restored.backward(g)
            return loss
        loss = self.loss_fn(restored, clean)
        self._backward(loss)
        return loss

    def _optimizer_step(self, lr, hyper=None):
        """AdamW over the arenas (1/world folded in); fp16 path: non-finite check + unscale + skip + scale update"""
        gs = 1.0 / self.world
        if self.scaler is not None:
            ops.scaled_adamw_step(self.flat_p, self.flat_g, self.flat_m, self.flat_v, self.scaler, 0.0 if hyper is not None else lr,
                                  self.betas[0], self.betas[1], self.eps, self.wd, gs, hyper=hyper, interval=self.growth_interval)
        elif hyper is not None:
            ops.flat_adamw(self.flat_p, self.flat_g, self.flat_m, self.flat_v, 0.0, 0, self.betas[0], self.betas[1], self.eps, self.wd,
                           gs, hyper=hyper)
        else:
            ops.flat_adamw(self.flat_p, self.flat_g, self.flat_m, self.flat_v, lr, self.step_count, self.betas[0], self.betas[1],
                           self.eps, self.wd, gs)

    def _hyper_values(self, lr, step):
        import math
        return [lr, 1.0 - self.betas[0] ** step, math.sqrt(1.0 - self.betas[1] ** step)]

    _HYPER_SLOTS = 4

    def _stage_hyper(self, values):
        """the step's host-computed scalars (lr, bias corrections) -> the device buffer the captured optimizer reads.  Through a ring
        of PINNED slots: a copy from pageable memory makes the host wait until the stream has drained -- every step would then start
        with an idle GPU waiting for the replay to be enqueued.  A slot is reused only after the copy that read it has finished
        (an event per slot), which also bounds how far the host runs ahead of the GPU (_HYPER_SLOTS steps)."""
        if os.environ.get("MPHSIR_HYPER_PAGEABLE", "0") == "1":
            self._hyper.copy_(torch.tensor(values), non_blocking=True)
            return
        ring = getattr(self, "_hyper_ring", None)
        if ring is None:
            ring = self._hyper_ring = (torch.empty((self._HYPER_SLOTS, len(values)), dtype=torch.float32).pin_memory(), [None] * self._HYPER_SLOTS)
        host, events = ring
        k = self.step_count % self._HYPER_SLOTS
        if events[k] is not None:
            events[k].synchronize()
        for i, v in enumerate(values):
            host[k, i] = float(v)
        self._hyper.copy_(host[k], non_blocking=True)
        events[k] = torch.cuda.Event()
        events[k].record()

    def _capture_key(self, degraded, clean, prompt):
        """everything the captured launch sequence depends on besides the buffer contents"""
        return (tuple(degraded.shape), degraded.dtype, tuple(clean.shape), clean.dtype, tuple(prompt.shape), prompt.dtype,
                self.net.training, getattr(self.net, "_dtype", lambda: None)())

    def _train_step_graph(self, degraded, clean, prompt, lr):
        dev = degraded.device
        key = self._capture_key(degraded, clean, prompt)
        if self._graph is not None and key != self._graph_key:
            # another batch shape / dtype / train-eval mode / compute dtype: the captured launches no longer apply.
            # Drop the graph and capture again (copy_ into the static buffers would otherwise broadcast or fail).
            self._graph = None
        if self._graph is None:
            self._graph_key = key
            self._sx, self._sc, self._sp = degraded.clone(), clean.clone(), prompt.clone()
            self._hyper = torch.zeros(3, dtype=torch.float32, device=dev)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                restored = self.net(self._sx, self._sp)
                overlap = self.world > 1 and getattr(self, "_overlap", False)
                if overlap:
                    self._remaining = [b[2] for b in self.buckets]
                    self._bucket_order, self._capturing = [], True
                try:
                    loss = self._loss_backward(restored, self._sc)
                finally:
                    self._capturing = False
                for bi in range(len(self.buckets)):
                    if not (overlap and bi in self._bucket_order):
                        self._gather_bucket(bi)
                if self.world == 1:
                    self._optimizer_step(None, hyper=self._hyper)
                    if self.plan is not None:
                        self.plan.refresh()
                self._sloss = loss.detach()
            self._graph = g
        assert self._sx.shape == degraded.shape and self._sc.shape == clean.shape and self._sp.shape == prompt.shape
        self._sx.copy_(degraded)
        self._sc.copy_(clean)
        self._sp.copy_(prompt)
        self.step_count += 1
        self._stage_hyper(self._hyper_values(self.lr if lr is None else lr, self.step_count))
        self._graph.replay()
        if self.world > 1:
            if getattr(self, "_overlap", False) and len(self._bucket_order) == len(self.buckets):
                handles, issued = [], set()
                try:
                    for bi in self._bucket_order:           # completion order of the captured backward
                        self._comm.wait_event(self._bucket_events[bi])
                        st, en, _ = self.buckets[bi]
                        with torch.cuda.stream(self._comm):
                            handles.append(dist.all_reduce(self.flat_g[st:en], group=self.pg, async_op=True))
                        issued.add(bi)
                    for h in handles:
                        h.wait()
                    torch.cuda.current_stream(dev).wait_stream(self._comm)
                except Exception as e:      # a stack that cannot do this: finish the step the plain way and stop trying
                    import warnings
                    warnings.warn("graph-mode all-reduce overlap disabled (%s: %s)" % (type(e).__name__, e))
                    self._overlap = False
                    for h in handles:
                        h.wait()
                    torch.cuda.synchronize(dev)
                    for bi in range(len(self.buckets)):
                        if bi not in issued:
                            st, en, _ = self.buckets[bi]
                            dist.all_reduce(self.flat_g[st:en], group=self.pg)
            else:
                dist.all_reduce(self.flat_g, group=self.pg)     # no external events on this stack: one message after the replay
            self._optimizer_step(None, hyper=self._hyper)
            if self.plan is not None:
                self.plan.refresh()
        return self._sloss

    # ---- optimizer state for checkpoints (the reference's Lightning checkpoints carry AdamW's moments and step) ----------
    def optimizer_state(self):
        """{'step', 'exp_avg': {param name: tensor}, 'exp_avg_sq': {...}, 'loss_scaler'} -- per parameter NAME, so it survives a
        different arena layout; parameters without gradient have no entry."""
        if self.arena is None:
            return {"step": self.step_count, "exp_avg": {}, "exp_avg_sq": {}}
        names = {id(p): n for n, p in self.net.named_parameters()}
        used, offs, _ = self.arena
        m = {names[id(p)]: self.flat_m[o:o + p.numel()].view(p.shape).detach().cpu().clone() for p, o in zip(used, offs)}
        v = {names[id(p)]: self.flat_v[o:o + p.numel()].view(p.shape).detach().cpu().clone() for p, o in zip(used, offs)}
        return {"step": self.step_count, "exp_avg": m, "exp_avg_sq": v,
                "loss_scaler": None if self.scaler is None else self.scaler.detach().cpu().clone()}

    def load_optimizer_state(self, state):
        """restore what optimizer_state() returned; call any time before or after the first step (it is applied as soon
        as the arenas exist)."""
        self._resume = state
        self.step_count = int(state["step"])
        dev = next(self.net.parameters()).device
        if self._use_scaler(dev):
            # the fp16 path counts its optimizer steps in scaler[3] (skipped steps do not advance the bias correction): adopt
            # the saved scaler, or -- a bf16 / fp32 checkpoint resumed in fp16 -- start a fresh one at the saved step count
            if state.get("loss_scaler") is not None:
                self.scaler = state["loss_scaler"].to(dev).float()
            else:
                self.scaler[3] = float(self.step_count)
        elif state.get("loss_scaler") is not None:
            self.step_count = int(state["loss_scaler"][3])       # an fp16 checkpoint resumed in bf16 / fp32: no scaler there
        if self.arena is not None:
            self._apply_resume()

    def _apply_resume(self):
        state, self._resume = getattr(self, "_resume", None), None
        if state is None:
            return
        names = {id(p): n for n, p in self.net.named_parameters()}
        used, offs, _ = self.arena
        for p, o in zip(used, offs):
            n = names[id(p)]
            if n in state["exp_avg"]:
                self.flat_m[o:o + p.numel()].copy_(state["exp_avg"][n].reshape(-1))
                self.flat_v[o:o + p.numel()].copy_(state["exp_avg_sq"][n].reshape(-1))

    def finish(self):
        """call before using the network eagerly again after graph-mode training (packed-weight caches were last
        refreshed inside the captured step, i.e. before its optimizer update)."""
        ops.bump_weight_epoch()

    def train_step(self, degraded, clean, prompt, lr=None):
        if self.use_pack_plan and self.plan is None and self.arena is not None:
            self.plan = PackPlan(self.flat_p).build()      # caches were populated by the step(s) before
        if (self.use_graph and not self.force_eager and self.arena is not None and self.step_count >= self.graph_warmup
                and degraded.is_cuda):
            return self._train_step_graph(degraded, clean, prompt, lr)
        first = self.arena is None
        if not first and self.world > 1 and not self.use_graph:
            self._remaining = [b[2] for b in self.buckets]
        restored = self.net(degraded, prompt)
        loss = self._loss_backward(restored, clean)
        if not first and (self.world == 1 or self.use_graph):
            for bi in range(len(self.buckets)):
                self._gather_bucket(bi)
            if self.world > 1:
                for s, e, _ in self.buckets:
                    self._pending.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))
        if first:
            self._build_arenas()
            self._apply_resume()
            if self.world > 1:
                for s, e, _ in self.buckets:
                    self._pending.append(dist.all_reduce(self.flat_g[s:e], group=self.pg, async_op=True))
        for h in self._pending:
            h.wait()
        self._pending = []
        self.step_count += 1
        self._optimizer_step(self.lr if lr is None else lr)
        if self.plan is not None:
            self.plan.refresh()
        return loss.detach()


class GraphedForward:
    """Inference through a replayed hipGraph: `net(x, prompt)` in eval mode is ~300 launches of which most are short, so
    at small batches the eager loop is host-bound (batch 16 and 32 take the same wall time).  The forward is captured
    once per (input shape, dtype, compute dtype) after `warmup` eager calls and replayed on private input / output
    buffers; the result is returned as a fresh tensor (the static output buffer is overwritten by the next call).
    Captures are tied to the weights they were taken with: the key carries the package weight epoch and every
    parameter's version counter, and entries of older weights (their graphs hold private memory pools -- GBs for
    512x512 cubes) are evicted as soon as the weights change."""

    def __init__(self, net, warmup=2):
        self.net, self.warmup = net, warmup
        self.entries = {}
        self._wkey = None

    def _weights_key(self):
        return (ops.weight_epoch(), tuple(p._version for p in self.net.parameters()), self.net.training,
                getattr(self.net, "_dtype", lambda: None)())

    @torch.no_grad()
    def __call__(self, x, prompt):
        if not x.is_cuda:
            return self.net(x, prompt)
        wkey = self._weights_key()
        if wkey != self._wkey:                 # load_state_dict / optimizer step / mode change: stale graphs go
            self.entries.clear()
            self._wkey = wkey
        key = (tuple(x.shape), x.dtype, tuple(prompt.shape), prompt.dtype)
        e = self.entries.get(key)
        if e is None:
            e = self.entries[key] = {"calls": 0}
        if "graph" not in e:
            e["calls"] += 1
            if e["calls"] <= self.warmup:
                return self.net(x, prompt)
            sx, sp = x.clone(), prompt.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self.net(sx, sp)
            e.update(graph=g, x=sx, p=sp, out=out)
        e["x"].copy_(x)
        e["p"].copy_(prompt)
        e["graph"].replay()
        return e["out"].clone()


def warmup_cosine_lr(epoch, base_lr, epochs, eta_min=1e-6):
    """LinearWarmupCosineAnnealingLR(warmup=int(0.1*epochs), max=epochs, eta_min=1e-6) stepped per epoch,
    as train.py:71-85 configures it (closed form, utils/schedulers.py:332-346).  lr(0) = 0."""
    import math
    wu = int(0.1 * epochs)
    if epoch < wu:
        return epoch * base_lr / (wu - 1) if wu > 1 else 0.0
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * (epoch - wu) / (epochs - wu)))
