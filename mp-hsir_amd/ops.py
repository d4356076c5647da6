"""Thin tensor-level wrappers over the C ABI (include/mphsir.h): shape checks, output allocation,
pointer/stream plumbing.  No arithmetic happens here."""
import ctypes
import os
import weakref

import torch

from . import _lib

_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}     # dtype codes of include/mphsir.h
_HALF = (torch.bfloat16, torch.float16)                            # 16-bit storage, fp32 accumulation

# optional algorithmic work accounting (bench.py roofline leg): kernel name -> [launches, flops, bytes]
ACCOUNT = None


def _acct(name, flops, nbytes):
    if ACCOUNT is not None:
        a = ACCOUNT.setdefault(name, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += flops
        a[2] += nbytes


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _check(*tensors):
    emu = _lib.is_emulated()
    for t in tensors:
        if t is None:
            continue
        if emu and t.is_cuda:
            raise RuntimeError("mp-hsir_amd: emulated test library bound but got a GPU tensor")
        if not emu and not t.is_cuda:
            raise RuntimeError("mp-hsir_amd: ops run on the GPU only (got a CPU tensor; there is no CPU fallback)")


# ---- side stream: small latency-bound kernels (64-256 workgroups) run beside the big streaming ones ------------------
_SIDE = {}
# Two users, measured separately on MI355X (graph replay, bench.py):
#  * the prompt gate (MPHSIR_SIDE_BRANCH, default ON): pg_gate_fwd / pg_gate_bwd run 128..2048 windows in 16-window
#    workgroups -- a few dozen to 128 workgroups for 20-45 us, most of the chip idle -- and nothing needs their output until
#    the branch sum / the window-attention backward.  Forked beside pass A (forward) / the channel-attention backward:
#    1159.5 -> 1182.4 patches/s.
#    (Tried and dropped: the fold backward and the weight-gradient GEMMs on side streams -- whole step unchanged, DESIGN.md.)
# Allocation stays stream-safe without record_stream on the inputs: tensors created inside the branch belong to the side
# stream's pool, every branch starts by waiting for the launch stream (a reused block is ordered after its last consumer
# there), and the caller holds the inputs until the join.
SIDE_BRANCH = os.environ.get("MPHSIR_SIDE_BRANCH", "1") == "1"
# the prompt modules of a pyramid level (TVSP + PromptFusion: prompt1 / fusion1 on e1, prompt2 / fusion2 on e2) feed the DECODER of that
# level only: issued on streams of their own they run beside the encoder / latent / decoder stages below them, whose launches leave
# part of the chip idle.  NO-GRAD FORWARD ONLY (512x512 forward 7.32 -> 6.97 ms, batch-16 forward 3.70 -> 3.38 ms).  In training the
# branches bought nothing (21.5 ms either way), and under capture together with the weight-gradient branch below they produced a wrong
# prompt1.text_prompt_learnable gradient twice (rounds 4 and 5: plain autograd ops on the prompt stream behind its last explicit join;
# one ordering hole was found and closed, a second configuration stayed 4 % off): round 6 removed the training switch
# (MPHSIR_PROMPT_SIDE_TRAIN) instead of shipping a knob known to yield wrong gradients -- the module forks them under no_grad only.
PROMPT_SIDE = os.environ.get("MPHSIR_PROMPT_SIDE", "1") == "1"
# (device, stream) of every TRACKED side stream forked since the backward pass began: the gradient hand-over waits for them (_dw_join).
# Only streams a backward pass can have work on are tracked (the prompt gate's); the no-grad prompt-module streams are not -- a captured
# training step must never wait on a stream that was last used outside its capture.
_SIDE_USED = set()


class side_stream:
    """`with ops.side_stream() as s:` launches onto a second HIP stream that first waits for everything already queued
    on the current one (fork); `s.join(*tensors)` makes the current stream wait for it and hands the tensors over
    (record_stream).  Works under hipGraph capture (fork/join from the capturing stream = parallel graph branches).
    No-op on CPU tensors (emulator) or when disabled."""

    def __init__(self, like, enabled=True, name="gate", track=True):
        self.on = enabled and like.is_cuda
        self.ctx = None
        self.track = track
        if self.on:
            dev = like.device
            self.main = torch.cuda.current_stream(dev)
            self.side = _SIDE.get((dev, name))         # one stream per user: a branch never queues behind another user's work
            if self.side is None:
                self.side = _SIDE[(dev, name)] = torch.cuda.Stream(dev)

    def __enter__(self):
        if self.on:
            self.side.wait_stream(self.main)
            if self.track:
                _SIDE_USED.add((self.main.device, self.side))
            self.ctx = torch.cuda.stream(self.side)
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.ctx.__exit__(*exc)
        return False

    def join(self, *tensors):
        if self.on:
            self.main.wait_stream(self.side)
            for t in tensors:
                if t is not None:
                    t.record_stream(self.main)


# ---- deferred ordered reduction of split partials (mphsir_reduce_parts) -----------------------------------------
_SCOPE = None


_DEFERRED = None      # list of segments while `deferred_reductions` is active (the engine's backward), else None


class reduce_scope:
    """Inside the scope every `reduce_parts` call only records its segment and returns the (not yet written) output
    tensor; one mphsir_reduce_parts launch per <= 32 segments fills them when the scope exits.  Callers may slice /
    reshape the outputs inside the scope but must not READ them (no kernels on them) before it exits.

    leaf=True declares that every sum of the scope is the gradient of a leaf parameter (nothing downstream of the backward
    function reads it): inside `deferred_reductions` those sums are then not launched at scope exit but handed to the
    enclosing context, which launches them together (see there)."""

    def __init__(self, leaf=False):
        self.leaf = leaf

    def __enter__(self):
        global _SCOPE
        self.prev, self.segs, self.gemms, self.calls = _SCOPE, [], [], []
        _SCOPE = self
        return self

    def __exit__(self, *exc):
        global _SCOPE
        _SCOPE = self.prev
        if exc[0] is None:
            if self.leaf and _DEFERRED is not None and DW_SIDE and _dw_side(self):
                pass                            # weight-gradient branch: issued beside the data-gradient chain (see _dw_side)
            elif self.leaf and _DEFERRED is not None:
                _flush_calls(self.calls)
                _flush_gemms(self.gemms)
                # ... their sums later, with everybody else's.  The output is kept alive through a detached alias: the tensor
                # object itself must stay uniquely referenced so that AccumulateGrad adopts it instead of cloning (= reading) it
                for g in self.segs:
                    g["keep"] = (g["keep"][0], g["keep"][1].detach())
                _poison(self.segs)
                _DEFERRED.extend(self.segs)
            else:
                _flush_calls(self.calls)
                _flush_gemms(self.gemms)        # the deferred weight-gradient GEMMs, grouped ...
                _flush(self.segs)               # ... then the ordered sums of everything they (and others) wrote
        self.segs, self.gemms, self.calls = [], [], []
        return False


# The parameter-gradient work of a backward function -- its grouped token-reduction GEMMs and the ordered sums of their partials --
# feeds nothing but the optimizer.  Inside the engine's backward (deferred_reductions) it can leave the launch stream: DW_SIDE = 1
# issues the sums, DW_SIDE = 2 the GEMMs and the sums, on a second stream forked at the end of the backward function and joined
# only where gradients are read (end of the backward pass / a bucket hook) -- a parallel branch of the captured graph that fills
# the CUs the small launches of the lower pyramid levels leave idle, reading the partials while they are still in the Infinity
# Cache.  Every tensor involved stays referenced until the join (nothing is recycled under the branch).
DW_SIDE = int(os.environ.get("MPHSIR_DW_SIDE", "2"))
_DW_STREAM = {}
_DW_KEEP = []


# MPHSIR_DEBUG_DEFERRED=1 (tests): every parameter-gradient sum that is handed out before it has been computed (deferred to the end
# of the backward pass, or issued on the weight-gradient branch) is first filled with NaN on the LAUNCH stream -- a consumer that
# reads or clones it early (a shared parameter, a tensor hook, gradient accumulation into an existing .grad, a strided view that
# AccumulateGrad has to copy) then yields NaN instead of stale memory, and the engine's finite check after the hand-over trips
DEBUG_DEFERRED = os.environ.get("MPHSIR_DEBUG_DEFERRED", "0") == "1"


def _poison(segs):
    if DEBUG_DEFERRED:
        for g in segs:
            g["keep"][1].fill_(float("nan"))


def _dw_side(scope):
    like = None
    for g in scope.gemms + scope.segs:
        like = g["keep"][0]
        break
    if like is None or not like.is_cuda:
        return False
    dev = like.device
    st = _DW_STREAM.get(dev)
    if st is None:
        st = _DW_STREAM[dev] = torch.cuda.Stream(dev)
    main = torch.cuda.current_stream(dev)
    for g in scope.segs:                # AccumulateGrad must find the output uniquely referenced: keep a detached alias
        g["keep"] = (g["keep"][0], g["keep"][1].detach())
    _poison(scope.segs)
    if DW_SIDE >= 2:
        # Small backward functions (the lower pyramid levels: a few MB of operands, launches that cost their 15-30 us floor whatever
        # they move) are BATCHED: their problems wait until DW_BATCH_BYTES of operands, a full grouped launch (8 problems) or 32 sum
        # segments have come together, and go out as ONE grouped GEMM launch + ONE ordered-sum launch; a big function flushes at once
        # (with whatever is pending, in issue order).  The operands stay referenced (pending list, then _DW_KEEP).
        nbytes = sum(g["M"] * (g["N1"] + g["N2"]) * 2 for g in scope.gemms)
        rows = max([g["M"] for g in scope.gemms] + [c["keep"][0].shape[0] for c in scope.calls if c["keep"][0].dim() == 2] + [0])
        if DW_DEFER and rows >= DW_DEFER_ROWS:
            # a BIG backward function (the full-resolution level: 131072 token rows at batch 32): its weight-gradient work is HELD, not
            # issued -- beside the equally big data-gradient kernels of the next block it only shares the CUs and HBM with them.  It goes
            # out when the backward pass reaches the lower pyramid levels (below), whose launches are latency-bound and leave most of the
            # chip idle: the branch then fills that idle time.  What the last big functions of a pass hold (encoder level 1) is issued
            # by the join.  Operands stay referenced (held list, then _DW_KEEP).
            _DW_HELD.setdefault(dev, []).append((scope.gemms, scope.calls, scope.segs))
            return True
        if DW_DEFER and rows > 0:
            _dw_release_held(dev)           # a small function: the held work of the big ones starts now, ahead of this function's own
        pend = _DW_PENDING.setdefault(dev, [[], [], [], 0])
        pend[0] += scope.gemms
        pend[1] += scope.calls
        pend[2] += scope.segs
        pend[3] += nbytes
        if scope.calls or pend[3] >= DW_BATCH_BYTES or len(pend[0]) + 3 > _lib.TN_GROUP_MAX or len(pend[2]) + 17 > _lib.REDUCE_MAX_SEGS:
            _dw_flush_pending(dev)
        return True
    else:
        _flush_calls(scope.calls)
        _flush_gemms(scope.gemms)
        st.wait_stream(main)
        for d2, s2 in list(_SIDE_USED):      # (as in _dw_flush_pending)
            if d2 == dev and s2 != st:
                st.wait_stream(s2)
        with torch.cuda.stream(st):
            _flush(scope.segs)
    _DW_KEEP.append((dev, [g["keep"] for g in scope.gemms] + [c["keep"] for c in scope.calls], [g["keep"] for g in scope.segs]))
    return True


DW_BATCH_BYTES = int(float(os.environ.get("MPHSIR_DW_BATCH_MB", "256")) * (1 << 20))      # 0 / 24 / 48 / 128 / 256 / 512 MB: 20.90 / 20.82 / 20.75 / 20.67 / 20.60 / 20.63 ms per step (one box, pairs)
_DW_PENDING = {}
# MPHSIR_DW_DEFER=1 (round-6 experiment, OFF): the weight-gradient work of the big (full-resolution) backward functions is held back until
# the pass reaches the lower pyramid levels, whose latency-bound launches leave most of the chip idle.  Measured SLOWER (20.31-20.35
# against 20.07-20.09 ms per step, two pairs on one box; with the 32768-row level held too: 20.45-20.50): beside a grouped GEMM that
# fills every CU the small launches of the main chain wait for workgroup slots -- the chain loses more than the branch hides.
DW_DEFER = os.environ.get("MPHSIR_DW_DEFER", "0") == "1"
DW_DEFER_ROWS = int(os.environ.get("MPHSIR_DW_DEFER_ROWS", "65536"))
_DW_HELD = {}
# the ordered sums of the weight-gradient branch: 1 = all of them at the join (the end of the backward pass / a bucket hook), as few
# launches of 32 segments as there can be, instead of one small launch behind every flush of the branch
DW_SUMS_LATE = os.environ.get("MPHSIR_DW_SUMS_LATE", "0") == "1"
_DW_SUMS = {}


def _dw_release_held(dev):
    """issue the held weight-gradient work of the big backward functions on the branch: function by function in the order they ran
    (a grouped GEMM launch + an ordered-sum launch each time DW_BATCH_BYTES of operands have come together)"""
    held = _DW_HELD.get(dev)
    if not held:
        return
    _dw_flush_pending(dev)                  # what was pending before goes first
    _DW_HELD[dev] = []
    for gemms, calls, segs in held:         # one batch per function: its sums follow its GEMMs while the partials are on the die
        _DW_PENDING[dev] = [list(gemms), list(calls), list(segs), 0]
        _dw_flush_pending(dev)


def _dw_flush_pending(dev):
    pend = _DW_PENDING.get(dev)
    if not pend or not (pend[0] or pend[1] or pend[2]):
        return
    gemms, calls, segs = pend[0], pend[1], pend[2]
    _DW_PENDING[dev] = [[], [], [], 0]
    st = _DW_STREAM[dev]
    st.wait_stream(torch.cuda.current_stream(dev))
    # a batch may hold problems whose operands were produced on another stream than the one that flushes it: the branch is ordered
    # behind every (tracked) side stream this pass has forked, not only the current one
    for d2, s2 in list(_SIDE_USED):
        if d2 == dev and s2 != st:
            st.wait_stream(s2)
    with torch.cuda.stream(st):
        _flush_calls(calls)
        _flush_gemms(gemms)
        if DW_SUMS_LATE:
            _DW_SUMS.setdefault(dev, []).extend(segs)      # ... summed at the join, 32 segments per launch
        else:
            _flush(segs)
    _DW_KEEP.append((dev, [g["keep"] for g in gemms] + [c["keep"] for c in calls], [g["keep"] for g in segs]))


def _dw_join(final=True):
    """the stream that reads gradients waits for the weight-gradient branch.  Only the FINAL join (the end of the backward pass, on
    the stream backward() was called from) releases the tensors and forgets the branch: a join in the middle -- a backward function
    that has to read a sum, possibly running on a prompt module's own stream -- makes ITS stream wait and leaves the rest as it is."""
    for dev in list(_DW_HELD):
        _dw_release_held(dev)
    for dev in list(_DW_PENDING):
        _dw_flush_pending(dev)
    for dev, segs in list(_DW_SUMS.items()):
        if segs:
            _DW_SUMS[dev] = []
            with torch.cuda.stream(_DW_STREAM[dev]):      # behind the GEMMs that wrote the partials (same stream)
                _flush(segs)
    if _DW_KEEP:
        for dev in {k[0] for k in _DW_KEEP}:
            torch.cuda.current_stream(dev).wait_stream(_DW_STREAM[dev])
        if final:
            del _DW_KEEP[:]
    # ... and for every other tracked side stream this backward pass has forked (the prompt gate's): work that autograd issued on one
    # of them behind its last explicit join is ordered before the hand-over here, not by luck
    for dev, st in list(_SIDE_USED):
        torch.cuda.current_stream(dev).wait_stream(st)
    if final:
        _SIDE_USED.clear()


class deferred_reductions:
    """`with ops.deferred_reductions():` around a whole backward pass: the partial-sum reductions of every reduce_scope(leaf=True)
    -- parameter gradients, which nothing reads before the optimizer / the gradient hand-over -- are collected and launched
    together when the context exits (or at `flush_deferred()`), 32 segments per launch, instead of one small latency-bound
    launch per backward function on the critical path between two backward functions (85 launches of ~15 us per training
    step).  Only the engine turns this on: it knows when gradients are read.  The partials stay allocated until then."""

    def __enter__(self):
        global _DEFERRED
        self.prev = _DEFERRED
        _DEFERRED = []
        if self.prev is None:
            _SIDE_USED.clear()      # streams forked before this backward pass (an earlier no-grad forward, another capture) are not its business
        return self

    def __exit__(self, *exc):
        global _DEFERRED
        segs, _DEFERRED = _DEFERRED, self.prev
        if exc[0] is None:
            _flush(segs)
        _dw_join()
        return False


def flush_deferred():
    """Launch what `deferred_reductions` has collected so far (the engine calls this before it reads gradients in the middle
    of a backward pass: the bucket hooks of the data-parallel path)."""
    if _DEFERRED:
        segs = _DEFERRED[:]
        del _DEFERRED[:]
        _flush(segs)
    _dw_join(final=False)


LOG_SEGS = os.environ.get("MPHSIR_LOG_SEGS", "0") == "1"      # diagnostics: print the problems / segments of every grouped launch


def _flush_calls(calls):
    """calls: deferred weight-gradient launches other than the grouped token-reduction GEMMs (dicts: `run` = the launch on the
    stream that is current NOW, `keep` = the tensors it reads / writes)."""
    for c in calls:
        c["run"]()


def _flush_gemms(gemms):
    """gemms: deferred bf16 token-reduction GEMMs (dicts of mphsir_gemm_tn_problem fields + tensors kept alive)."""
    lib = _lib.load()
    for i in range(0, len(gemms), _lib.TN_GROUP_MAX):
        chunk = gemms[i:i + _lib.TN_GROUP_MAX]
        # one launch = one element type on one device (a batch of the weight-gradient branch can mix backward functions: ADVICE r05)
        k0 = chunk[0]["keep"][0]
        assert all(g["keep"][0].dtype == k0.dtype and g["keep"][0].device == k0.device for g in chunk), "grouped token-reduction GEMMs of mixed dtype / device"
        arr = (_lib.TnProblem * len(chunk))()
        for k, g in enumerate(chunk):
            q = arr[k]
            q.A, q.lda, q.B, q.ldb, q.Cpart, q.colsum_part = g["A"], g["lda"], g["B"], g["ldb"], g["Cpart"], g["cs"]
            q.M, q.N1, q.N2, q.nsplit = g["M"], g["N1"], g["N2"], g["nsplit"]
        form = TN_FORM or (2 if all(g["form"] == 2 for g in chunk) else 1)
        if LOG_SEGS:       # diagnostics: (M, N1, N2, nsplit, column sums) per problem of the grouped launch
            print("gemm_tn_group form %d:" % form, [(g["M"], g["N1"], g["N2"], g["nsplit"], g["cs"] is not None) for g in chunk], flush=True)
        _lib.check(lib.mphsir_gemm_tn_group(arr, len(chunk), form, _DT[chunk[0]["keep"][0].dtype], _stream(chunk[0]["keep"][0])), "gemm_tn_group")


def _flush(segs):
    """segs: dicts with the fields of mphsir_reduce_seg + the tensors that keep the memory alive until the launch."""
    lib = _lib.load()
    for i in range(0, len(segs), _lib.REDUCE_MAX_SEGS):
        chunk = segs[i:i + _lib.REDUCE_MAX_SEGS]
        arr = (_lib.ReduceSeg * len(chunk))()
        nbytes = 0.0
        for k, g in enumerate(chunk):
            a = arr[k]
            a.src, a.dst = g["src"], g["dst"]
            a.n, a.stride, a.src_batch_stride, a.dst_batch_stride = g["n"], g["stride"], g["sbs"], g["dbs"]
            a.nsplit, a.nbatch, a.rows, a.dst_col_stride = g["nsplit"], g["nbatch"], g["rows"], g["dcs"]
            a.src_ld, a.dst_ld = g["src_ld"], g["dst_ld"]
            nbytes += 4.0 * g["nbatch"] * max(1, g["rows"]) * g["n"] * (g["nsplit"] + 1)
        _lib.check(lib.mphsir_reduce_parts(arr, len(chunk), _stream(chunk[0]["keep"][0])), "reduce_parts")
        _acct("reduce_parts", nbytes / 4.0, nbytes)
        if LOG_SEGS:       # diagnostics: one line per launch -- (n, rows, nsplit, nbatch, dst column stride, 16-byte aligned source) per segment
            print("reduce_parts %.1f MB:" % (nbytes / 1e6), [(g["n"], g["rows"], g["nsplit"], g["nbatch"], g["dcs"], g["src"] % 16 == 0 and g["stride"] % 4 == 0) for g in chunk], flush=True)


def _submit(seg, immediate):
    if _SCOPE is None or immediate:
        _flush([seg])
    else:
        _SCOPE.segs.append(seg)


def reduce_parts(part, batched=False, immediate=False):
    """part fp32 (nsplit, *shape) [batched: (Bt, nsplit, *shape)], each partial contiguous (the split / batch axes may be strided:
    the slots an in-kernel group sum leaves) -> sum over the split axis in a fixed order (deterministic), shape (*shape)
    [(Bt, *shape)].  Deferred to the end of the enclosing `reduce_scope`, if any."""
    _check(part)
    assert part.dtype == torch.float32
    if batched:
        Bt, nsplit, shape = part.shape[0], part.shape[1], tuple(part.shape[2:])
        sbs, sst = part.stride(0), part.stride(1)
    else:
        Bt, nsplit, shape = 1, part.shape[0], tuple(part.shape[1:])
        sbs, sst = 0, part.stride(0)
    n = 1
    for d in shape:
        n *= d
    assert part[0, 0].is_contiguous() if batched else part[0].is_contiguous()
    if nsplit == 1 and (not batched or Bt == 1 or sbs == n):
        return part[:, 0] if batched else part[0]
    # (a single strided slot per batch entry -- the in-kernel group sum of <= 8 splits -- still goes through the launch: callers get
    # a contiguous tensor)
    out = torch.empty(((Bt,) + shape) if batched else shape, dtype=torch.float32, device=part.device)
    _submit(dict(src=part.data_ptr(), dst=out.data_ptr(), n=n, stride=sst, sbs=sbs, dbs=n, nsplit=nsplit, nbatch=Bt,
                 rows=1, dcs=1, src_ld=0, dst_ld=0, keep=(part, out)), immediate)
    return out


def reduce_block(part, r0, nr, c0, nc, out, transpose=False, immediate=False):
    """Sum the (r0:r0+nr, c0:c0+nc) sub-block of the partial matrices part (nsplit, R, Cc) over the split axis straight
    into the 2-D fp32 view `out`: shape (nr, nc) with unit column stride, or -- transpose=True -- shape (nc, nr).  This is
    where padded hidden rows are dropped, sub-blocks of a factor product are cut out and tap gradients are transposed,
    instead of in separate copy / cat launches.  Deferred like reduce_parts."""
    _check(part, out)
    assert part.dtype == torch.float32 and part.dim() == 3 and part[0].is_contiguous() and out.dtype == torch.float32
    nsplit, R, Cc = part.shape
    assert 0 <= r0 and r0 + nr <= R and 0 <= c0 and c0 + nc <= Cc and out.dim() == 2
    if transpose:
        assert tuple(out.shape) == (nc, nr) and out.stride(1) == 1
        dst_ld, dcs = 1, out.stride(0)
    else:
        assert tuple(out.shape) == (nr, nc) and out.stride(1) == 1
        dst_ld, dcs = out.stride(0), 1
    _submit(dict(src=part.data_ptr() + 4 * (r0 * Cc + c0), dst=out.data_ptr(), n=nc, stride=part.stride(0), sbs=0, dbs=0, nsplit=nsplit,
                 nbatch=1, rows=max(nr, 1), dcs=dcs, src_ld=Cc, dst_ld=dst_ld, keep=(part, out)), immediate)
    if nr == 1:      # the C side treats rows <= 1 as the 1-D form: identical addressing for a single row
        pass
    return out


def pack_gather(arena, index, dtype, out=None):
    """out[i] = arena[index[i]] cast to dtype, 0 where index[i] < 0 (index int32, len % 8 == 0)."""
    lib = _lib.load()
    _check(arena, index, out)
    assert arena.dtype == torch.float32 and index.dtype == torch.int32 and arena.is_contiguous() and index.is_contiguous()
    n = index.numel()
    if out is None:
        out = torch.empty((n,), dtype=dtype, device=arena.device)
    assert out.numel() == n and out.dtype == dtype and out.is_contiguous()
    _lib.check(lib.mphsir_pack_gather(_p(arena), _p(index), _p(out), n, _DT[dtype], _stream(arena)), "pack_gather")
    _acct("pack_gather", 0.0, n * (8.0 + out.element_size()))
    return out


class MultiCopy:
    """dsts[i].copy_(srcs[i]) for lists of contiguous fp32 tensors in ONE launch (mphsir_multi_copy): the table of pointers is
    written into a pinned host buffer (a small ring of them: an eager caller may run a step ahead of the GPU) and copied to the
    device on the launch stream -- under hipGraph capture that copy is a node that re-reads the same pinned rows on every replay,
    and the captured pointers stay valid because a graph's allocations live in its private pool."""
    BLOCK = 4096

    def __init__(self, device, capacity, ring=4):
        self.device, self.capacity, self.ring = device, capacity, ring
        self.host = [torch.empty((capacity, 4), dtype=torch.int64).pin_memory() if device.type == "cuda" else torch.empty((capacity, 4), dtype=torch.int64)
                     for _ in range(ring)]
        self.dev = [torch.empty((capacity, 4), dtype=torch.int64, device=device) for _ in range(ring)]
        self.events = [None] * ring
        self.turn = 0
        # table pairs for calls made under hipGraph capture: a captured copy node re-reads its pinned rows on every replay, so such a
        # pair is never reused -- and pinned memory cannot be allocated while a stream is capturing, so they are set aside here
        pin = device.type == "cuda"
        self.for_capture = [(torch.empty((capacity, 4), dtype=torch.int64).pin_memory() if pin else torch.empty((capacity, 4), dtype=torch.int64),
                             torch.empty((capacity, 4), dtype=torch.int64, device=device)) for _ in range(24)]
        self.captured = []

    def __call__(self, dsts, srcs):
        n = len(dsts)
        if n == 0:
            return
        assert n <= self.capacity
        capturing = self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if capturing:
            if not self.for_capture:          # more captured hand-overs than tables set aside: the framework's multi-tensor copy
                torch._foreach_copy_(list(dsts), list(srcs))
                return
            h, dv = self.for_capture.pop()
            self.captured.append((h, dv))
        else:
            k = self.turn
            self.turn = (k + 1) % self.ring
            if self.events[k] is not None:
                self.events[k].synchronize()                  # the launch that last read this pinned buffer has fetched it
            h, dv = self.host[k], self.dev[k]
        rows, blk = [], 0
        for d, s_ in zip(dsts, srcs):
            ne = d.numel()
            rows.append((s_.data_ptr(), d.data_ptr(), ne, blk))
            blk += (ne + self.BLOCK - 1) // self.BLOCK
        h[:n] = torch.tensor(rows, dtype=torch.int64)
        dv[:n].copy_(h[:n], non_blocking=True)
        _lib.check(_lib.load().mphsir_multi_copy(_p(dv), n, blk, _stream(dv)), "multi_copy")
        if self.device.type == "cuda" and not capturing:
            self.events[k] = torch.cuda.Event()
            self.events[k].record()
        _acct("multi_copy", 0.0, 8.0 * sum(d.numel() for d in dsts))


class _L1ClampLoss(torch.autograd.Function):
    """mean |clamp(y, 0, 1) - clean| with its gradient from the same pass (mphsir_l1_clamp_loss; train.py:58-61)"""

    @staticmethod
    def forward(ctx, y, clean):
        lib = _lib.load()
        _check(y, clean)
        y, clean = y.contiguous(), clean.contiguous()
        assert y.dtype == torch.float32 and clean.dtype == torch.float32 and y.shape == clean.shape
        n = y.numel()
        nblk = max(1, min(1024, (n + 1023) // 1024))
        need = ctx.needs_input_grad[0]
        g = torch.empty_like(y) if need else None
        part = torch.empty((nblk, 1), dtype=torch.float32, device=y.device)
        _lib.check(lib.mphsir_l1_clamp_loss(_p(y), _p(clean), _p(g), _p(part), n, nblk, _stream(y)), "l1_clamp_loss")
        _acct("l1_clamp_loss", 4.0 * n, 4.0 * n * (3 if need else 2))
        ctx.g = g
        return reduce_parts(part, immediate=True).reshape(())

    @staticmethod
    def backward(ctx, dloss):
        g, ctx.g = ctx.g, None
        assert g is not None, "l1_clamp_loss: its gradient buffer is scaled in place and handed over: backward() runs once"
        return g.mul_(dloss), None


def l1_clamp_loss(y, clean):
    return _L1ClampLoss.apply(y, clean)


def l1_clamp_loss_grad(y, clean):
    """(mean |clamp(y, 0, 1) - clean|, its gradient w.r.t. y) from one pass, outside autograd: for a caller that starts the backward pass
    at y itself (engine.DataParallelEngine: `y.backward(g)` instead of `loss.backward()`, whose ones_like + scale launches it saves)"""
    lib = _lib.load()
    _check(y, clean)
    y, clean = y.contiguous(), clean.contiguous()
    assert y.dtype == torch.float32 and clean.dtype == torch.float32 and y.shape == clean.shape
    n = y.numel()
    nblk = max(1, min(1024, (n + 1023) // 1024))
    g = torch.empty_like(y)
    part = torch.empty((nblk, 1), dtype=torch.float32, device=y.device)
    _lib.check(lib.mphsir_l1_clamp_loss(_p(y), _p(clean), _p(g), _p(part), n, nblk, _stream(y)), "l1_clamp_loss")
    _acct("l1_clamp_loss", 4.0 * n, 12.0 * n)
    return reduce_parts(part, immediate=True).reshape(()), g


def _rows(t):
    """(rows, ld) of a 2-D view whose last dim is contiguous."""
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D view"
    return t.shape[0], t.stride(0)


# token GEMM kernel form (include/mphsir.h, mphsir_gemm_args.form): 0 = the library chooses (ring form for input-heavy shapes, K >= 3 N),
# 1 = one workgroup per token tile, 2 = ring form wherever it applies
TOK_FORM = int(os.environ.get("MPHSIR_TOK_FORM", "0"))


def gemm_tok(x, w, bias=None, ln=None, epi=0, res=None, sa=None, gate=None, keep=None, geom=None, out=None):
    """Y = epi(pro(X) @ W^T).  x (M,K) row-major view; w (N,K) or per-sample (B,N,K) in x.dtype.
    ln = (weight, bias) fp32 -> LayerNorm prologue.  epi 0/1/2 as in include/mphsir.h.
    geom = (H, W, shift) for epi 2.  Returns (M,N)."""
    lib = _lib.load()
    _check(x, w, bias, res, sa, gate, keep)
    M, ldx = _rows(x)
    K = x.shape[1]
    per_sample = w.dim() == 3
    N = w.shape[-2]
    assert w.shape[-1] == K and w.is_contiguous() and w.dtype == x.dtype, (w.shape, K, w.dtype, x.dtype)
    y = out if out is not None else torch.empty((M, N), dtype=x.dtype, device=x.device)
    a = _lib.GemmArgs()
    a.X, a.ldx = _p(x), ldx
    a.W = _p(w)
    a.w_batch_stride = N * K if per_sample else 0
    a.rows_per_batch = M // w.shape[0] if per_sample else 0
    a.bias = _p(bias)
    if ln is not None:
        a.ln_w, a.ln_b = _p(ln[0]), _p(ln[1])
    a.Y, a.ldy = _p(y), _rows(y)[1]
    a.M, a.N, a.K, a.epi = M, N, K, epi
    if res is not None:
        a.R, a.ldr = _p(res), _rows(res)[1]
    if sa is not None:
        a.SA, a.ldsa = _p(sa), _rows(sa)[1]
    a.gate, a.keep = _p(gate), _p(keep)
    if geom is not None:
        a.H, a.Wimg, a.shift = geom
    a.form = TOK_FORM
    _lib.check(lib.mphsir_gemm_tok(ctypes.byref(a), _DT[x.dtype], _stream(x)), "gemm_tok")
    es = x.element_size()
    _acct("gemm_tok", 2.0 * M * N * K, (M * K + M * N * (1 + (res is not None) + (sa is not None))) * es + w.numel() * es)
    return y





def layernorm_tok(x, ln_w, ln_b, out_dtype, want_cast=False):
    """x (M,C) contiguous, fp32 or compute dtype -> LN(x) in out_dtype (statistics in fp32); want_cast: -> (LN(x), x cast to out_dtype)."""
    lib = _lib.load()
    _check(x, ln_w, ln_b)
    M, C = x.shape
    assert x.is_contiguous()
    y = torch.empty((M, C), dtype=out_dtype, device=x.device)
    xc = torch.empty((M, C), dtype=out_dtype, device=x.device) if want_cast else None
    _lib.check(lib.mphsir_layernorm_tok(_p(x), _DT[x.dtype], _p(ln_w), _p(ln_b), _p(y), _p(xc), _DT[out_dtype], M, C, _stream(x)), "layernorm_tok")
    _acct("layernorm_tok", 8.0 * M * C, M * C * (x.element_size() + y.element_size() * (2 if want_cast else 1)))
    return (y, xc) if want_cast else y


def tvsp_text_map(L, clip, ps):
    """L (B,D) fp32, clip (B,512) fp32 -> text (B,ps,ps,D) fp32 (TVSP.forward :575-577 incl. its batch-on-rows broadcast)."""
    lib = _lib.load()
    _check(L, clip)
    B, D = L.shape
    assert L.dtype == torch.float32 and clip.dtype == torch.float32 and clip.shape == (B, 512) and L.is_contiguous() and clip.is_contiguous()
    text = torch.empty((B, ps, ps, D), dtype=torch.float32, device=L.device)
    _lib.check(lib.mphsir_tvsp_text_map(_p(L), _p(clip), _p(text), B, ps, D, _stream(L)), "tvsp_text_map")
    _acct("resample", 1.0 * text.numel(), 4.0 * text.numel())
    return text


def tvsp_text_map_bwd(dtext, clip):
    """dtext (B,ps,ps,D) fp32 -> dL (B,D) fp32"""
    lib = _lib.load()
    _check(dtext, clip)
    B, ps, _, D = dtext.shape
    assert dtext.dtype == torch.float32 and dtext.is_contiguous() and clip.is_contiguous()
    part = torch.empty((B, ps, D), dtype=torch.float32, device=dtext.device)
    _lib.check(lib.mphsir_tvsp_text_map_bwd(_p(dtext), _p(clip), _p(part), B, ps, D, _stream(dtext)), "tvsp_text_map_bwd")
    _acct("resample", 2.0 * dtext.numel(), 4.0 * dtext.numel())
    return reduce_parts(part, batched=True, immediate=True)


def resize_bilinear(x, H, W, backward=False):
    """x (B,h,w,C) channels-last -> (B,H,W,C) bilinear, align_corners=False; backward=True: x is dY (B,hh,ww,C) of a forward
    to (hh,ww) from (H,W) and the result is dX (B,H,W,C)."""
    lib = _lib.load()
    _check(x)
    B, h, w, C = x.shape
    assert x.is_contiguous()
    y = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    if backward:
        _lib.check(lib.mphsir_resize_bilinear(_p(x), _p(y), B, H, W, h, w, C, 1, _DT[x.dtype], _stream(x)), "resize_bilinear")
    else:
        _lib.check(lib.mphsir_resize_bilinear(_p(x), _p(y), B, h, w, H, W, C, 0, _DT[x.dtype], _stream(x)), "resize_bilinear")
    _acct("resample", 8.0 * y.numel(), (x.numel() + y.numel()) * x.element_size())
    return y


def nchw_to_cl(x, dtype, cp):
    """x (B,C,H,W) fp32 -> (B,H,W,cp) channels-last in `dtype`, channels C..cp-1 zero: `inp_img` as the patch embedding reads it,
    and the gradient of the output head (one launch for .to / permute / contiguous / pad)."""
    lib = _lib.load()
    _check(x)
    B, C, H, W = x.shape
    assert x.is_contiguous() and x.dtype == torch.float32 and cp >= C
    y = torch.empty((B, H, W, cp), dtype=dtype, device=x.device)
    _lib.check(lib.mphsir_nchw_to_cl(_p(x), _p(y), B, C, H * W, cp, _DT[dtype], _stream(x)), "nchw_to_cl")
    _acct("layout", 0.0, x.numel() * 4.0 + y.numel() * y.element_size())
    return y


def cl_to_nchw_add(y, C, res=None):
    """y (B,H,W,Cy) channels-last, Cy >= C -> (B,C,H,W) fp32 = float(y[..., :C]) + res: the output head `self.output(...) + inp_img`
    (one launch for slice / permute / .to / add)."""
    lib = _lib.load()
    _check(y, res)
    B, H, W, Cy = y.shape
    assert y.is_contiguous() and Cy >= C
    if res is not None:
        assert res.shape == (B, C, H, W) and res.dtype == torch.float32 and res.is_contiguous()
    o = torch.empty((B, C, H, W), dtype=torch.float32, device=y.device)
    _lib.check(lib.mphsir_cl_to_nchw_add(_p(y), Cy, _p(res), _p(o), B, C, H * W, _DT[y.dtype], _stream(y)), "cl_to_nchw_add")
    _acct("layout", 0.0, o.numel() * (8.0 if res is not None else 4.0) + B * H * W * C * y.element_size())
    return o


def task_weights(ids, T):
    """ids (B,n) int64 task ids -> (B,T) fp32: the mean of each sample's one-hot rows (Text_Prompt.forward's training path)"""
    lib = _lib.load()
    _check(ids)
    assert ids.dim() == 2 and ids.dtype == torch.int64 and ids.is_contiguous()
    B, n = ids.shape
    w = torch.empty((B, T), dtype=torch.float32, device=ids.device)
    _lib.check(lib.mphsir_task_weights(_p(ids), _p(w), B, n, T, _stream(ids)), "task_weights")
    return w


def mix_rows(A, Bm, scale, transA=False):
    """scale * A @ Bm in fp32 for the few-row matrices of the task prompts: A (I,J) -- or stored (J,I) with transA -- , Bm (J,D)"""
    lib = _lib.load()
    _check(A, Bm)
    assert A.dtype == Bm.dtype == torch.float32 and A.is_contiguous() and Bm.is_contiguous() and A.dim() == Bm.dim() == 2
    I, J = (A.shape[1], A.shape[0]) if transA else A.shape
    assert Bm.shape[0] == J
    D = Bm.shape[1]
    o = torch.empty((I, D), dtype=torch.float32, device=A.device)
    _lib.check(lib.mphsir_mix_rows(_p(A), _p(Bm), _p(o), I, J, D, float(scale), 1 if transA else 0, _stream(A)), "mix_rows")
    return o


def round_up(n, m):
    return (n + m - 1) // m * m


def pack_gated_mlp(fc1_w, fc1_b, fc2_w, dtype):
    """(W1 [2*HP][C], b1 [2*HP] fp32, W2 [C][HP]) in the padded layout mphsir_gated_mlp_fwd reads."""
    two_hid, C = fc1_w.shape
    hid = two_hid // 2
    HP = round_up(hid, 32)
    W1 = torch.zeros((2 * HP, C), dtype=cdt(dtype), device=fc1_w.device)
    W1[:hid] = fc1_w[:hid].to(cdt(dtype))
    W1[HP:HP + hid] = fc1_w[hid:].to(cdt(dtype))
    b1 = torch.zeros((2 * HP,), dtype=torch.float32, device=fc1_w.device)
    b1[:hid] = fc1_b[:hid]
    b1[HP:HP + hid] = fc1_b[hid:]
    W2 = torch.zeros((C, HP), dtype=cdt(dtype), device=fc1_w.device)
    W2[:, :hid] = fc2_w.to(cdt(dtype))
    return W1, b1, W2




BASE_SKIP_FUSED = os.environ.get("MPHSIR_BASE_SKIP_FUSED", "1") == "1"     # BaseBlock's `+ x` inside its last gated-MLP launch (6 launches fewer per forward)
BASE_SKIP_TRAIN = os.environ.get("MPHSIR_BASE_SKIP_TRAIN", "1") == "1"     # ... in training passes too
BASE_SKIP_BWD = os.environ.get("MPHSIR_BASE_SKIP_BWD", "1") == "1"         # ... and the skip's gradient added by the first block's last backward launch
# hidden split of the gated MLP kernels for small launches (< 256 token tiles at C >= 192: the latent level): 0 = off, else the number of
# workgroups per token tile is chosen so that about 256 workgroups exist
MLP_HSPLIT = os.environ.get("MPHSIR_MLP_HSPLIT", "1") == "1"


def mlp_hsplit(M, C, HP):
    if not MLP_HSPLIT or C < 192 or M // 64 >= 256:
        return 1
    s, chunks = 1, HP // 32
    while (M // 64) * s * 2 <= 256 and chunks % (s * 2) == 0:
        s *= 2
    return s


MLP_FUSE_SUM = os.environ.get("MPHSIR_MLP_FUSE_SUM", "1") == "1"      # the block's branch sum (pass B + gate + residual) inside the gated-MLP launch
MLP_FUSE_MIN_TILES = int(os.environ.get("MPHSIR_MLP_FUSE_MIN_TILES", "256"))   # ... from this many 128-token tiles (where the plain launch is the eight-wave form too)


def gated_mlp_fuses(M, C, HW, dtype):
    """the gated-MLP forward can form its own input y = x + keep1 * (sa * gate + v Mb^T) (the launch of gemm_tok epi 2 disappears)"""
    return (MLP_FUSE_SUM and dtype in _HALF and HW % 128 == 0 and M // 128 >= MLP_FUSE_MIN_TILES
            and bool(_lib.load().mphsir_gated_mlp_fwd_fuses(C, M, _DT[dtype])))


def gated_mlp_fwd(x, ln_w, ln_b, W1, b1, W2, b2, keep=None, rows_per_batch=0, out=None, tiles_per_wave=0, hsplit=None, res=None, branch=None):
    """x (M,C) row-major view -> x + keep * mlp(LN(x)) [+ res, a second residual (M,C) row-major view: the skip of a whole BaseBlock
    folded into its last block's launch]; weights from pack_gated_mlp.
    branch = dict(v, Mb, sa, gate, keep, geom=(H, W, shift), want_y): the FUSED branch sum -- the kernel first forms
    y = x + keep1 * (sa * gate[window] + v Mb^T) (what gemm_tok(v, Mb, epi=2, res=x, ...) returns, bitwise) and runs on it; returns
    (z, y) with y None unless want_y."""
    lib = _lib.load()
    _check(x, W1, W2, b1, b2, ln_w, ln_b, keep, res)
    M, ldx = _rows(x)
    C = x.shape[1]
    HP = W2.shape[1]
    assert W1.shape == (2 * HP, C) and W1.dtype == x.dtype and W2.dtype == x.dtype
    y = out if out is not None else torch.empty((M, C), dtype=x.dtype, device=x.device)
    a = _lib.MlpArgs()
    a.X, a.ldx, a.ln_w, a.ln_b = _p(x), ldx, _p(ln_w), _p(ln_b)
    a.W1, a.b1, a.W2, a.b2 = _p(W1), _p(b1), _p(W2), _p(b2)
    a.keep, a.rows_per_batch = _p(keep), rows_per_batch
    if res is not None:
        assert res.shape == x.shape and res.dtype == x.dtype
        a.R, a.ldr = _p(res), _rows(res)[1]
    a.Y, a.ldy, a.M, a.C, a.HP, a.tiles_per_wave = _p(y), _rows(y)[1], M, C, HP, tiles_per_wave
    yb = None
    if branch is not None:
        v, Mb, sa, gate = branch["v"], branch["Mb"], branch["sa"], branch["gate"]
        H, W, shift = branch["geom"]
        _check(v, Mb, sa, gate, branch.get("keep"))
        assert x.dtype in _HALF and (H * W) % 128 == 0 and lib.mphsir_gated_mlp_fwd_fuses(C, M, _DT[x.dtype]) and Mb.dim() == 3 and Mb.shape[1:] == (C, C) and Mb.is_contiguous() and M == Mb.shape[0] * H * W
        assert v.dtype == x.dtype and sa.dtype == x.dtype and gate.dtype == torch.float32 and gate.is_contiguous() and hsplit in (None, 1)
        a.PV, a.ldpv, a.PM, a.pm_batch_stride = _p(v), _rows(v)[1], _p(Mb), C * C
        a.PSA, a.ldpsa, a.pgate, a.pkeep = _p(sa), _rows(sa)[1], _p(gate), _p(branch.get("keep"))
        a.H, a.Wimg, a.shift = H, W, shift
        if branch.get("want_y"):
            yb = torch.empty((M, C), dtype=x.dtype, device=x.device)
            a.Yb, a.ldyb = _p(yb), C
        hsplit = 1
    if hsplit is None:
        hsplit = mlp_hsplit(M, C, HP) if tiles_per_wave == 0 and (x.dtype in _HALF or C < 256) else 1
    if hsplit > 1:
        ypart = torch.empty((hsplit, M, C), dtype=torch.float32, device=x.device)
        a.hsplit, a.ypart, a.tiles_per_wave = hsplit, _p(ypart), 1
    _lib.check(lib.mphsir_gated_mlp_fwd(ctypes.byref(a), _DT[x.dtype], _stream(x)), "gated_mlp_fwd")
    es = x.element_size()
    if branch is not None:        # + the branch sum: v Mb^T, reads of v and sa (x is counted below), the optional write of y
        _acct("gated_mlp", 6.0 * M * C * HP + 2.0 * M * C * C, (4.0 + (1.0 if yb is not None else 0.0)) * M * C * es + 3.0 * C * HP * es)
        return y, yb
    _acct("gated_mlp", 6.0 * M * C * HP, 2.0 * M * C * es + 3.0 * C * HP * es)
    return y


def pack_win_proj(proj_w, heads, dtype):
    """attn.proj.weight (C, C) -> [C][heads*HDP] with each head's input columns zero-padded to HDP."""
    C = proj_w.shape[0]
    hd = C // heads
    hdp = _lib.load().mphsir_win_attn_hdp(hd, _DT[dtype])
    if hdp == hd:
        return proj_w.to(cdt(dtype)).contiguous()
    out = torch.zeros((C, heads, hdp), dtype=cdt(dtype), device=proj_w.device)
    out[:, :, :hd] = proj_w.reshape(C, heads, hd).to(cdt(dtype))
    return out.reshape(C, heads * hdp)


def win_attn_fwd(x, ln_w, ln_b, Wqkv, bqkv, rpb, Wproj, bproj, pg, heads, shift, save=False, gate=True):
    """x (B,H,W,C) contiguous.  pg = dict of the fp32 local_spectral_attn parameters.
    Returns (sa (B,H,W,C), gate (B*nW, C) fp32) and, with save=True, also (mu (B*nW,C) fp32, o_attn [B*nW*64][C] in
    window-token order).  Two launches: the window attention (which also emits the window means) and the gate.
    gate=False: only the first launch -> (sa, mu, o_attn or None); the caller runs pg_gate_fwd(mu, pg) itself (side branch)."""
    lib = _lib.load()
    _check(x, Wqkv, Wproj, *pg.values())
    B, H, W, C = x.shape
    assert x.is_contiguous() and Wqkv.shape == (3 * C, C) and Wqkv.dtype == x.dtype and Wproj.dtype == x.dtype
    sa = torch.empty_like(x)
    mu = torch.empty((B * (H // 8) * (W // 8), C), dtype=torch.float32, device=x.device)
    a = _lib.WinAttnArgs()
    a.X, a.ln_w, a.ln_b = _p(x), _p(ln_w), _p(ln_b)
    a.Wqkv, a.bqkv, a.rpb, a.Wproj, a.bproj = _p(Wqkv), _p(bqkv), _p(rpb), _p(Wproj), _p(bproj)
    a.SA, a.mu = _p(sa), _p(mu)
    oattn = None
    if save:
        oattn = torch.empty_like(x)
        a.Oattn = _p(oattn)
    a.B, a.H, a.W, a.C, a.heads, a.shift = B, H, W, C, heads, shift
    _lib.check(lib.mphsir_win_attn_fwd(ctypes.byref(a), _DT[x.dtype], _stream(x)), "win_attn_fwd")
    M = B * H * W
    _acct("win_attn", M * (8.0 * C * C + 4.0 * 64 * C), 2.0 * M * C * x.element_size() + mu.numel() * 4 + 4.0 * C * C * x.element_size())
    if not gate:
        return sa, mu, oattn
    gate = pg_gate_fwd(mu, pg)
    if save:
        return sa, gate, mu, oattn
    return sa, gate


def pg_gate_fwd(mu, pg):
    """mu (nW,C) fp32 window means -> gate (nW,C) fp32 (PG_Spectral_Attention.forward :136-152)."""
    lib = _lib.load()
    _check(mu, *pg.values())
    nW, C = mu.shape
    r = pg["linear_down.weight"].shape[0]
    gate = torch.empty_like(mu)
    a = _lib.PgFwdArgs()
    a.mu, a.gate = _p(mu), _p(gate)
    a.Wprompt, a.prompt_param = _p(pg["linear_prompt.weight"]), _p(pg["prompt_param"])
    a.Wq, a.Wkv, a.Wdown = _p(pg["q.weight"]), _p(pg["kv.weight"]), _p(pg["linear_down.weight"])
    a.Wpproj, a.bpproj, a.Wup = _p(pg["proj.weight"]), _p(pg["proj.bias"]), _p(pg["linear_up.weight"])
    a.nW, a.C, a.r = nW, C, r
    _lib.check(lib.mphsir_pg_gate_fwd(ctypes.byref(a), _stream(mu)), "pg_gate_fwd")
    _acct("pg_gate", 2.0 * nW * C * (128 + 2 * r), 4.0 * nW * 2 * C)
    return gate


def pack_dw(w):
    """depthwise conv weight (C,1,3,3) -> fp32 [9][C] tap-major."""
    return w.reshape(w.shape[0], 9).t().contiguous().float()


def choose_nsplit(B, H, W):
    """workgroups per sample for dwconv_gram: up to ~1024 workgroups in flight (measured: 256 costs +80 % on the
    Gram kernel, more than the extra partials cost spectral_fold); must divide H*W/64."""
    tiles = H * W // 64
    n = tiles
    while n > 1 and B * n > 1024 and n % 2 == 0:
        n //= 2
    return n


def dwconv_gram(tq, tk, tv, wq, wk, wv, ldw, B, H, W, C, heads, nsplit=None, keep_qk=False):
    """tq/tk/tv: 2-D row-major views [B*H*W, >=C] of the 1x1-conv output starting at the first q/k/v
    channel; wq/wk/wv fp32 tap-major views with row pitch ldw.  Returns (v (M,C), Gpart, Spart, nsplit); with
    keep_qk=True a 5th value: q|k after the depthwise conv as (M, 2C), or None when this shape cannot emit it."""
    lib = _lib.load()
    _check(tq, tk, tv, wq, wk, wv)
    M = B * H * W
    nsplit = nsplit or choose_nsplit(B, H, W)
    hd = C // heads
    v = torch.empty((M, C), dtype=tq.dtype, device=tq.device)
    gp = torch.empty((B, nsplit, heads, hd, hd), dtype=torch.float32, device=tq.device)
    sp = torch.empty((B, nsplit, 2, C), dtype=torch.float32, device=tq.device)
    a = _lib.GramArgs()
    a.Tq, a.ldq, a.Tk, a.ldk, a.Tv, a.ldv = _p(tq), tq.stride(0), _p(tk), tk.stride(0), _p(tv), tv.stride(0)
    a.wq, a.wk, a.wv, a.ldw = _p(wq), _p(wk), _p(wv), ldw
    a.V, a.ldvo, a.Gpart, a.Spart = _p(v), C, _p(gp), _p(sp)
    a.B, a.H, a.W, a.C, a.heads, a.nsplit = B, H, W, C, heads, nsplit
    qk = None
    if keep_qk and lib.mphsir_dwconv_gram_keeps_qk(C, W, _DT[tq.dtype]):
        qk = torch.empty((M, 2 * C), dtype=tq.dtype, device=tq.device)
        a.QK, a.ldqk = _p(qk), 2 * C
    _lib.check(lib.mphsir_dwconv_gram(ctypes.byref(a), _DT[tq.dtype], _stream(tq)), "dwconv_gram")
    _acct("dwconv_gram", M * (54.0 * C + 2.0 * C * hd), (4.0 + (2.0 if qk is not None else 0.0)) * M * C * tq.element_size() + gp.numel() * 4 + sp.numel() * 4)
    _acct("dwconv_gram:qk", 2.0 * M * C * hd, 0.0)            # the QK^T (Gram) FLOPs alone, for bench.py's spectral roofline
    if keep_qk:
        return v, gp, sp, nsplit, qk
    return v, gp, sp, nsplit


# training forward through the fused pass A (t and q|k as extra outputs); MPHSIR_FUSED_TRAIN=0 = 1x1 GEMM + depthwise/Gram kernel
FUSED_TRAIN = os.environ.get("MPHSIR_FUSED_TRAIN", "1") == "1"


def qkv_dwconv_gram_fits(C, heads, H, W, dtype):
    return bool(_lib.load().mphsir_qkv_dwconv_gram_fits(C, heads, H, W, _DT[dtype]))


# workgroups of the fused pass A: 512 = two per CU-slot of work; measured 1.57 -> 1.45 ms/step against 1024 (each workgroup
# then walks two tiles with the next slab's weights already in flight), 256 no better
FUSED_WGS = 512


def fused_wgs(C):
    """workgroup target of the tile form: one round of one workgroup per CU at C >= 256 (measured, tools/bench_c256.py:
    128x128 32.4 -> 27.8 us, batch 32 of 16x16 20.1 -> 16.3 us against two rounds), FUSED_WGS below"""
    return 256 if C >= 256 else FUSED_WGS


def choose_nsplit_fused(B, H, W, C=0):
    """workgroups per sample for the fused pass A (8x16-pixel tiles): up to ~fused_wgs(C) workgroups in all."""
    n = (H // 8) * (W // 16)
    while n > 1 and B * n > fused_wgs(C) and n % 2 == 0:
        n //= 2
    return n


def choose_head_groups(B, nsplit, heads, C=0):
    """workgroups per tile set for the fused pass A: split the heads while that still adds workgroups below the target"""
    g = 1
    while g < heads and heads % (2 * g) == 0 and B * nsplit * 2 * g <= fused_wgs(C):
        g *= 2
    return g


def qkv_dwconv_gram_rows_fits(C, heads, H, W, dtype, ln=False):
    """the row-walking form of the fused pass A (spectral_rows.hip): 16-bit dtypes, no LayerNorm, W % 32 == 0, C <= 192"""
    return bool(_lib.load().mphsir_qkv_dwconv_gram_rows_fits(C, heads, H, W, _DT[dtype], int(bool(ln))))


# the row-walking form: MPHSIR_ROWS_FORM=0 keeps every shape on the tile form
ROWS_FORM = os.environ.get("MPHSIR_ROWS_FORM", "1") == "1"


def choose_row_segments(B, H, W, C, heads):
    """row segments per 32-pixel strip: the most rows per workgroup (least halo recompute: 2 rows per segment) that still
    gives every CU a workgroup"""
    hgroups = max(1, heads // (2 if C // heads == 32 else 1))
    s = 1
    while B * (W // 32) * s * hgroups < 256 and H % (2 * s) == 0 and H // (2 * s) >= 4:
        s *= 2
    return s


def qkv_dwconv_gram(x, wqkv, w9, B, H, W, C, heads, ln=None, nsplit=None, head_groups=None, keep=False, row_segments=None):
    """Fused pass A: x (M, >=C) row-major view, wqkv (3C, C) in x.dtype, w9 fp32 (9, >=3C) taps of q|k|v,
    ln = (weight, bias) fp32 or None.  Returns (v (M,C), Gpart, Spart, nsplit) like dwconv_gram(gemm_tok(x, wqkv));
    keep=True (training) appends t = qkv(LN(x)) (M,3C) and q|k after the depthwise conv (M,2C).
    row_segments: None = the row-walking form where it applies (unless nsplit / head_groups ask for the tile form), 0 = tile
    form, s > 0 = row-walking form with s segments per strip (nsplit = (W/32)*s partial slots per sample)."""
    lib = _lib.load()
    _check(x, wqkv, w9)
    M, ldx = _rows(x)
    assert M == B * H * W and wqkv.shape == (3 * C, C) and wqkv.is_contiguous() and wqkv.dtype == x.dtype
    if row_segments is None:
        row_segments = 0
        if ROWS_FORM and nsplit is None and head_groups is None and qkv_dwconv_gram_rows_fits(C, heads, H, W, x.dtype, ln is not None):
            row_segments = choose_row_segments(B, H, W, C, heads)
    if row_segments:
        nsplit = (W // 32) * row_segments
    nsplit = nsplit or choose_nsplit_fused(B, H, W, C)
    hd = C // heads
    v = torch.empty((M, C), dtype=x.dtype, device=x.device)
    gp = torch.empty((B, nsplit, heads, hd, hd), dtype=torch.float32, device=x.device)
    sp = torch.empty((B, nsplit, 2, C), dtype=torch.float32, device=x.device)
    a = _lib.FusedGramArgs()
    a.X, a.ldx, a.Wqkv, a.w9, a.ldw = _p(x), ldx, _p(wqkv), _p(w9), w9.stride(0)
    if ln is not None:
        a.ln_w, a.ln_b = _p(ln[0]), _p(ln[1])
    a.V, a.ldvo, a.Gpart, a.Spart = _p(v), C, _p(gp), _p(sp)
    a.B, a.H, a.W, a.C, a.heads, a.nsplit = B, H, W, C, heads, nsplit
    a.head_groups = 1 if row_segments else (head_groups or choose_head_groups(B, nsplit, heads, C))
    a.row_segments = row_segments
    t = qk = None
    if keep:
        t = torch.empty((M, 3 * C), dtype=x.dtype, device=x.device)
        qk = torch.empty((M, 2 * C), dtype=x.dtype, device=x.device)
        a.T, a.ldt, a.QK, a.ldqk = _p(t), 3 * C, _p(qk), 2 * C
    _lib.check(lib.mphsir_qkv_dwconv_gram(ctypes.byref(a), _DT[x.dtype], _stream(x)), "qkv_dwconv_gram")
    # the 1x1 conv is counted on the pixels it is useful for (the halo recompute is overhead, not work)
    _acct("qkv_dwconv_gram", M * (6.0 * C * C + 54.0 * C + 2.0 * C * hd), (7.0 if keep else 2.0) * M * C * x.element_size()
          + wqkv.numel() * x.element_size() + gp.numel() * 4 + sp.numel() * 4)
    _acct("qkv_dwconv_gram:qk", 2.0 * M * C * hd, 0.0)
    _acct("qkv_dwconv_gram:rows" if row_segments else "qkv_dwconv_gram:tile", 0.0, 0.0)      # which form ran (tests assert on it)
    if keep:
        return v, gp, sp, nsplit, t, qk
    return v, gp, sp, nsplit


def spectral_fold(gp, sp, temperature, Wo, dtype, transposed=False):
    """-> per-sample folded matrix M (B, C, C) in `dtype`; with transposed=True (training) -> (M, M^T, gsum, ssum):
    gsum (B,1,heads,hd,hd) / ssum (B,1,2,C) are the reduced partials, the form spectral_fold_bwd wants."""
    lib = _lib.load()
    _check(gp, sp, temperature, Wo)
    B, nsplit, heads, hd, _ = gp.shape
    C = heads * hd
    assert Wo.dtype == torch.float32 and Wo.is_contiguous() and temperature.is_contiguous()
    if nsplit > 64:
        # few samples, large images (a 512x512 test cube: B = 1, 1024 partials): the fold kernel has only B*heads*C/32
        # workgroups, so the long ordered sum is done by the wide reduction kernel first
        with reduce_scope():            # both sums in ONE launch (its own scope: they are read right below)
            gp = reduce_parts(gp, batched=True)
            sp = reduce_parts(sp, batched=True)
        gp, sp = gp.unsqueeze(1), sp.unsqueeze(1)
        nsplit = 1
    Mo = torch.empty((B, C, C), dtype=dtype, device=gp.device)
    a = _lib.FoldArgs()
    a.Gpart, a.Spart, a.temperature, a.Wo, a.M = _p(gp), _p(sp), _p(temperature), _p(Wo), _p(Mo)
    MT = torch.empty_like(Mo) if transposed else None
    a.MT = _p(MT)
    gsum = ssum = None
    if transposed:
        gsum = torch.empty((B, 1, heads, hd, hd), dtype=torch.float32, device=gp.device)
        ssum = torch.empty((B, 1, 2, C), dtype=torch.float32, device=gp.device)
        a.Gsum, a.Ssum = _p(gsum), _p(ssum)
    a.B, a.C, a.heads, a.nsplit = B, C, heads, nsplit
    _lib.check(lib.mphsir_spectral_fold(ctypes.byref(a), _DT[dtype], _stream(gp)), "spectral_fold")
    _acct("spectral_fold", 2.0 * B * C * C * hd, gp.numel() * 4 + sp.numel() * 4 + C * C * 4 + Mo.numel() * Mo.element_size())
    return (Mo, MT, gsum, ssum) if transposed else Mo


def dwconv_gate(t, w9, B, H, W):
    """t (M, 2*HP) -> gelu(dw(t)[:, :HP]) * dw(t)[:, HP:]  (M, HP)."""
    lib = _lib.load()
    _check(t, w9)
    M, ldt = _rows(t)
    HP = t.shape[1] // 2
    u = torch.empty((M, HP), dtype=t.dtype, device=t.device)
    a = _lib.GateArgs()
    a.T, a.ldt, a.w9, a.ldw, a.U, a.ldu = _p(t), ldt, _p(w9), w9.stride(0), _p(u), HP
    a.B, a.H, a.W, a.HP = B, H, W, HP
    _lib.check(lib.mphsir_dwconv_gate(ctypes.byref(a), _DT[t.dtype], _stream(t)), "dwconv_gate")
    _acct("dwconv_gate", M * HP * 40.0, 3.0 * M * HP * t.element_size())
    return u


def gdfn_fused_fits(D, HP, H, W, dtype):
    return bool(_lib.load().mphsir_gdfn_fused_fits(D, HP, H, W, _DT[dtype]))


def gdfn_fused(x2, ln, w_in, w9, w_out, B, H, W, nsplit=None, keep=False):
    """x2 (M,D) -> x2 + project_out(gelu(x1) * x2'),  [x1|x2'] = dwconv3x3(project_in(LN(x2)))  in one launch.
    w_in (2HP,D), w9 (9,2HP) fp32, w_out (D,HP).  keep=True (training) also returns t = project_in(LN(x2)) (M,2HP)."""
    lib = _lib.load()
    _check(x2, w_in, w9, w_out, ln[0], ln[1])
    M, ldx = _rows(x2)
    D, HP = x2.shape[1], w_out.shape[1]
    if nsplit is None:
        nsplit = (H // 8) * (W // lib.mphsir_gdfn_fused_tile_width(D))
        while nsplit > 1 and B * nsplit > GDFN_WGS and nsplit % 2 == 0:
            nsplit //= 2
    y = torch.empty((M, D), dtype=x2.dtype, device=x2.device)
    a = _lib.GdfnArgs()
    a.X, a.ldx, a.ln_w, a.ln_b, a.Win, a.w9, a.ldw, a.Wout = _p(x2), ldx, _p(ln[0]), _p(ln[1]), _p(w_in), _p(w9), w9.stride(0), _p(w_out)
    a.Y, a.ldy, a.B, a.H, a.W, a.D, a.HP, a.nsplit = _p(y), D, B, H, W, D, HP, nsplit
    t = torch.empty((M, 2 * HP), dtype=x2.dtype, device=x2.device) if keep else None
    a.T, a.ldt = _p(t), 2 * HP
    _lib.check(lib.mphsir_gdfn_fused(ctypes.byref(a), _DT[x2.dtype], _stream(x2)), "gdfn_fused")
    _acct("gdfn_fused", 2.0 * M * 3 * HP * D + M * HP * 40.0, (2.0 * M * D + (2.0 * M * HP if keep else 0.0)) * x2.element_size())
    return (y, t) if keep else y


# workgroups of the fused GDFN (its tile takes ~100 KB of LDS: one per CU at a time; two rounds measured 3 % faster than one: 291 vs 300 us)
GDFN_WGS = 512
# below this many pixels the three launches are as fast (a handful of tiles cannot fill the chip either way)
GDFN_FUSED_MIN_PIXELS = 16384


_WEIGHT_EPOCH = [0]


def weight_epoch():
    return _WEIGHT_EPOCH[0]


def bump_weight_epoch():
    """Call after updating parameters through raw pointers (the flat-arena optimizer kernel): the
    per-module packed-weight caches key on this counter as well as on tensor version counters."""
    _WEIGHT_EPOCH[0] += 1


# ---- kernel-layout weights: per-module caches and the one-launch pack plan ------------------------------------------
_TRACE = [False]


def cdt(dtype):
    """dtype the packers materialise a kernel-layout weight in: `dtype`, except while a PackPlan traces the packers with
    index-valued parameters (fp32 keeps the indices exact; the LAYOUT still follows `dtype`)."""
    return torch.float32 if _TRACE[0] else dtype


class WeightCache:
    """Cache of one module's weights converted / padded / transposed for the kernels, keyed on the compute dtype, the
    version counters of the source parameters (in-place updates, load_state_dict) and the package-wide epoch that
    raw-pointer optimizers bump (bump_weight_epoch).  A PackPlan may `pin` persistent buffers it refreshes itself with
    one pack_gather launch per optimizer step; the pinned value is served while the plan is in sync with the epoch."""
    _all = []

    def __init__(self):
        self.key = self.val = self.last = self.pinned = None
        WeightCache._all.append(weakref.ref(self))

    @classmethod
    def live(cls):
        out = [r() for r in cls._all]
        cls._all = [r for r, c in zip(cls._all, out) if c is not None]
        return [c for c in out if c is not None]

    def get(self, params, dtype, build):
        pin = self.pinned
        versions = tuple(p._version for p in params)
        if pin is not None and pin[0] == dtype and pin[1].epoch == weight_epoch() and pin[2] == versions:
            return pin[3]
        key = (dtype, weight_epoch(), versions, params[0].device)
        if key != self.key:
            with torch.no_grad():
                self.val = build()
            self.key = key
            self.last = (list(params), dtype, build)
        return self.val


def flat_adamw(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, grad_scale=1.0, hyper=None):
    """In-place AdamW step on flat fp32 arenas (length a multiple of 4).  hyper: optional device fp32 tensor
    [lr, 1-beta1^step, sqrt(1-beta2^step)] read by the kernel instead of lr/step (for captured launches)."""
    lib = _lib.load()
    _check(p, g, m, v)
    assert p.dtype == torch.float32 and p.is_contiguous() and p.numel() == g.numel() == m.numel() == v.numel()
    _lib.check(lib.mphsir_flat_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                                     grad_scale, _p(hyper), _stream(p)), "flat_adamw")
    bump_weight_epoch()


def new_loss_scaler(device, init_scale=65536.0):
    """device state of the dynamic loss scaler (torch GradScaler defaults: 2^16, x2 every 2000 good steps, x0.5 on overflow):
    fp32 [scale, good steps in a row, found-inf flag, optimizer steps taken]."""
    return torch.tensor([init_scale, 0.0, 0.0, 0.0], dtype=torch.float32, device=device)


def scaled_adamw_step(p, g, m, v, scaler, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2, grad_scale=1.0, hyper=None,
                      growth=2.0, backoff=0.5, interval=2000):
    """the fp16 path's optimizer step: non-finite check of the (all-reduced) gradients, AdamW on g / scale unless the check
    fired, scaler update -- three capturable launches (include/mphsir.h)."""
    lib = _lib.load()
    _check(p, g, m, v, scaler)
    assert scaler.dtype == torch.float32 and scaler.numel() == 4 and p.numel() == g.numel()
    st = _stream(p)
    _lib.check(lib.mphsir_grad_check(_p(g), g.numel(), _p(scaler), st), "grad_check")
    _lib.check(lib.mphsir_flat_adamw_scaled(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, grad_scale,
                                            _p(hyper), _p(scaler), st), "flat_adamw_scaled")
    _lib.check(lib.mphsir_scaler_update(_p(scaler), growth, backoff, interval, st), "scaler_update")
    bump_weight_epoch()


def dwconv3x3(x, w9, flip=False, out=None):
    """x (B,H,W,C) channels-last (last dim contiguous, row pitch = stride of W axis); w9 fp32 [9][C] view.
    out: optional (B,H,W,C) view to write (a channel slice of a wider buffer: its pixel pitch is passed on) instead of a new tensor."""
    lib = _lib.load()
    _check(x, w9)
    B, H, W, C = x.shape
    assert x.stride(3) == 1 and x.stride(1) == W * x.stride(2) and x.stride(0) == H * x.stride(1)
    y = out if out is not None else torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    assert y.shape == x.shape and y.dtype == x.dtype and y.stride(3) == 1 and y.stride(1) == W * y.stride(2) and y.stride(0) == H * y.stride(1)
    _lib.check(lib.mphsir_dwconv3x3(_p(x), x.stride(2), _p(w9), w9.stride(0), _p(y), y.stride(2), B, H, W, C, int(flip),
                                    _DT[x.dtype], _stream(x)), "dwconv3x3")
    _acct("dwconv3x3", 18.0 * B * H * W * C, 2.0 * B * H * W * C * x.element_size())
    return y


def dwconv3x3_wgrad(x, dy, nblk=None, col_ranges=None, out=None):
    """-> fp32 [9][C] = sum_p x[p+tap] * dy[p].  With col_ranges=[(c0, nc), ...] the result is returned in the
    parameter's own layout instead: (sum nc, 9), channel-major, only those channel ranges (transposed + un-padded inside
    the partial reduction); `out` = a (sum nc, 9) fp32 view to write that into (e.g. a row range of a joint buffer)."""
    lib = _lib.load()
    _check(x, dy)
    B, H, W, C = x.shape
    assert x.stride(3) == 1 and dy.stride(3) == 1 and dy.shape == x.shape
    if nblk is None:
        if lib.mphsir_dwconv3x3_wgrad_tiled(H, W, C, _DT[x.dtype]):      # LDS-tile form: one workgroup per CU walks the tiles
            nblk = max(1, min(B * (H // 8) * (W // 16), 256 // ((C + 95) // 96)))
        else:
            nblk = max(1, min(1024, B * H * W // 128))
    part = torch.empty((nblk, 9, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.mphsir_dwconv3x3_wgrad(_p(x), x.stride(2), _p(dy), dy.stride(2), _p(part), nblk, B, H, W, C,
                                          _DT[x.dtype], _stream(x)), "dwconv3x3_wgrad")
    _acct("dwconv3x3_wgrad", 18.0 * B * H * W * C, 2.0 * B * H * W * C * x.element_size())
    if col_ranges is None:
        return reduce_parts(part)
    if out is None:
        out = torch.empty((sum(nc for _, nc in col_ranges), 9), dtype=torch.float32, device=x.device)
    assert tuple(out.shape) == (sum(nc for _, nc in col_ranges), 9) and out.dtype == torch.float32
    o = 0
    for c0, nc in col_ranges:
        reduce_block(part, 0, 9, c0, nc, out[o:o + nc], transpose=True)
        o += nc
    return out


def dwconv3x3_bwd_fits(H, W, C, dtype):
    return bool(_lib.load().mphsir_dwconv3x3_wgrad_tiled(H, W, C, _DT[dtype]))


def dwconv3x3_bwd(x, dy, w9, col_ranges=None):
    """Both gradients of y = dwconv3x3(x, w9) in one launch: -> (dx, dw) with dx = dwconv3x3(dy, w9, flip=True) and
    dw = dwconv3x3_wgrad(x, dy, col_ranges=col_ranges).  Shapes dwconv3x3_bwd_fits accepts; else the two launches."""
    lib = _lib.load()
    _check(x, dy, w9)
    B, H, W, C = x.shape
    if not dwconv3x3_bwd_fits(H, W, C, x.dtype):
        return dwconv3x3(dy, w9, flip=True), dwconv3x3_wgrad(x, dy, col_ranges=col_ranges)
    assert x.stride(3) == 1 and dy.stride(3) == 1 and dy.shape == x.shape
    for t in (x, dy):
        assert t.stride(1) == W * t.stride(2) and t.stride(0) == H * t.stride(1)
    nblk = max(1, min(B * (H // 8) * (W // 16), 256 // ((C + 95) // 96)))
    dx = torch.empty((B, H, W, C), dtype=x.dtype, device=x.device)
    part = torch.empty((nblk, 9, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.mphsir_dwconv3x3_bwd(_p(x), x.stride(2), _p(dy), dy.stride(2), _p(w9), w9.stride(0), _p(dx), C, _p(part), nblk,
                                        B, H, W, C, _DT[x.dtype], _stream(x)), "dwconv3x3_bwd")
    _acct("dwconv3x3_bwd", 36.0 * B * H * W * C, 3.0 * B * H * W * C * x.element_size())
    if col_ranges is None:
        return dx, reduce_parts(part)
    out = torch.empty((sum(nc for _, nc in col_ranges), 9), dtype=torch.float32, device=x.device)
    o = 0
    for c0, nc in col_ranges:
        reduce_block(part, 0, 9, c0, nc, out[o:o + nc], transpose=True)
        o += nc
    return dx, out


# ---- backward of the channel attention between the fold and the 1x1 conv in one launch (csrc/spectral_bwd.hip) ----------------
SPECTRAL_BWD_FUSED = os.environ.get("MPHSIR_SPECTRAL_BWD_FUSED", "1") == "1"     # 0: the three launches it replaces (gemm_tok x2 + dwconv3x3_bwd)
SPECTRAL_BWD_WGS = int(os.environ.get("MPHSIR_SPECTRAL_BWD_WGS", "512"))          # workgroups per launch aimed at (two per CU)


def spectral_dqkv_bwd_fits(C, heads, H, W, dtype):
    return SPECTRAL_BWD_FUSED and dtype in _HALF and bool(_lib.load().mphsir_spectral_dqkv_bwd_fits(C, heads, H, W, _DT[dtype]))


def spectral_dqkv_bwd(qk, d_out, t, W2, MbT, w9, B, H, W, C, heads, nblk=None, round_dall=False, vscale=None):
    """dt (M, 3C) and the tap gradients (3C, 9) of the channel attention from q|k (M, 2C), d_out (M, C), t (M, 3C), the per-sample
    matrices W2 (B, 2C, 2C) / MbT (B, C, C) and the taps w9 fp32 [9][3C]:  dv = d_out M_b, [dq|dk] = [q|k] W2^T, then both gradients of
    the depthwise conv -- [dq|dk|dv] stays on the chip (include/mphsir.h).  nblk: tile ranges (x channel slabs = workgroups)."""
    lib = _lib.load()
    _check(qk, d_out, t, W2, MbT, w9)
    M = B * H * W
    for x_, n in ((qk, 2 * C), (d_out, C), (t, 3 * C)):
        assert x_.dim() == 2 and x_.shape == (M, n) and x_.stride(1) == 1, (tuple(x_.shape), tuple(x_.stride()), n)
    assert W2.is_contiguous() and MbT.is_contiguous() and W2.shape == (B, 2 * C, 2 * C) and MbT.shape == (B, C, C)
    assert w9.dtype == torch.float32 and w9.shape == (9, 3 * C) and w9.stride(1) == 1
    nslab = lib.mphsir_spectral_dqkv_bwd_slabs(C, heads)
    tiles = B * (H // 8) * (W // 16)
    if nblk is None:
        nblk = max(1, min(tiles, SPECTRAL_BWD_WGS // nslab))
        if nblk >= 8:
            nblk = nblk // 8 * 8          # grid a multiple of 8: the slabs of a tile range then share an XCD (xcd_contiguous_block)
    dt = torch.empty((M, 3 * C), dtype=qk.dtype, device=qk.device)
    part = torch.empty((nblk, 9, 3 * C), dtype=torch.float32, device=qk.device)
    a = _lib.SpectralBwdArgs()
    a.QK, a.ldqk, a.DO, a.lddo, a.T, a.ldt = _p(qk), qk.stride(0), _p(d_out), d_out.stride(0), _p(t), t.stride(0)
    a.W2, a.MbT, a.w9, a.ldw, a.dT, a.lddt, a.part = _p(W2), _p(MbT), _p(w9), w9.stride(0), _p(dt), 3 * C, _p(part)
    a.B, a.H, a.W, a.C, a.heads, a.nblk, a.round_dall = B, H, W, C, heads, nblk, int(round_dall)
    a.vscale = _p(vscale)
    _lib.check(lib.mphsir_spectral_dqkv_bwd(ctypes.byref(a), _DT[qk.dtype], _stream(qk)), "spectral_dqkv_bwd")
    hd = C // heads
    _acct("spectral_dqkv_bwd", M * (2.0 * C * C + 8.0 * C * hd + 108.0 * C), 10.0 * M * C * qk.element_size() + part.numel() * 4)
    dw = torch.empty((3 * C, 9), dtype=torch.float32, device=qk.device)
    reduce_block(part, 0, 9, 0, 3 * C, dw, transpose=True)
    return dt, dw


def gated_mlp_bwd(x, dy, dm, ln_w, ln_b, W1, b1, W1T, W2T, variant=0, keep=None, rows_per_batch=0, hsplit=None, operands=True):
    """-> dx, xn, h, dpre, part (see include/mphsir.h).  x, dy, dm: contiguous (M,C).  With keep (DropPath factors, one
    per rows_per_batch rows) dm is ignored as input: the kernel computes keep*dy itself and it is returned as a 6th value.
    operands=False: h and dpre are not written (returned as None): the parameter gradients come from gated_mlp_wgrad."""
    lib = _lib.load()
    _check(x, dy, dm, W1, W1T, W2T)
    M, C = x.shape
    HP = W2T.shape[0]
    assert x.is_contiguous() and dy.is_contiguous() and (dm is None or dm.is_contiguous()) and W1T.shape == (C, 2 * HP) and W2T.shape == (HP, C)
    dev, dt = x.device, x.dtype
    dx = torch.empty_like(x)
    xn = torch.empty_like(x)
    h = torch.empty((M, HP), dtype=dt, device=dev) if operands else None
    dpre = torch.empty((M, 2 * HP), dtype=dt, device=dev) if operands else None
    part = torch.empty((M // 64, 2, C), dtype=torch.float32, device=dev)
    a = _lib.MlpBwdArgs()
    if keep is not None:
        _check(keep)
        assert keep.dtype == torch.float32 and keep.is_contiguous() and rows_per_batch > 0
        dm = torch.empty_like(dy)
        a.keep, a.rows_per_batch = _p(keep), rows_per_batch
    a.X, a.dY, a.DM, a.ln_w, a.ln_b = _p(x), _p(dy), _p(dm), _p(ln_w), _p(ln_b)
    a.W1, a.b1, a.W1T, a.W2T = _p(W1), _p(b1), _p(W1T), _p(W2T)
    a.dX, a.XN, a.H, a.DPRE, a.part = _p(dx), _p(xn), _p(h) if operands else None, _p(dpre) if operands else None, _p(part)
    a.M, a.C, a.HP, a.variant = M, C, HP, variant
    if hsplit is None:
        hsplit = mlp_hsplit(M, C, HP) if variant == 0 and dt in _HALF else 1
    if hsplit > 1:
        dxn_part = torch.empty((hsplit, M, C), dtype=torch.float32, device=dev)
        a.hsplit, a.dxn_part = hsplit, _p(dxn_part)
    _lib.check(lib.mphsir_gated_mlp_bwd(ctypes.byref(a), _DT[dt], _stream(x)), "gated_mlp_bwd")
    _acct("gated_mlp_bwd", 12.0 * M * C * HP, (3.0 * M * C + (3.0 * M * HP if operands else 0.0) + M * C) * x.element_size())
    if keep is not None:
        return dx, xn, h, dpre, part, dm
    return dx, xn, h, dpre, part


MLP_WGRAD = os.environ.get("MPHSIR_MLP_WGRAD", "1") == "1"      # parameter gradients of the gated MLP by recomputation (no h / dpre in HBM)
MLP_WGRAD_NCH = int(os.environ.get("MPHSIR_MLP_WGRAD_NCH", "0"))  # chunks of 32 hidden units per workgroup: 0 = per shape, 1, 2
MLP_WGRAD_WGS = int(os.environ.get("MPHSIR_MLP_WGRAD_WGS", "512"))
MLP_WGRAD_MAXC = int(os.environ.get("MPHSIR_MLP_WGRAD_MAXC", "128"))
MLP_WGRAD_CAP = float(os.environ.get("MPHSIR_MLP_WGRAD_CAP", "4.0"))   # partial bytes <= this x the bytes of the two token matrices (1 / 2 / 4: 21.04-21.09 / 20.98-20.99 / 20.95-20.97 ms per step: at M = 32768 a full round of workgroups -- 40 ranges instead of 24 -- is worth more than the 9 MB of partials it adds)


def gated_mlp_wgrad_fits(M, C, HP, dtype):
    """where the library takes the recomputing weight-gradient kernel: 16-bit types, >= 16384 tokens (below, the partials of a range
    outweigh the token matrices), C <= 128 -- at C = 192 the kernel has no registers left for the tile in flight (19 spilled) and
    measured slower than the operand path: remote-sensing fp16 step 27.0 against 26.2 ms, 195-210 against 120 us per launch"""
    return (MLP_WGRAD and dtype in _HALF and M % 64 == 0 and HP % 32 == 0 and M >= 16384 and C <= MLP_WGRAD_MAXC
            and bool(_lib.load().mphsir_gated_mlp_wgrad_fits(C, 1, _DT[dtype])))


def mlp_wgrad_plan(M, C, HP, dtype, nch=None, ranges=None):
    """(chunks per workgroup, token ranges): ~MLP_WGRAD_WGS workgroups, ranges a multiple of 8, partial bytes capped"""
    lib = _lib.load()
    if nch is None:
        nch = MLP_WGRAD_NCH or 1
    if not lib.mphsir_gated_mlp_wgrad_fits(C, nch, _DT[dtype]):
        nch = 1
    S = (HP // 32 + nch - 1) // nch
    if ranges is None:
        # one round of resident workgroups (two 256-thread workgroups or one 512-thread workgroup per CU); a partial of
        # 3 HP C floats per range: capped against the bytes of the two token matrices
        ranges = max(1, (MLP_WGRAD_WGS // nch) // S)
        cap = int(MLP_WGRAD_CAP * 2.0 * M * C * 2 / (3.0 * HP * C * 4))
        ranges = max(8, min(ranges, cap, M // 64) // 8 * 8)
    return nch, ranges


def gated_mlp_wgrad(xn, dm, W1, b1, W2T, hid, nch=None, ranges=None):
    """The four parameter gradients of the gated MLP from xn = LN(x) and dm = keep*dy (M,C), recomputing value / gate / dh per
    hidden slab (csrc/gated_mlp_wgrad.hip): -> dW1 (2*hid, C) [value rows | gate rows], db1 (2*hid,), dW2 (C, hid), db2 (C,), fp32,
    ordered sums of per-range partials (deferred like every reduce_parts; inside a reduce_scope the launch itself is deferred
    to the scope's exit, i.e. to the weight-gradient branch)."""
    lib = _lib.load()
    _check(xn, dm, W1, W2T)
    M, C = xn.shape
    HP = W2T.shape[0]
    assert xn.is_contiguous() and dm.is_contiguous() and W1.shape == (2 * HP, C) and W2T.shape == (HP, C) and xn.dtype in _HALF
    nch, R = mlp_wgrad_plan(M, C, HP, xn.dtype, nch, ranges)
    dev = xn.device
    dW1p = torch.empty((R, 2 * HP, C), dtype=torch.float32, device=dev)
    dW2p = torch.empty((R, C, HP), dtype=torch.float32, device=dev)
    db1p = torch.empty((R, 1, 2 * HP), dtype=torch.float32, device=dev)
    db2p = torch.empty((R, C), dtype=torch.float32, device=dev)
    a = _lib.MlpWgradArgs()
    a.XN, a.DM, a.W1, a.b1, a.W2T = _p(xn), _p(dm), _p(W1), _p(b1), _p(W2T)
    a.dW1p, a.dW2p, a.db1p, a.db2p = _p(dW1p), _p(dW2p), _p(db1p), _p(db2p)
    a.M, a.C, a.HP, a.ranges, a.chunks_per_wg = M, C, HP, R, nch
    dt = _DT[xn.dtype]

    def run():
        _lib.check(lib.mphsir_gated_mlp_wgrad(ctypes.byref(a), dt, _stream(xn)), "gated_mlp_wgrad")

    if _SCOPE is not None:
        _SCOPE.calls.append(dict(run=run, keep=(xn, dm, W1, b1, W2T, dW1p, dW2p, db1p, db2p, a)))
    else:
        run()
    _acct("gated_mlp_wgrad", 12.0 * M * C * HP, 2.0 * M * C * xn.element_size())
    _acct("gated_mlp_wgrad:partials", 0.0, (dW1p.numel() + dW2p.numel()) * 4.0)
    dW1 = torch.empty((2 * hid, C), dtype=torch.float32, device=dev)
    db1 = torch.empty((1, 2 * hid), dtype=torch.float32, device=dev)
    dW2 = torch.empty((C, hid), dtype=torch.float32, device=dev)
    reduce_block(dW1p, 0, hid, 0, C, dW1[:hid])
    reduce_block(dW1p, HP, hid, 0, C, dW1[hid:])
    reduce_block(db1p, 0, 1, 0, hid, db1[:, :hid])
    reduce_block(db1p, 0, 1, HP, hid, db1[:, hid:])
    reduce_block(dW2p, 0, C, 0, hid, dW2)
    db2 = reduce_parts(db2p)
    return dW1, db1[0], dW2, db2


def combine_bwd(dy, sa, gate, keep, shift, want_dout=True):
    """dy, sa (B,H,W,C) -> (d_out (= dy when keep is None; None with want_dout=False: the consumers apply keep themselves), d_sa,
    dgate (B*nW,C) fp32)."""
    lib = _lib.load()
    _check(dy, sa, gate, keep)
    B, H, W, C = dy.shape
    assert dy.is_contiguous() and sa.is_contiguous()
    d_out = (torch.empty_like(dy) if want_dout else None) if keep is not None else dy
    d_sa = torch.empty_like(dy)
    dgate = torch.empty_like(gate)
    _lib.check(lib.mphsir_combine_bwd(_p(dy), _p(sa), _p(gate), _p(keep), _p(d_out) if (keep is not None and want_dout) else None, _p(d_sa),
                                      _p(dgate), B, H, W, C, shift, _DT[dy.dtype], _stream(dy)), "combine_bwd")
    _acct("combine_bwd", 4.0 * dy.numel(), 4.0 * dy.numel() * dy.element_size())
    return d_out, d_sa, dgate


def win_attn_bwd_fits(C, heads, dtype):
    return bool(_lib.load().mphsir_win_attn_bwd_fits(C, heads, _DT[dtype]))


def win_attn_bwd(x, dsa, dmu, ln_w, ln_b, Wqkv, bqkv, rpb, WprojT, heads, shift, head_split=0):
    """-> dqkv (M,3C) and xn (M,C) in window-token order, dsa_total (B,H,W,C), drpb (B*nW,225,heads) fp32."""
    lib = _lib.load()
    _check(x, dsa, dmu, Wqkv, WprojT)
    B, H, W, C = x.shape
    M = B * H * W
    assert x.is_contiguous() and dsa.is_contiguous() and WprojT.shape == (C, C)
    dqkv = torch.empty((M, 3 * C), dtype=x.dtype, device=x.device)
    xnw = torch.empty((M, C), dtype=x.dtype, device=x.device)
    dsat = torch.empty_like(x)
    drpb = torch.empty((M // 64, 225, heads), dtype=torch.float32, device=x.device)
    a = _lib.WinAttnBwdArgs()
    a.X, a.dSA, a.dmu, a.ln_w, a.ln_b = _p(x), _p(dsa), _p(dmu), _p(ln_w), _p(ln_b)
    a.Wqkv, a.bqkv, a.rpb, a.WprojT = _p(Wqkv), _p(bqkv), _p(rpb), _p(WprojT)
    a.dQKV, a.XNw, a.dSAt, a.drpb = _p(dqkv), _p(xnw), _p(dsat), _p(drpb)
    a.B, a.H, a.W, a.C, a.heads, a.shift, a.head_split = B, H, W, C, heads, shift, head_split
    _lib.check(lib.mphsir_win_attn_bwd(ctypes.byref(a), _DT[x.dtype], _stream(x)), "win_attn_bwd")
    _acct("win_attn_bwd", M * (8.0 * C * C + 10.0 * 64 * C), 7.0 * M * C * x.element_size())
    return dqkv, xnw, dsat, drpb


def ln_bwd_win(x, dxn_w, dres, ln_w, shift):
    """dx = dres + LN_backward(dxn_w) (dxn_w in window-token order) -> (dx (B,H,W,C), part (B*nW,2,C))."""
    lib = _lib.load()
    _check(x, dxn_w, dres, ln_w)
    B, H, W, C = x.shape
    assert x.is_contiguous() and dxn_w.is_contiguous() and dres.is_contiguous()
    dx = torch.empty_like(x)
    part = torch.empty((B * H * W // 64, 2, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.mphsir_ln_bwd_win(_p(x), _p(dxn_w), _p(dres), _p(ln_w), _p(dx), _p(part), B, H, W, C, shift, None, None, 0,
                                     _DT[x.dtype], _stream(x)), "ln_bwd_win")
    _acct("ln_bwd_win", 10.0 * x.numel(), 4.0 * x.numel() * x.element_size())
    return dx, part


LN_BWD_DXN = os.environ.get("MPHSIR_LN_BWD_DXN", "1") == "1"      # d_xn = dQKV Wqkv formed inside the LayerNorm-backward launch (0: gemm_tok + ln_bwd_win)
LN_BWD_DXN_MAX_ROWS = int(os.environ.get("MPHSIR_LN_BWD_DXN_ROWS", "0"))      # 0: every size; else only launches of at most this many token rows


def ln_bwd_win_dxn_fits(M, C, dtype):
    return (LN_BWD_DXN and dtype in _HALF and (LN_BWD_DXN_MAX_ROWS == 0 or M <= LN_BWD_DXN_MAX_ROWS)
            and bool(_lib.load().mphsir_ln_bwd_win_dxn_fits(C, _DT[dtype])))


def ln_bwd_win_dxn(x, dqkv, wqkvT, dres, ln_w, shift, dres2=None):
    """dx = dres + [dres2 +] LN_backward(dqkv wqkvT^T) with the token GEMM inside the launch (dqkv (M, 3C) in window-token order, wqkvT (C, 3C))
    -> (dx (B,H,W,C), part (B*nW,2,C)).  dres2: a second residual gradient in image order (the skip of the enclosing BaseBlock)."""
    lib = _lib.load()
    _check(x, dqkv, wqkvT, dres, ln_w, dres2)
    B, H, W, C = x.shape
    assert x.is_contiguous() and dqkv.is_contiguous() and dres.is_contiguous() and wqkvT.is_contiguous()
    assert dqkv.shape == (B * H * W, 3 * C) and wqkvT.shape == (C, 3 * C) and wqkvT.dtype == x.dtype
    assert dres2 is None or (dres2.is_contiguous() and dres2.shape == x.shape and dres2.dtype == x.dtype)
    dx = torch.empty_like(x)
    part = torch.empty((B * H * W // 64, 2, C), dtype=torch.float32, device=x.device)
    _lib.check(lib.mphsir_ln_bwd_win_dxn(_p(x), _p(dqkv), _p(wqkvT), _p(dres), _p(dres2), _p(ln_w), _p(dx), _p(part), B, H, W, C, shift,
                                         _DT[x.dtype], _stream(x)), "ln_bwd_win_dxn")
    _acct("ln_bwd_win", 10.0 * x.numel() + 6.0 * x.numel() * C, 6.0 * x.numel() * x.element_size())
    _acct("ln_bwd_win:dxn", 0.0, 0.0)
    return dx, part


def ln_bwd_tok_dxn_f32_fits(M, C, dtype):
    """... with fp32 x rows and 16-bit dy / weights (TVSP's norm11)"""
    return ln_bwd_win_dxn_fits(M, C, dtype) and C <= 192


def ln_bwd_tok_dxn(x2, dy, wT, dres, ln_w, ln_b, want_xn=True):
    """ln_bwd_tok with the 1x1 conv's data gradient inside: d_xn = dy wT^T (dy (M, K), wT (C, K)) is formed per 64 rows on the matrix cores
    -> (dx = dres + LN_backward(d_xn), d ln weight, d ln bias, LN(x)) for x2 (M, C), M % 64 == 0.  dres may be None (no residual path);
    x2 may be fp32 while dy / wT / dres are a 16-bit type (dx then fp32 too; LN(x), if wanted, in the 16-bit type)."""
    lib = _lib.load()
    _check(x2, dy, wT, dres, ln_w, ln_b)
    M, C = x2.shape
    K = dy.shape[1]
    xf = x2.dtype == torch.float32 and dy.dtype != torch.float32
    assert x2.is_contiguous() and dy.is_contiguous() and (dres is None or dres.is_contiguous()) and wT.is_contiguous() and M % 64 == 0 and K % 32 == 0
    assert dy.shape == (M, K) and wT.shape == (C, K) and wT.dtype == dy.dtype and (xf or dy.dtype == x2.dtype) and (dres is None or dres.dtype == dy.dtype)
    dx = torch.empty_like(x2)
    xn = torch.empty((M, C), dtype=dy.dtype, device=x2.device) if want_xn else None
    part = torch.empty((M // 64, 2, C), dtype=torch.float32, device=x2.device)
    _lib.check(lib.mphsir_ln_bwd_tok_dxn(_p(x2), _p(dy), _p(wT), _p(dres), _p(ln_w), _p(ln_b), _p(dx), _p(xn), _p(part), M, C, K, 1 if xf else 0,
                                         _DT[dy.dtype], _stream(x2)), "ln_bwd_tok_dxn")
    _acct("ln_bwd_win", 12.0 * x2.numel() + 2.0 * M * C * K, (5.0 * C + K) * M * x2.element_size())
    _acct("ln_bwd_win:dxn", 0.0, 0.0)
    g = reduce_parts(part)
    return dx, g[0], g[1], xn


def ln_bwd_tok(x2, dxn, dres, ln_w, ln_b):
    """plain token order: (dx = dres + LN_backward(dxn), d ln weight, d ln bias, LN(x)) for x2 (M,C), M % 64 == 0."""
    lib = _lib.load()
    _check(x2, dxn, dres, ln_w, ln_b)
    M, C = x2.shape
    assert x2.is_contiguous() and dxn.is_contiguous() and dres.is_contiguous() and M % 64 == 0
    dx, xn = torch.empty_like(x2), torch.empty_like(x2)
    part = torch.empty((M // 64, 2, C), dtype=torch.float32, device=x2.device)
    _lib.check(lib.mphsir_ln_bwd_win(_p(x2), _p(dxn), _p(dres), _p(ln_w), _p(dx), _p(part), M // 64, 8, 8, C, 0, _p(ln_b), _p(xn), 1,
                                     _DT[x2.dtype], _stream(x2)), "ln_bwd_win")
    _acct("ln_bwd_win", 12.0 * x2.numel(), 5.0 * x2.numel() * x2.element_size())
    g = reduce_parts(part)
    return dx, g[0], g[1], xn


TN_GROUPED = True          # weight-gradient GEMMs of a backward function in one launch
TN_BIG_TILES = True        # bf16 token-reduction GEMMs use the transposed-LDS-read kernel (ds_read_b64_tr_b16)
# 16-bit big-tile kernel: 1 = transposed-read kernel (256 threads, ~2 workgroups per CU), 2 = its ring form (512 threads, one per CU),
# 0 = per problem: the ring form where it measured faster -- ONE output tile, unbatched, >= 65536 tokens (tools/bench/bench_tn.py) -- else 1
TN_FORM = int(os.environ.get("MPHSIR_TN_FORM", "0"))
TN_RING_WGS = 256          # ring form: workgroups per launch aimed at (one per CU)
TN_PART_CAP = float(os.environ.get("MPHSIR_TN_PART_CAP", "0.4"))    # 0 / 0.2 / 0.3 / 0.4 / 0.55: 21.59 / 21.82 / 21.47 / 21.43 / 21.45 ms per step (one box)
TN_BIG_ROUNDS = float(os.environ.get("MPHSIR_TN_ROUNDS", "1.0"))        # ... and aim for this many full rounds of resident workgroups (re-measured with the partial sums deferred: 0.5 / 0.75 / 1 / 2 -> 22.48 / 22.33 / 22.37 / 22.47 ms per step)


def gemm_tn(a, b, nsplit=None, colsum=False, tile128=None, immediate=False, reduce=True):
    """sum over tokens of a[m,:]^T b[m,:].  a (M,N1), b (M,N2) row-major views -> fp32 (N1,N2);
    batched: a (Bt,M,N1), b (Bt,M,N2) -> (Bt,N1,N2).  colsum=True also returns sum_m a[m,:] (fp32, (N1,)).
    reduce=False returns the raw split partials (Bt,nsplit,N1,N2) [and (Bt,nsplit,N1)] for reduce_block."""
    lib = _lib.load()
    _check(a, b)
    batched = a.dim() == 3
    Bt = a.shape[0] if batched else 1
    M, N1, N2 = a.shape[-2], a.shape[-1], b.shape[-1]
    assert a.stride(-1) == 1 and b.stride(-1) == 1 and b.shape[-2] == M and a.dtype == b.dtype
    if tile128 is None:
        tile128 = TN_BIG_TILES and a.dtype in _HALF
    form = TN_FORM or (2 if (N1 <= 128 and N2 <= 128 and not batched and M >= 65536) else 1)
    if nsplit is None:
        if tile128 and a.dtype in _HALF:     # transposed-read kernel: 64/128-wide tile per operand, 2 workgroups per CU
            tiles = ((N1 + 127) // 128 if N1 > 64 else 1) * ((N2 + 127) // 128 if N2 > 64 else 1) * Bt
            if form == 2:                    # ring form: one 512-thread workgroup per CU
                nsplit = max(1, min(M // 256, 128, max(1, TN_RING_WGS // tiles)))
            else:
                resident = 2 if (N1 > 64 and N2 > 64) else (4 if (N1 <= 64 and N2 <= 64) else 3)    # workgroups per CU (LDS, VGPRs)
                nsplit = max(1, min(M // 256, 128, max(1, int(256 * resident * TN_BIG_ROUNDS) // tiles)))
        else:
            ts = 128 if tile128 else 64
            tiles = ((N1 + ts - 1) // ts) * ((N2 + ts - 1) // ts) * Bt
            nsplit = max(1, min(M // 512, 64, max(1, 768 // tiles)))      # ~3 workgroups per CU (measured optimum)
        if TN_PART_CAP > 0:                      # partial bytes <= TN_PART_CAP x input bytes (small-M problems: the partials ARE the traffic)
            cap = int(TN_PART_CAP * M * (N1 + N2) * a.element_size() / (N1 * N2 * 4.0))
            nsplit = max(1, (M + 4095) // 4096, min(nsplit, cap))
    part = torch.empty((Bt, nsplit, N1, N2), dtype=torch.float32, device=a.device)
    cs = torch.empty((Bt, nsplit, N1), dtype=torch.float32, device=a.device) if colsum else None
    if _SCOPE is not None and TN_GROUPED and not immediate and not batched and tile128 and a.dtype in _HALF:
        # nothing but the partial reduction at the end of the scope reads the result: issue it there, grouped
        _SCOPE.gemms.append(dict(A=a.data_ptr(), lda=a.stride(-2), B=b.data_ptr(), ldb=b.stride(-2), Cpart=part.data_ptr(),
                                 cs=cs.data_ptr() if colsum else None, M=M, N1=N1, N2=N2, nsplit=nsplit, form=form,
                                 keep=(a, b, part, cs)))
    else:
        _lib.check(lib.mphsir_gemm_tn(_p(a), a.stride(-2), a.stride(0) if batched else 0, _p(b), b.stride(-2),
                                      b.stride(0) if batched else 0, _p(part), _p(cs), M, N1, N2, nsplit, Bt,
                                      (form if a.dtype in _HALF else 1) if tile128 else 0,
                                      _DT[a.dtype], _stream(a)),
                   "gemm_tn")
    # algorithmic bytes = the two token matrices, read once; the kernel's own split partials are overhead, accounted apart
    _acct("gemm_tn", 2.0 * Bt * M * N1 * N2, Bt * M * (N1 + N2) * a.element_size())
    _acct("gemm_tn:partials", 0.0, part.numel() * 4.0)
    if not reduce:
        return (part, cs) if colsum else part
    out = reduce_parts(part, batched=True, immediate=immediate)
    out = out if batched else out[0]
    if colsum:
        c = reduce_parts(cs, batched=True, immediate=immediate)
        return out, (c if batched else c[0])
    return out


def gemm_tn_blocks(a, b, row_ranges, ncols=None, colsum=False):
    """gemm_tn that keeps only the row ranges [(r0, nr), ...] (stacked in that order) and the first `ncols` columns of
    the (N1, N2) product: the un-padding of the hidden dimension happens in the partial reduction itself.
    -> fp32 (sum nr, ncols) [and the matching entries of the column sums of a, (sum nr,)]."""
    N2 = b.shape[-1]
    ncols = N2 if ncols is None else ncols
    res = gemm_tn(a, b, colsum=colsum, reduce=False)
    part, cs = (res if colsum else (res, None))
    rows = sum(nr for _, nr in row_ranges)
    out = torch.empty((rows, ncols), dtype=torch.float32, device=a.device)
    outc = torch.empty((1, rows), dtype=torch.float32, device=a.device) if colsum else None
    o = 0
    for r0, nr in row_ranges:
        reduce_block(part[0], r0, nr, 0, ncols, out[o:o + nr])
        if colsum:
            reduce_block(cs[0].unsqueeze(1), 0, 1, r0, nr, outc[:, o:o + nr])
        o += nr
    return (out, outc[0]) if colsum else out


def gdfn_gate_bwd(t, du):
    """t (M,2*HP), du (M,HP) contiguous -> (u (M,HP), dt (M,2*HP))."""
    lib = _lib.load()
    _check(t, du)
    M, HP = du.shape
    assert t.shape == (M, 2 * HP) and t.is_contiguous() and du.is_contiguous()
    u, dt_ = torch.empty_like(du), torch.empty_like(t)
    _lib.check(lib.mphsir_gdfn_gate_bwd(_p(t), _p(du), _p(u), _p(dt_), M, HP, _DT[t.dtype], _stream(t)), "gdfn_gate_bwd")
    _acct("gdfn_gate_bwd", 30.0 * M * HP, 7.0 * M * HP * t.element_size())
    return u, dt_


def dwconv_gate_bwd(t, w9, du, B, H, W):
    """Backward of u = gelu(x1) * x2, [x1|x2] = dwconv3x3(t), with the conv recomputed inside: t (M,2*HP), du (M,HP) contiguous ->
    (u (M,HP), d[x1|x2] (M,2*HP)).  Shapes the tile form covers; else dwconv3x3 + gdfn_gate_bwd."""
    lib = _lib.load()
    _check(t, w9, du)
    M, HP = du.shape
    assert t.shape == (M, 2 * HP) and t.is_contiguous() and du.is_contiguous() and M == B * H * W
    if not lib.mphsir_dwconv_gate_bwd_fits(H, W, HP, _DT[t.dtype]):
        return gdfn_gate_bwd(dwconv3x3(t.reshape(B, H, W, 2 * HP), w9).reshape(M, 2 * HP), du)
    u, dt_ = torch.empty_like(du), torch.empty_like(t)
    _lib.check(lib.mphsir_dwconv_gate_bwd(_p(t), _p(w9), w9.stride(0), _p(du), _p(u), _p(dt_), B, H, W, HP, _DT[t.dtype], _stream(t)), "dwconv_gate_bwd")
    _acct("gdfn_gate_bwd", 66.0 * M * HP, 6.0 * M * HP * t.element_size())
    return u, dt_


FOLD_BWD_DM_MAX_TOKENS = int(os.environ.get("MPHSIR_FOLD_BWD_DM_TOKENS", "1024"))      # per sample; 0: always the token-reduction GEMM


def fold_bwd_forms_dm(N, C, heads, dtype):
    """the fold backward can form dM = d_out^T v itself (small images: the lower pyramid levels, where the token-reduction GEMM in
    front of it is a 15-22 us launch on the critical path for 4-8 MB of operands)"""
    return dtype in _HALF and 0 < N <= FOLD_BWD_DM_MAX_TOKENS and N % 64 == 0 and (C // heads) in (32, 48, 64)      # (96-wide heads: the token tiles do not fit beside the 150 KB the kernel already takes)


GDFN_DW_BWD = os.environ.get("MPHSIR_GDFN_DW_BWD", "0") == "1"      # gate backward + depthwise backward of the GDFN in one launch: correct, 1 GB per step less traffic, but level in A/B and +50 us in the serial trace (803 against 751 us for the four GDFNs): OFF
GDFN_DW_BWD_WGS = int(os.environ.get("MPHSIR_GDFN_DW_BWD_WGS", "2048"))      # four rounds of resident workgroups (measured: 88 ranges x 22 slabs 333 us, 16 ranges 425)


def gdfn_dw_bwd_fits(H, W, HP, dtype):
    return GDFN_DW_BWD and dtype in _HALF and bool(_lib.load().mphsir_gdfn_dw_bwd_fits(H, W, HP, _DT[dtype]))


def gdfn_dw_bwd(t, w9, du, B, H, W, nblk=None, round_mid=False):
    """t (M, 2HP) = project_in(LN(x)), du (M, HP) contiguous, w9 fp32 [9][2HP] -> (u (M, HP), dt (M, 2HP), tap-gradient partials
    (nblk, 9, 2HP)): the conv is recomputed on every tile's halo, [d x1 | d x2] stays on the chip (include/mphsir.h)."""
    lib = _lib.load()
    _check(t, w9, du)
    M, HP = du.shape
    assert t.shape == (M, 2 * HP) and t.is_contiguous() and du.is_contiguous() and M == B * H * W and w9.shape == (9, 2 * HP) and w9.stride(1) == 1
    nslab = HP // 16
    tiles = B * (H // 8) * (W // 16)
    if nblk is None:
        nblk = max(1, min(tiles, GDFN_DW_BWD_WGS // nslab))
        if nblk >= 8:
            nblk = nblk // 8 * 8
    u = torch.empty((M, HP), dtype=t.dtype, device=t.device)
    dt = torch.empty((M, 2 * HP), dtype=t.dtype, device=t.device)
    part = torch.empty((nblk, 9, 2 * HP), dtype=torch.float32, device=t.device)
    _lib.check(lib.mphsir_gdfn_dw_bwd(_p(t), _p(w9), w9.stride(0), _p(du), _p(u), _p(dt), _p(part), nblk, B, H, W, HP, int(round_mid),
                                      _DT[t.dtype], _stream(t)), "gdfn_dw_bwd")
    _acct("gdfn_gate_bwd", M * HP * (40.0 + 72.0), (2.0 * 1.9 + 1.4 + 1.0 + 2.0) * M * HP * t.element_size())
    return u, dt, part


def spectral_fold_bwd(gp, sp, temperature, Wo, dM, dtype, reduce=True, d_out=None, v=None, dm_scale=None, w2_blocks=False):
    """-> W2 (B,2C,2C) in `dtype`, dWo (C,C) fp32, dtemp (heads,) fp32 (reduce=False: the per-sample partials
    (B,C,C) / (B,heads) instead, for the caller to pass to reduce_parts).  dM=None with d_out, v (B*N, C) in `dtype`: dM is formed in
    the kernel (fold_bwd_forms_dm)."""
    lib = _lib.load()
    B, nsplit, heads, hd, _ = gp.shape
    C = heads * hd
    if dM is None:
        _check(gp, sp, temperature, Wo, d_out, v)
        N = d_out.shape[0] // B
        assert d_out.shape == (B * N, C) and v.shape == (B * N, C) and d_out.dtype == dtype and v.dtype == dtype and d_out.stride(1) == 1 and v.stride(1) == 1
        assert fold_bwd_forms_dm(N, C, heads, dtype)
        W2 = torch.empty((B, 2 * C, 2 * C), dtype=dtype, device=gp.device)
        dWo = torch.empty((B, C, C), dtype=torch.float32, device=gp.device)
        dtemp = torch.empty((B, heads), dtype=torch.float32, device=gp.device)
        a = _lib.FoldBwdArgs()
        a.Gpart, a.Spart, a.temperature, a.Wo = _p(gp), _p(sp), _p(temperature), _p(Wo)
        a.W2, a.dWo, a.dtemp = _p(W2), _p(dWo), _p(dtemp)
        a.B, a.C, a.heads, a.nsplit, a.dM_nsplit = B, C, heads, nsplit, 0
        a.DO, a.lddo, a.V, a.ldv, a.N = _p(d_out), d_out.stride(0), _p(v), v.stride(0), N
        a.dm_scale = _p(dm_scale)
        a.w2_blocks = 1 if w2_blocks else 0
        _lib.check(lib.mphsir_spectral_fold_bwd(ctypes.byref(a), _DT[dtype], _stream(gp)), "spectral_fold_bwd")
        _acct("spectral_fold_bwd", 4.0 * B * C * C * hd + 2.0 * B * N * C * C, 3.0 * B * C * C * 4 + 2.0 * B * N * C * d_out.element_size())
        _acct("spectral_fold_bwd:dm", 0.0, 0.0)
        if not reduce:
            return W2, dWo, dtemp
        return W2, reduce_parts(dWo), reduce_parts(dtemp)
    _check(gp, sp, temperature, Wo, dM)
    # dM (B, C, C), or (B, splits, C, C): the raw split partials of the token-reduction GEMM (gemm_tn(..., reduce=False)) -- the kernel
    # sums them in split order while it stages them, so no ordered-sum launch sits between the two
    dm_nsplit = dM.shape[1] if dM.dim() == 4 else 1
    assert dM.shape[0] == B and dM.shape[-2:] == (C, C) and dM.dtype == torch.float32 and dM.is_contiguous()
    W2 = torch.empty((B, 2 * C, 2 * C), dtype=dtype, device=gp.device)
    dWo = torch.empty((B, C, C), dtype=torch.float32, device=gp.device)
    dtemp = torch.empty((B, heads), dtype=torch.float32, device=gp.device)
    a = _lib.FoldBwdArgs()
    a.Gpart, a.Spart, a.temperature, a.Wo, a.dM = _p(gp), _p(sp), _p(temperature), _p(Wo), _p(dM)
    a.W2, a.dWo, a.dtemp = _p(W2), _p(dWo), _p(dtemp)
    a.B, a.C, a.heads, a.nsplit, a.dM_nsplit = B, C, heads, nsplit, dm_nsplit
    a.dm_scale = _p(dm_scale)
    a.w2_blocks = 1 if w2_blocks else 0
    _lib.check(lib.mphsir_spectral_fold_bwd(ctypes.byref(a), _DT[dtype], _stream(gp)), "spectral_fold_bwd")
    _acct("spectral_fold_bwd", 4.0 * B * C * C * hd, 3.0 * B * C * C * 4)
    if not reduce:
        return W2, dWo, dtemp
    return W2, reduce_parts(dWo), reduce_parts(dtemp)


def pg_gate_bwd(mu, dgate, pg, factor_dtype=torch.float32):
    """mu, dgate (nW,C) fp32; pg = fp32 parameter dict (as for win_attn_fwd) -> dmu (nW,C) and the dict of
    parameter gradients of the local spectral-prompt branch."""
    lib = _lib.load()
    _check(mu, dgate, *pg.values())
    nW, C = mu.shape
    r = pg["linear_down.weight"].shape[0]
    KL, KR = round_up(C + 5 * r + 256, 8), round_up(5 * r + 1 + C, 8)
    dmu = torch.empty_like(mu)
    # factor rows in the compute dtype: their product then rides in the grouped bf16 GEMM launch of the backward
    L = torch.empty((nW, KL), dtype=factor_dtype, device=mu.device)
    R = torch.empty((nW, KR), dtype=factor_dtype, device=mu.device)
    a = _lib.PgBwdArgs()
    a.mu, a.dgate = _p(mu), _p(dgate)
    a.Wprompt, a.prompt_param = _p(pg["linear_prompt.weight"]), _p(pg["prompt_param"])
    a.Wq, a.Wkv, a.Wdown = _p(pg["q.weight"]), _p(pg["kv.weight"]), _p(pg["linear_down.weight"])
    a.Wpproj, a.bpproj, a.Wup = _p(pg["proj.weight"]), _p(pg["proj.bias"]), _p(pg["linear_up.weight"])
    a.dmu, a.L, a.R = _p(dmu), _p(L), _p(R)
    a.nW, a.C, a.r, a.KL, a.KR, a.lr_bf16 = nW, C, r, KL, KR, _DT[factor_dtype]
    _lib.check(lib.mphsir_pg_gate_bwd(ctypes.byref(a), _stream(mu)), "pg_gate_bwd")
    _acct("pg_gate_bwd", 4.0 * nW * C * (128 + 2 * r), 4.0 * nW * (2 * C + KL + KR))
    part = gemm_tn(L, R, reduce=False)[0]                      # (nsplit, KL, KR): every parameter gradient is a sub-block
    blocks = {
        "linear_up.weight": (0, C, 0, r),
        "proj.weight": (C, r, r, r),
        "proj.bias": (C, r, 2 * r, 1),
        "kv.weight": (C + r, 2 * r, 2 * r + 1, r),
        "q.weight": (C + 3 * r, r, 3 * r + 1, r),
        "prompt_param": (C + 4 * r, 128, 4 * r + 1, r),
        "linear_prompt.weight": (C + 4 * r + 128, 128, 5 * r + 1, C),
        "linear_down.weight": (C + 4 * r + 256, r, 5 * r + 1, C),
    }
    g = {k: reduce_block(part, r0, nr, c0, nc, torch.empty((nr, nc), dtype=torch.float32, device=mu.device))
         for k, (r0, nr, c0, nc) in blocks.items()}
    g["proj.bias"] = g["proj.bias"].reshape(r)
    return dmu, g


def pack_conv3x3(w, dtype, flip_transpose=False):
    """conv weight (Cout,Cin,3,3) -> [Np][9*Cp] tap-major, channels zero-padded (Cin -> mult of 32, Cout -> mult of 16).
    flip_transpose=True packs the weights of the input-gradient convolution instead (in/out swapped, taps flipped)."""
    if flip_transpose:
        w = w.flip(2, 3).transpose(0, 1)
    Co, Ci = w.shape[0], w.shape[1]
    Np, Cp = round_up(Co, 16), round_up(Ci, 32)
    out = torch.zeros((Np, 9, Cp), dtype=cdt(dtype), device=w.device)
    out[:Co, :, :Ci] = w.permute(0, 2, 3, 1).reshape(Co, 9, Ci).to(cdt(dtype))
    return out.reshape(Np, 9 * Cp)


def pixel_pitch(t):
    """row pitch (elements) of a channels-last (B,H,W,C) view whose pixels are equally spaced rows -- a contiguous tensor or a channel slice
    of one -- else None"""
    B, H, W, C = t.shape
    ld = t.stride(2)
    ok = t.stride(3) == 1 and ld >= C and t.stride(1) == W * ld and t.stride(0) == H * W * ld and (ld * t.element_size()) % 16 == 0 and t.data_ptr() % 16 == 0
    return ld if ok else None


def conv3x3_tok(x, wp, out=None):
    """x (B,H,W,Cp) channels-last with Cp % 32 == 0 (contiguous, or a channel slice of a wider buffer read through its row pitch); wp from
    pack_conv3x3 -> (B,H,W,Np), written into `out` (same kind of view) when given."""
    lib = _lib.load()
    _check(x, wp, out)
    B, H, W, Cp = x.shape
    Np = wp.shape[0]
    ldx = pixel_pitch(x)
    assert ldx is not None and wp.shape[1] == 9 * Cp and wp.dtype == x.dtype
    y = out if out is not None else torch.empty((B, H, W, Np), dtype=x.dtype, device=x.device)
    ldy = pixel_pitch(y)
    assert ldy is not None and y.shape == (B, H, W, Np) and y.dtype == x.dtype
    _lib.check(lib.mphsir_conv3x3_tok(_p(x), ldx, _p(wp), _p(y), ldy, B, H, W, Cp, Np, _DT[x.dtype], _stream(x)), "conv3x3_tok")
    _acct("conv3x3_tok", 18.0 * B * H * W * Cp * Np, (B * H * W * (Cp + Np) + wp.numel()) * x.element_size())
    return y


def conv3x3_wgrad(dy2, x, nsplit=None, cout=None, cin=None):
    """dy2 (M, Np) [Np % 8 == 0], x (B,H,W,Cp) contiguous [Cp % 8 == 0], 16-bit: the weight gradient of a dense 3x3 conv with the
    im2col gather inside the token-reduction GEMM.  -> fp32 (Np, 9*Cp) in (tap, channel) column order; with cout / cin given:
    the nn.Conv2d layout (cout, cin, 3, 3) -- the padded rows / channels dropped and (tap, channel) transposed inside the ordered
    partial reduction itself (no slice / permute / contiguous launches; deferred like every parameter-gradient sum)."""
    lib = _lib.load()
    _check(dy2, x)
    B, H, W, Cp = x.shape
    M, Np = dy2.shape
    assert x.is_contiguous() and dy2.stride(1) == 1 and M == B * H * W and x.dtype == dy2.dtype and x.dtype in _HALF
    if nsplit is None:
        tiles = ((Np + 127) // 128 if Np > 64 else 1) * ((9 * Cp + 127) // 128)
        nsplit = max(1, min(M // 256, 128, max(1, (TN_RING_WGS if TN_FORM == 2 else int(256 * 2 * TN_BIG_ROUNDS)) // tiles)))
    part = torch.empty((1, nsplit, Np, 9 * Cp), dtype=torch.float32, device=x.device)
    _lib.check(lib.mphsir_conv3x3_wgrad(_p(dy2), dy2.stride(0), _p(x), Cp, _p(part), B, H, W, Np, Cp, nsplit, TN_FORM or 1, _DT[x.dtype],
                                        _stream(x)), "conv3x3_wgrad")
    _acct("gemm_tn", 2.0 * M * Np * 9 * Cp, M * (Np + Cp) * x.element_size())
    _acct("gemm_tn:partials", 0.0, part.numel() * 4.0)
    if cout is None:
        return reduce_parts(part, batched=True, immediate=True)[0]
    assert cout <= Np and cin <= Cp
    out = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    # one segment: batch = output channel, rows = taps, columns = input channels; out[co][ci][tap]: row pitch 1, column stride 9
    _submit(dict(src=part.data_ptr(), dst=out.data_ptr(), n=cin, stride=part.stride(1), sbs=9 * Cp, dbs=cin * 9, nsplit=part.shape[1], nbatch=cout,
                 rows=9, dcs=9, src_ld=Cp, dst_ld=1, keep=(part, out)), False)
    return out


def im2col3x3(x):
    """x (B,H,W,Cp) -> (B*H*W, 9*Cp) gathered neighbourhoods (zero padding)."""
    lib = _lib.load()
    _check(x)
    B, H, W, Cp = x.shape
    assert x.is_contiguous()
    col = torch.empty((B * H * W, 9 * Cp), dtype=x.dtype, device=x.device)
    _lib.check(lib.mphsir_im2col3x3(_p(x), Cp, _p(col), B, H, W, Cp, _DT[x.dtype], _stream(x)), "im2col3x3")
    _acct("im2col3x3", 0.0, 10.0 * x.numel() * x.element_size())
    return col
