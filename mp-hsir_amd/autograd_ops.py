"""autograd wiring of the fused ops.  Forward AND backward of every module are HIP kernels called through the C ABI
(include/mphsir.h): the backward functions below only sequence launches -- block / prompt-module data gradients, and
every parameter gradient as a token-reduction GEMM (gemm_tn), column sum or ordered partial reduction.  There is no
PyTorch-composite or library-GEMM path: a shape the kernels do not cover raises."""
import torch
import torch.nn.functional as F

from . import ops


class SkipGrad:
    """One BaseBlock's skip gradient on its way from the backward of the last block (where `+ x` was added, ref :760) to the backward of
    the first block (whose input x is): the first block's final launch adds it to dx, and autograd sees no second gradient for x."""

    def __init__(self):
        self.dz = None

    def take(self):
        dz, self.dz = self.dz, None
        return dz


def _skip_put(skip, dz, needed):
    """the res-gradient a last block returns: handed to the holder when there is one"""
    if not needed:
        return None
    if skip is not None and skip[2]:
        skip[0].dz = dz
        return None
    return dz


def _skip_take(skip):
    return skip[0].take() if (skip is not None and skip[1]) else None


class _GatedMlp(torch.autograd.Function):
    """z = y + keep*mlp(LN2(y)): HIP forward and HIP data-gradient; the parameter gradients are the
    token-reduction GEMMs / column sums of the three matrices the backward kernel writes."""

    @staticmethod
    def forward(ctx, blk, k2, y, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b, res=None, skip=None):
        """res: optional second residual (the input of the enclosing BaseBlock, ref :727-761), added by the same launch"""
        B, H, W, Cc = y.shape
        pk = blk.packed(y.dtype)
        ctx.blk, ctx.k2, ctx.skip = blk, k2, skip
        ctx.save_for_backward(y)
        z = ops.gated_mlp_fwd(y.reshape(-1, Cc), pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"], pk["W2"], pk["b2"],
                              keep=k2, rows_per_batch=H * W, res=None if res is None else res.reshape(-1, Cc))
        return z.reshape(B, H, W, Cc)

    @staticmethod
    def backward(ctx, dz):
        (y,) = ctx.saved_tensors
        return (None, None) + _gated_mlp_backward(ctx.blk, ctx.k2, y, dz) + (_skip_put(ctx.skip, dz, ctx.needs_input_grad[9]), None)


def _gated_mlp_backward(blk, k2, y, dz):
    """backward of z = y + keep2 * mlp(LN2(y)): (dy, d ln2.weight, d ln2.bias, d fc1.weight, d fc1.bias, d fc2.weight, d fc2.bias)"""
    B, H, W, Cc = y.shape
    dt = y.dtype
    pk = blk.packed(dt)
    dz2 = dz.reshape(-1, Cc).contiguous()
    hid = blk.mlp.fc2.weight.shape[1]
    HP = pk["W2T"].shape[0]
    # 16-bit types, big launches: the parameter gradients are recomputed per hidden slab from LN(y) and dm alone
    # (ops.gated_mlp_wgrad); h and [dval | dgate] then never reach HBM
    fused = ops.gated_mlp_wgrad_fits(dz2.shape[0], Cc, HP, dt)
    if k2 is None:
        dm = dz2
        dx, xn, h, dpre, part = ops.gated_mlp_bwd(y.reshape(-1, Cc), dz2, dm, pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"],
                                                  pk["W1T"], pk["W2T"], operands=not fused)
    else:               # DropPath: dm = keep[b] * dz is formed (and kept for dW2 / db2) inside the kernel
        dx, xn, h, dpre, part, dm = ops.gated_mlp_bwd(y.reshape(-1, Cc), dz2, None, pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"],
                                                      pk["W1T"], pk["W2T"], keep=k2, rows_per_batch=H * W, operands=not fused)
    with ops.reduce_scope(leaf=True):                                   # one ordered-sum launch for all five partial buffers
        if fused:
            dW1, db1, dW2, db2 = ops.gated_mlp_wgrad(xn, dm, pk["W1"], pk["b1"], pk["W2T"], hid)
        else:
            dW2, db2 = ops.gemm_tn_blocks(dm, h, [(0, Cc)], ncols=hid, colsum=True)             # drops the padded hidden columns
            dW1, db1 = ops.gemm_tn_blocks(dpre, xn, [(0, hid), (HP, hid)], colsum=True)            # value rows | gate rows
        dln = ops.reduce_parts(part)
    return (dx.reshape(B, H, W, Cc), dln[0], dln[1], dW1, db1, dW2, db2)


def channel_attention_bwd(d_out, t_q, t_k, t_v, w9q, w9k, w9v, v, gp, sp, Mb, MbT, temperature, wo, heads, B, H, W, qk=None, keep=None):
    """Backward of the folded channel attention  out = M_b v,  M_b = Wo blockdiag(softmax(normalised Gram)).

    d_out (M,C); t_q/t_k/t_v: (B,H,W,*) views of the 1x1-conv outputs that fed dwconv_gram (channels-last,
    C channels each); w9*: fp32 tap-major dw weights [9][C] views; v, gp, sp, Mb, MbT as saved by the forward.
    keep (optional, (B,) fp32): d_out is handed over WITHOUT the DropPath factor of its block -- the fold backward scales dM and the fused
    launch scales dv by keep[b] (both are linear in d_out), so nothing has to form keep * d_out first (fused path only).
    Returns d(t_q), d(t_k), d(t_v) (B,H,W,C each), d(dw taps) (C,9) x3, d temperature (heads,), d Wo (C,C).
    All HIP: dM = d_out^T v (gemm_tn), fold backward (one launch), then [dq|dk], dv and the depthwise backward (data + taps) in ONE
    launch (ops.spectral_dqkv_bwd, 16-bit types, self-attention: q|k|v adjacent slices of one t) -- else per-sample token GEMMs +
    the depthwise backward kernels."""
    C = v.shape[1]
    N = H * W
    M = B * N
    dt = v.dtype
    temp32, wo32 = temperature.detach().reshape(heads).float().contiguous(), wo.detach().reshape(C, C).float().contiguous()
    joint = (t_q.data_ptr() + C * t_q.element_size() == t_k.data_ptr() and t_k.data_ptr() + C * t_k.element_size() == t_v.data_ptr()
             and t_q.stride() == t_k.stride() == t_v.stride() and w9q.data_ptr() + 4 * C == w9k.data_ptr()
             and w9k.data_ptr() + 4 * C == w9v.data_ptr())
    fused = joint and t_q.stride(2) == 3 * C and w9q.stride(0) == 3 * C and ops.spectral_dqkv_bwd_fits(C, heads, H, W, dt)
    # (the fused launch reads only each head's own column blocks of W2: the fold backward then writes nothing else -- 2 hd of 2C columns)
    if ops.fold_bwd_forms_dm(N, C, heads, dt):
        # small images (the lower pyramid levels): dM = d_out^T v is formed INSIDE the fold backward -- one launch fewer on the chain
        W2, dwo_p, dtemp_p = ops.spectral_fold_bwd(gp, sp, temp32, wo32, None, dt, reduce=False, d_out=d_out.reshape(M, C), v=v.reshape(M, C), dm_scale=keep,
                                                   w2_blocks=fused)
    else:
        dM = ops.gemm_tn(d_out.reshape(B, N, C), v.reshape(B, N, C), reduce=False)    # (B, splits, C, C) fp32 partials: summed by the fold backward as it stages them
        W2, dwo_p, dtemp_p = ops.spectral_fold_bwd(gp, sp, temp32, wo32, dM, dt, reduce=False, dm_scale=keep, w2_blocks=fused)
    # q, k of the forward: either kept by it (qk) or recomputed by the depthwise kernel; when q|k|v are adjacent channel
    # slices of one tensor (self-attention) every depthwise pass runs once over the joint channel range.
    if qk is not None:
        pass                                                    # kept by the forward (dwconv_gram keep_qk)
    elif joint:
        qk = ops.dwconv3x3(torch.as_strided(t_q, (B, H, W, 2 * C), t_q.stride()), torch.as_strided(w9q, (9, 2 * C), w9q.stride()))
    else:                                                       # cross attention: q and k come from different tensors -- both
        qk = torch.empty((B, H, W, 2 * C), dtype=dt, device=v.device)      # depthwise outputs go straight into the halves of [q | k]
        ops.dwconv3x3(t_q, w9q, out=qk[..., :C])
        ops.dwconv3x3(t_k, w9k, out=qk[..., C:])
    if fused:
        # ONE launch from the fold backward's matrix to dt: dv = d_out M_b and [dq | dk] = [q | k] W2^T are formed per halo tile on the
        # matrix cores and fed to the depthwise backward in LDS -- [dq | dk | dv] (3C per token) is neither written nor read back
        dwo, dtemp = ops.reduce_parts(dwo_p), ops.reduce_parts(dtemp_p)
        t_all = torch.as_strided(t_q, (M, 3 * C), (3 * C, 1))
        w9_all = torch.as_strided(w9q, (9, 3 * C), w9q.stride())
        dt_all, dw_all = ops.spectral_dqkv_bwd(qk.reshape(M, 2 * C), d_out, t_all, W2, MbT, w9_all, B, H, W, C, heads, vscale=keep)
        dt_all = dt_all.reshape(B, H, W, 3 * C)
        return (dt_all[..., :C], dt_all[..., C:2 * C], dt_all[..., 2 * C:], dw_all[:C], dw_all[C:2 * C], dw_all[2 * C:], dtemp, dwo)
    assert keep is None, "the unscaled d_out form needs the fused launch (ops.spectral_dqkv_bwd_fits)"
    dall = torch.empty((M, 3 * C), dtype=dt, device=v.device)
    ops.gemm_tok(d_out, MbT, out=dall[:, 2 * C:])                                     # dv = d_out M_b
    dwo, dtemp = ops.reduce_parts(dwo_p), ops.reduce_parts(dtemp_p)
    ops.gemm_tok(qk.reshape(M, 2 * C), W2, out=dall[:, :2 * C])                       # [dq | dk]
    dall4 = dall.reshape(B, H, W, 3 * C)
    if joint:
        t_all = torch.as_strided(t_q, (B, H, W, 3 * C), t_q.stride())
        w9_all = torch.as_strided(w9q, (9, 3 * C), w9q.stride())
        # both gradients of the depthwise conv in one launch (dall read once); dw_all (3C, 9): the parameter's layout
        dt_all, dw_all = ops.dwconv3x3_bwd(t_all, dall4, w9_all, col_ranges=[(0, 3 * C)])
        return (dt_all[..., :C], dt_all[..., C:2 * C], dt_all[..., 2 * C:], dw_all[:C], dw_all[C:2 * C], dw_all[2 * C:],
                dtemp, dwo)
    dq4, dk4, dv4 = dall4[..., :C], dall4[..., C:2 * C], dall4[..., 2 * C:]
    # the three tap gradients are row ranges of ONE buffer: the callers join them as a view (a cat would READ sums that may
    # not have been launched yet, ops.deferred_reductions)
    dw3 = torch.empty((3 * C, 9), dtype=torch.float32, device=v.device)
    # d(t_k), d(t_v) as the halves of ONE buffer: the cross attention's caller takes them as [d t_k | d t_v] without a cat
    dkv = torch.empty((B, H, W, 2 * C), dtype=dt, device=v.device)
    return (ops.dwconv3x3(dq4, w9q, flip=True), ops.dwconv3x3(dk4, w9k, flip=True, out=dkv[..., :C]), ops.dwconv3x3(dv4, w9v, flip=True, out=dkv[..., C:]),
            ops.dwconv3x3_wgrad(t_q, dq4, col_ranges=[(0, C)], out=dw3[:C]), ops.dwconv3x3_wgrad(t_k, dk4, col_ranges=[(0, C)], out=dw3[C:2 * C]),
            ops.dwconv3x3_wgrad(t_v, dv4, col_ranges=[(0, C)], out=dw3[2 * C:]), dtemp, dwo)


import os
# MPHSIR_COMBINE_SIDE=1 (round-6 experiment, OFF): the branch sum's backward leaves the chain -- the spectral kernels take dy itself (the fold
# backward scales dM, the fused launch scales dv by the DropPath factor: keep * dy is never formed) and combine_bwd runs on the prompt gate's
# side branch.  Measured level to slower (19.50-19.60 against 19.44-19.46 ms, three pairs on one box): like every other branch of the captured
# step it hides nothing, and the d_out store it saves was hidden already.
COMBINE_SIDE = os.environ.get("MPHSIR_COMBINE_SIDE", "0") == "1"


_PG_KEYS = ("linear_down.weight", "linear_up.weight", "linear_prompt.weight", "prompt_param", "q.weight", "kv.weight",
            "proj.weight", "proj.bias")


def _pass_a_infer(x2, wqkv, w9, B, H, W, C, heads, ln=None):
    """pass A of the channel attention without a backward: (v, Gram partials, sum-of-squares partials).  One fused launch
    (csrc/spectral_fused.hip) where the shape is covered, else the two-kernel path without its training outputs."""
    if ops.qkv_dwconv_gram_fits(C, heads, H, W, x2.dtype):
        v, gp, sp, _ = ops.qkv_dwconv_gram(x2, wqkv, w9, B, H, W, C, heads, ln=ln)
    else:
        t = ops.gemm_tok(x2, wqkv, ln=ln)
        v, gp, sp, _ = ops.dwconv_gram(t[:, :C], t[:, C:2 * C], t[:, 2 * C:], w9[:, :C], w9[:, C:2 * C], w9[:, 2 * C:], 3 * C, B, H, W, C, heads)
    return v, gp, sp


def _pgsstb_attn_infer(blk, k1, x, fuse=False):
    """First residual branch of a PGSSTB block under no_grad: nothing is kept for a backward, and the 1x1 qkv conv, the
    depthwise conv and the Gram of the global spectral branch run as ONE launch (t = qkv(sa) never reaches HBM).
    fuse: return the operands of the branch sum (ops.gated_mlp_fwd branch=...) instead of launching it."""
    B, H, W, Cc = x.shape
    dt = x.dtype
    pk = blk.packed(dt)
    sp = blk.gobal_spectral_attn.packed(dt)
    heads, shift = blk.num_heads, blk.shift_size
    sa, mu, _ = ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"], pk["bproj"],
                                 pk["pg"], heads, shift, gate=False)
    sa2 = sa.reshape(-1, Cc)
    v, gp, spart = _pass_a_infer(sa2, sp["wqkv"], sp["w9"], B, H, W, Cc, heads)
    # the prompt gate forks AFTER pass A is issued (beside the fused pass A, whose workgroups fill the LDS of every CU, it costs
    # 2 % of the 512x512 forward): it runs beside the partial-sum reduction and the fold, three small latency-bound launches
    # (7.62 -> 7.58 ms)
    with ops.side_stream(sa, ops.SIDE_BRANCH) as br:
        gate = ops.pg_gate_fwd(mu, pk["pg"])
    Mb = ops.spectral_fold(gp, spart, sp["temp"], sp["wo"], dt)
    br.join()
    if fuse:
        return dict(v=v, Mb=Mb, sa=sa2, gate=gate, keep=k1, geom=(H, W, shift))
    y = ops.gemm_tok(v, Mb, epi=2, res=x.reshape(-1, Cc), sa=sa2, gate=gate, keep=k1, geom=(H, W, shift))
    return y.reshape(B, H, W, Cc)


def _self_channel_attn_infer(attn, ln, geom, t2):
    """a = t + M_b v (ref Attention :289-322 inside :476) under no_grad: fused pass A with the LayerNorm prologue."""
    B, H, W = geom
    pa = attn.packed(t2.dtype)
    v, gp, sp = _pass_a_infer(t2, pa["wqkv"], pa["w9"], B, H, W, t2.shape[1], attn.num_heads, ln=ln.pair())
    return ops.gemm_tok(v, ops.spectral_fold(gp, sp, pa["temp"], pa["wo"], t2.dtype), epi=1, res=t2)


def _pgsstb_attn_forward(blk, k1, x, fuse=False):
    """forward of the first residual branch; returns (y, tensors kept for the backward).  fuse: y is left to the gated-MLP launch
    (ops.gated_mlp_fwd branch=...: the operands are saved[6] = v, saved[9] = Mb, saved[1] = sa, saved[2] = gate) and None is returned."""
    B, H, W, Cc = x.shape
    dt = x.dtype
    pk = blk.packed(dt)
    sp = blk.gobal_spectral_attn.packed(dt)
    heads, shift = blk.num_heads, blk.shift_size
    w9 = sp["w9"]
    sa, mu, oattn = ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"],
                                     pk["bproj"], pk["pg"], heads, shift, save=True, gate=False)
    with ops.side_stream(sa, ops.SIDE_BRANCH) as br:      # the prompt gate (few workgroups, latency-bound) runs beside pass A
        gate = ops.pg_gate_fwd(mu, pk["pg"])
    sa2 = sa.reshape(-1, Cc)
    # pass A; t and q | k after the depthwise conv are kept for the backward (q|k: 2C values per token, cheaper than
    # recomputing them).  One fused launch where the kernel covers the shape, else 1x1 GEMM + depthwise/Gram kernel.
    if ops.FUSED_TRAIN and ops.qkv_dwconv_gram_fits(Cc, heads, H, W, dt):
        v, gp, spart, _, t, qk = ops.qkv_dwconv_gram(sa2, sp["wqkv"], w9, B, H, W, Cc, heads, keep=True)
    else:
        t = ops.gemm_tok(sa2, sp["wqkv"])
        v, gp, spart, _, qk = ops.dwconv_gram(t[:, :Cc], t[:, Cc:2 * Cc], t[:, 2 * Cc:], w9[:, :Cc], w9[:, Cc:2 * Cc], w9[:, 2 * Cc:],
                                              3 * Cc, B, H, W, Cc, heads, keep_qk=True)
    Mb, MbT, gp, spart = ops.spectral_fold(gp, spart, sp["temp"], sp["wo"], dt, transposed=True)   # keep the sums, drop the partials
    br.join()
    saved = (x, sa, gate, mu, oattn, t, v, gp, spart, Mb, MbT) + ((qk,) if qk is not None else ())
    if fuse:
        return None, saved
    y = ops.gemm_tok(v, Mb, epi=2, res=x.reshape(-1, Cc), sa=sa2, gate=gate, keep=k1, geom=(H, W, shift))
    return y.reshape(B, H, W, Cc), saved


def _pgsstb_attn_backward(blk, k1, saved, dy, extra=None):
    """extra: a gradient of x from outside the block (SkipGrad), added to dx by the last launch.
    backward of the first residual branch: (dx, d norm1.weight, d norm1.bias, d qkv.weight, d qkv.bias, d proj.weight, d proj.bias,
    d rpb table, d temperature, d spectral qkv, d spectral dw, d project_out, *d prompt-gate parameters (_PG_KEYS))"""
    x, sa, gate, mu, oattn, t, v, gp, spart, Mb, MbT = saved[:11]
    qk = saved[11] if len(saved) > 11 else None
    B, H, W, Cc = x.shape
    dt = x.dtype
    M = B * H * W
    heads, shift = blk.num_heads, blk.shift_size
    pk = blk.packed(dt)
    sp = blk.gobal_spectral_attn.packed(dt)
    dy = dy.contiguous()
    with ops.reduce_scope(leaf=True):      # every split partial of this backward is summed by ONE launch when the scope exits
        # (1) branch sum  y = x + keep*(sa*gate + out).  Where the fused spectral backward applies, the branch sum's backward leaves the
        # chain: the spectral kernels take dy itself (the fold backward scales dM, the fused launch scales dv by keep[b]: d_out = keep * dy
        # is never formed), and combine_bwd -- now only d_sa = keep dy gate and the gate's gradient -- runs on the side branch in front of
        # the prompt gate's backward, joined where the 1x1 conv's data gradient adds to d_sa.
        f32_factors = dt == torch.float16 or mu.shape[0] <= 512
        lean = COMBINE_SIDE and ops.spectral_dqkv_bwd_fits(Cc, heads, H, W, dt) and sp["w9"].stride(0) == 3 * Cc and t.is_contiguous()
        if lean:
            with ops.side_stream(dy, ops.SIDE_BRANCH) as br:
                _, d_sa, dgate = ops.combine_bwd(dy, sa, gate, k1, shift, want_dout=False)
                dmu, gpg = ops.pg_gate_bwd(mu, dgate, pk["pg"], factor_dtype=torch.float32 if f32_factors else dt)
            d_out, keep_s = dy, k1
        else:
            d_out, d_sa, dgate = ops.combine_bwd(dy, sa, gate, k1, shift)
            keep_s = None
            # (3, issued first on a side branch) local spectral-prompt gate: one launch per block + one token-reduction GEMM
            # over the windows.  Factor rows in the compute dtype ride in the grouped 16-bit GEMM launch; fp16's narrow exponent
            # would flush the gate's tiny d-logits (w ~ 1/128 of an already small gradient), so that path keeps them in fp32
            with ops.side_stream(dy, ops.SIDE_BRANCH) as br:
                # ... and so do levels with few windows (the latent level: 4 per sample), where a parameter gradient is the sum of a
                # few hundred signed terms: with bf16 factor rows linear_down / kv of a latent block came out 23 % off in the whole-net
                # check (the reference's own bf16 autocast: 2 %); the fp32 product of <= 512 rows costs nothing
                dmu, gpg = ops.pg_gate_bwd(mu, dgate, pk["pg"], factor_dtype=torch.float32 if f32_factors else dt)
        # (2) global spectral attention
        t4 = t.reshape(B, H, W, 3 * Cc)
        w9 = sp["w9"]
        tq, tk, tv = t4[..., :Cc], t4[..., Cc:2 * Cc], t4[..., 2 * Cc:]
        dtq, dtk, dtv, dwq, dwk, dwv, dtemp, dwo = channel_attention_bwd(
            d_out.reshape(M, Cc), tq, tk, tv, w9[:, :Cc], w9[:, Cc:2 * Cc], w9[:, 2 * Cc:], v, gp, spart, Mb, MbT,
            blk.gobal_spectral_attn.temperature, blk.gobal_spectral_attn.project_out.weight, heads, B, H, W, qk=qk, keep=keep_s)
        if dtq.data_ptr() + Cc * dtq.element_size() == dtk.data_ptr() and dtq.stride(2) == 3 * Cc:
            dt3 = torch.as_strided(dtq, (M, 3 * Cc), (3 * Cc, 1))
        else:
            dt3 = torch.cat([dtq, dtk, dtv], dim=-1).reshape(M, 3 * Cc)
        if lean:
            br.join()                     # d_sa = keep dy gate comes from the side branch
        d_sa = ops.gemm_tok(dt3, sp["wqkvT"], epi=1, res=d_sa.reshape(M, Cc))    # + dt Wqkv  (1x1 conv backward)
        d_sqkv = ops.gemm_tn(dt3, sa.reshape(M, Cc)).reshape(3 * Cc, Cc, 1, 1)
        dpg = tuple(gpg[k].reshape(getattr_path(blk.local_spectral_attn, k).shape) for k in _PG_KEYS)
        if not lean:
            br.join()
        # (4) window attention core
        dqkv, xnw, dsat, drpb = ops.win_attn_bwd(x, d_sa.reshape(B, H, W, Cc), dmu, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"],
                                                 pk["rpb"], pk["wprojT"], heads, shift)
        d_qkv_w, d_qkv_b = ops.gemm_tn(dqkv, xnw, colsum=True)
        dsat2 = dsat.reshape(M, Cc)
        d_proj_w, d_proj_b = ops.gemm_tn(dsat2, oattn.reshape(M, Cc), colsum=True)
        # (5) d_xn = dqkv Wqkv, then norm1 backward and the residual path.  (The LayerNorm backward as an epilogue of that GEMM
        # was built and measured slower in round 4 -- four barriers and an fp32 staging tile per GEMM tile against a 2 C per token
        # round trip -- and removed in round 5.)
        if ops.ln_bwd_win_dxn_fits(M, Cc, dt):      # d_xn = dqkv Wqkv formed inside the LayerNorm-backward launch
            dx, part = ops.ln_bwd_win_dxn(x, dqkv, pk["wqkvT"], dy, pk["ln1"][0], shift, dres2=None if extra is None else extra.contiguous())
        else:
            dx, part = ops.ln_bwd_win(x, ops.gemm_tok(dqkv, pk["wqkvT"]), dy, pk["ln1"][0], shift)
            if extra is not None:
                dx = dx + extra
        dln = ops.reduce_parts(part)
        drpb = ops.reduce_parts(drpb)
    d_sdw = _join_taps(dwq, dwk, dwv).reshape(3 * Cc, 1, 3, 3)
    return (dx, dln[0], dln[1], d_qkv_w, d_qkv_b, d_proj_w, d_proj_b, drpb,
            dtemp.reshape(heads, 1, 1), d_sqkv, d_sdw, dwo.reshape(Cc, Cc, 1, 1)) + tuple(dpg)




class _PgsstbAttn(torch.autograd.Function):
    """First residual branch of a PGSSTB block: HIP forward (5 launches) and HIP backward."""

    @staticmethod
    def forward(ctx, blk, k1, skip, x, n1w, n1b, qkv_w, qkv_b, proj_w, proj_b, rpb, s_temp, s_qkv, s_dw, s_out, *pg):
        y, saved = _pgsstb_attn_forward(blk, k1, x)
        ctx.blk, ctx.k1, ctx.skip = blk, k1, skip
        ctx.save_for_backward(*saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        return (None, None, None) + _pgsstb_attn_backward(ctx.blk, ctx.k1, ctx.saved_tensors, dy, extra=_skip_take(ctx.skip))


_N_ATTN_PARAMS = 11 + len(_PG_KEYS)


class _Pgsstb(torch.autograd.Function):
    """A whole PGSSTB block with the branch sum y = x + keep1 (sa gate + v Mb^T) formed INSIDE the gated-MLP launch (16-bit types,
    C <= 128, big launches: ops.gated_mlp_fuses): one launch and one read of y fewer than _PgsstbAttn + _GatedMlp; y is written once,
    for the backward.  The backward is the two halves' backward, in sequence."""

    @staticmethod
    def forward(ctx, blk, k1, k2, res, skip, x, *params):
        B, H, W, Cc = x.shape
        _, saved = _pgsstb_attn_forward(blk, k1, x, fuse=True)
        pk = blk.packed(x.dtype)
        z, y = ops.gated_mlp_fwd(x.reshape(-1, Cc), pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"], pk["W2"], pk["b2"], keep=k2, rows_per_batch=H * W,
                                 res=None if res is None else res.reshape(-1, Cc),
                                 branch=dict(v=saved[6], Mb=saved[9], sa=saved[1].reshape(-1, Cc), gate=saved[2], keep=k1,
                                             geom=(H, W, blk.shift_size), want_y=True))
        ctx.blk, ctx.k1, ctx.k2, ctx.skip = blk, k1, k2, skip
        ctx.save_for_backward(y.reshape(B, H, W, Cc), *saved)
        return z.reshape(B, H, W, Cc)

    @staticmethod
    def backward(ctx, dz):
        y, saved = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        d_res = _skip_put(ctx.skip, dz, ctx.needs_input_grad[3])       # (before the take: a one-block BaseBlock is first and last)
        gm = _gated_mlp_backward(ctx.blk, ctx.k2, y, dz)
        ga = _pgsstb_attn_backward(ctx.blk, ctx.k1, saved, gm[0], extra=_skip_take(ctx.skip))
        return (None, None, None, d_res, None) + ga + gm[1:]


def _join_taps(*dw):
    """(C,9) tap gradients of adjacent channel ranges -> (sum C, 9); a view when they already are row slices of one buffer."""
    w0 = dw[0]
    ok, off = True, w0.data_ptr()
    for w in dw:
        ok = (ok and w.data_ptr() == off and w.is_contiguous()
              and w.untyped_storage().data_ptr() == w0.untyped_storage().data_ptr())
        off += w.numel() * w.element_size()
    if ok:
        return torch.as_strided(w0, (sum(w.shape[0] for w in dw), 9), (9, 1))
    ops.flush_deferred()            # the cat reads the sums: they must have been launched
    return torch.cat(dw, dim=0)


def getattr_path(mod, dotted):
    for part in dotted.split("."):
        mod = getattr(mod, part)
    return mod


def pgsstb(blk, x, k1, k2, res=None, skip=None):
    """One PGSSTB block (ref :662-723): attention-side residual branch, then the gated-MLP residual branch (+ res: the skip of the
    enclosing BaseBlock when this is its last block; skip: see SkipGrad)."""
    a, sp, pgm = blk.attn, blk.gobal_spectral_attn, blk.local_spectral_attn
    m = blk.mlp
    B, H, W, Cc = x.shape
    fuse = ops.gated_mlp_fuses(B * H * W, Cc, H * W, x.dtype)
    if not torch.is_grad_enabled():
        if fuse:
            pk = blk.packed(x.dtype)
            z, _ = ops.gated_mlp_fwd(x.reshape(-1, Cc), pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"], pk["W2"], pk["b2"], keep=k2, rows_per_batch=H * W,
                                     res=None if res is None else res.reshape(-1, Cc), branch=_pgsstb_attn_infer(blk, k1, x, fuse=True))
            return z.reshape(B, H, W, Cc)
        y = _pgsstb_attn_infer(blk, k1, x)
    else:
        attn_params = (blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias, a.proj.weight, a.proj.bias,
                       a.relative_position_bias_table, sp.temperature, sp.qkv.weight, sp.qkv_dwconv.weight,
                       sp.project_out.weight, *[getattr_path(pgm, k) for k in _PG_KEYS])
        if fuse:
            return _Pgsstb.apply(blk, k1, k2, res, skip, x, *attn_params, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)
        y = _PgsstbAttn.apply(blk, k1, skip, x, *attn_params)
    return _GatedMlp.apply(blk, k2, y, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias, res, skip)


# ---- GDFN / channel attention sub-chains ------------------------------------------------------------
class _GdfnRes(torch.autograd.Function):
    """y = a + project_out(gelu(x1) * x2), [x1|x2] = dwconv(project_in(LN(a)))   (ref FFN :251-265 / FeedForward
    :374-391 inside the pre-norm residual of :286 / :477).  HIP forward; backward = HIP depthwise / gate /
    token-reduction kernels, the two data gradients as gemm_tok on the pre-transposed weights."""

    @staticmethod
    def forward(ctx, ffn, ln, geom, a2, ln_w, ln_b, w_in, w_dw, w_out):
        B, H, W = geom
        pf = ffn.packed(a2.dtype)
        D, HP = a2.shape[1], pf["w_out"].shape[1]
        if B * H * W >= ops.GDFN_FUSED_MIN_PIXELS and ops.gdfn_fused_fits(D, HP, H, W, a2.dtype):
            # one launch; t (what the backward needs) is written, the gate product never reaches HBM
            y, t = ops.gdfn_fused(a2, ln.pair(), pf["w_in"], pf["w9"], pf["w_out"], B, H, W, keep=True)
        else:
            t = ops.gemm_tok(a2, pf["w_in"], ln=ln.pair())
            u = ops.dwconv_gate(t, pf["w9"], B, H, W)
            y = ops.gemm_tok(u, pf["w_out"], epi=1, res=a2)
        ctx.ffn, ctx.ln, ctx.geom = ffn, ln, geom
        ctx.save_for_backward(a2, t)
        return y

    @staticmethod
    def backward(ctx, dy):
        a2, t = ctx.saved_tensors
        ffn, ln, (B, H, W) = ctx.ffn, ctx.ln, ctx.geom
        pf = ffn.packed(a2.dtype)
        HP = pf["w_out"].shape[1]
        hid = ffn.project_out.weight.shape[1]
        D = a2.shape[1]
        dy = dy.contiguous()
        t4 = t.reshape(B, H, W, 2 * HP)
        with ops.reduce_scope(leaf=True):
            du = ops.gemm_tok(dy, pf["w_outT"])                                 # (M,HP)
            if ops.gdfn_dw_bwd_fits(H, W, HP, a2.dtype):
                # gate backward + depthwise backward in ONE launch: the conv is recomputed on every tile's halo, [d x1 | d x2] (2 HP values per
                # token) is neither written nor read back
                u, dt_, dwp = ops.gdfn_dw_bwd(t, pf["w9"], du, B, H, W)
                d_out_w = ops.gemm_tn_blocks(dy, u, [(0, D)], ncols=hid).reshape(D, hid, 1, 1)
                d_dw = torch.empty((2 * hid, 9), dtype=torch.float32, device=t.device)
                ops.reduce_block(dwp, 0, 9, 0, hid, d_dw[:hid], transpose=True)
                ops.reduce_block(dwp, 0, 9, HP, hid, d_dw[hid:], transpose=True)
                d_dw = d_dw.reshape(2 * hid, 1, 3, 3)
            else:
                u, dtdw = ops.dwconv_gate_bwd(t, pf["w9"], du, B, H, W)             # the depthwise conv is recomputed inside
                d_out_w = ops.gemm_tn_blocks(dy, u, [(0, D)], ncols=hid).reshape(D, hid, 1, 1)
                dtdw4 = dtdw.reshape(B, H, W, 2 * HP)
                dt_, d_dw = ops.dwconv3x3_bwd(t4, dtdw4, pf["w9"], col_ranges=[(0, hid), (HP, hid)])
                dt_, d_dw = dt_.reshape(-1, 2 * HP), d_dw.reshape(2 * hid, 1, 3, 3)
            lw, lb = ln.pair()
            if ops.ln_bwd_win_dxn_fits(a2.shape[0], D, a2.dtype):        # project_in's data gradient formed inside the LayerNorm-backward launch
                da, dlw, dlb, xn = ops.ln_bwd_tok_dxn(a2, dt_, pf["w_inT"], dy, lw, lb)
            else:
                da, dlw, dlb, xn = ops.ln_bwd_tok(a2, ops.gemm_tok(dt_, pf["w_inT"]), dy, lw, lb)
            d_in_w = ops.gemm_tn_blocks(dt_, xn, [(0, hid), (HP, hid)]).reshape(2 * hid, D, 1, 1)
        return None, None, None, da, dlw, dlb, d_in_w, d_dw, d_out_w


def _gdfn_res_ag(ffn, ln, a2, B, H, W):
    if not torch.is_grad_enabled():
        # no_grad: the whole block in one launch where the fused kernel covers the shape and the image is big enough to fill
        # the chip with its pixel tiles (t = project_in(LN(a)) and the gate product never reach HBM)
        pf = ffn.packed(a2.dtype)
        D, HP = a2.shape[1], pf["w_out"].shape[1]
        if B * H * W >= ops.GDFN_FUSED_MIN_PIXELS and ops.gdfn_fused_fits(D, HP, H, W, a2.dtype):
            return ops.gdfn_fused(a2, ln.pair(), pf["w_in"], pf["w9"], pf["w_out"], B, H, W)
    return _GdfnRes.apply(ffn, ln, (B, H, W), a2, ln.body.weight, ln.body.bias, ffn.project_in.weight, ffn.dwconv.weight,
                          ffn.project_out.weight)


class _SelfChannelAttnRes(torch.autograd.Function):
    """a = t + M_b v  with q,k,v from dwconv(qkv(LN(t)))   (ref Attention :289-322 inside :476)."""

    @staticmethod
    def forward(ctx, attn, ln, geom, t2, ln_w, ln_b, w_qkv, w_dw, w_out, temp):
        B, H, W = geom
        D = t2.shape[1]
        dt = t2.dtype
        pa = attn.packed(dt)
        w9 = pa["w9"]
        if ops.FUSED_TRAIN and ops.qkv_dwconv_gram_fits(D, attn.num_heads, H, W, dt):
            v, gp, sp, _, q, qk = ops.qkv_dwconv_gram(t2, pa["wqkv"], w9, B, H, W, D, attn.num_heads, ln=ln.pair(), keep=True)
        else:
            q = ops.gemm_tok(t2, pa["wqkv"], ln=ln.pair())
            v, gp, sp, _, qk = ops.dwconv_gram(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], w9[:, :D], w9[:, D:2 * D], w9[:, 2 * D:], 3 * D,
                                               B, H, W, D, attn.num_heads, keep_qk=True)
        Mb, MbT, gp, sp = ops.spectral_fold(gp, sp, pa["temp"], pa["wo"], dt, transposed=True)
        a = ops.gemm_tok(v, Mb, epi=1, res=t2)
        ctx.attn, ctx.ln, ctx.geom = attn, ln, geom
        ctx.has_qk = qk is not None
        ctx.save_for_backward(t2, q, v, gp, sp, Mb, MbT, *([qk] if qk is not None else []))
        return a

    @staticmethod
    def backward(ctx, da):
        t2, q, v, gp, sp, Mb, MbT = ctx.saved_tensors[:7]
        qk = ctx.saved_tensors[7] if ctx.has_qk else None
        attn, ln, (B, H, W) = ctx.attn, ctx.ln, ctx.geom
        D = t2.shape[1]
        M = t2.shape[0]
        pa = attn.packed(t2.dtype)
        w9 = pa["w9"]
        da = da.contiguous()
        q4 = q.reshape(B, H, W, 3 * D)
        with ops.reduce_scope(leaf=True):      # every split partial of this backward is summed by ONE launch when the scope exits
            dtq, dtk, dtv, dwq, dwk, dwv, dtemp, dwo = channel_attention_bwd(
                da, q4[..., :D], q4[..., D:2 * D], q4[..., 2 * D:], w9[:, :D], w9[:, D:2 * D], w9[:, 2 * D:], v, gp, sp, Mb, MbT,
                attn.temperature, attn.project_out.weight, attn.num_heads, B, H, W, qk=qk)
            if dtq.data_ptr() + D * dtq.element_size() == dtk.data_ptr() and dtq.stride(2) == 3 * D:
                dt3 = torch.as_strided(dtq, (M, 3 * D), (3 * D, 1))
            else:
                dt3 = torch.cat([dtq, dtk, dtv], dim=-1).reshape(M, 3 * D)
            lw, lb = ln.pair()
            if ops.ln_bwd_win_dxn_fits(M, D, t2.dtype) and dt3.is_contiguous():      # the qkv conv's data gradient inside the LayerNorm-backward launch
                dt_in, dlw, dlb, xn = ops.ln_bwd_tok_dxn(t2, dt3, pa["wqkvT"], da, lw, lb)
            else:
                dt_in, dlw, dlb, xn = ops.ln_bwd_tok(t2, ops.gemm_tok(dt3, pa["wqkvT"]), da, lw, lb)
            d_qkv = ops.gemm_tn(dt3, xn).reshape(3 * D, D, 1, 1)
        d_dw = _join_taps(dwq, dwk, dwv).reshape(3 * D, 1, 3, 3)
        return None, None, None, dt_in, dlw, dlb, d_qkv, d_dw, dwo.reshape(D, D, 1, 1), dtemp.reshape(-1, 1, 1)


class _CrossChannelAttnRes(torch.autograd.Function):
    """a = x_q + M_b v, q from x_q, k/v from x_kv   (ref CrossAttention :220-249 inside :282).

    x_q is TVSP's text map and arrives in fp32: it is rank one (clip[i,j] * L[b,:]), which makes the gradients through
    norm11 residuals of large cancelling sums -- rounding it to bf16 before the LayerNorm changes d(norm11.weight) and
    d(q.weight) by 15 % (csrc/layernorm.hip).  So norm11 runs on the fp32 input (layernorm_tok, fp32 statistics, output
    in the compute dtype) and its backward in fp32, as the reference's autocast does; everything after it is the
    compute-dtype path.

    x_kv is TVSP's visual prompt, ONE (ps, ps, D) map shared by the whole batch (the reference expands it to B copies, :578): vis1 is
    that one map (ps*ps, D).  norm12 and the kv 1x1 conv run on it once (their B results are identical) and only t_kv is repeated for
    the depthwise / Gram kernel; the backward sums d t_kv over the batch first and runs the kv / norm12 backward on ps*ps tokens."""

    @staticmethod
    def forward(ctx, ct, geom, dt, text32, vis1, vis1_f32, visual_prompt, n11w, n11b, n12w, n12b, w_q, w_kv, w_qdw, w_kvdw, w_out, temp):
        """vis1: TVSP.packed()'s token view of `visual_prompt` (the parameter itself is the autograd input)"""
        B, H, W = geom
        D = text32.shape[1]
        pa = ct.attn.packed(dt)
        lw, lb = ct.norm11.pair()
        xq, text_c = ops.layernorm_tok(text32, lw, lb, dt, want_cast=True)      # + the text map in the compute dtype: the residual operand
        tq = ops.gemm_tok(xq, pa["wq"])
        tkv = ops.gemm_tok(vis1, pa["wkv"], ln=ct.norm12.pair()).repeat(B, 1)      # (B ps ps, 2D): the same rows for every sample
        w9 = pa["w9"]
        v, gp, sp, _ = ops.dwconv_gram(tq, tkv[:, :D], tkv[:, D:], w9[:, :D], w9[:, D:2 * D], w9[:, 2 * D:], 3 * D,
                                       B, H, W, D, ct.attn.num_heads)
        Mb, MbT, gp, sp = ops.spectral_fold(gp, sp, pa["temp"], pa["wo"], dt, transposed=True)
        a = ops.gemm_tok(v, Mb, epi=1, res=text_c)
        ctx.ct, ctx.geom = ct, geom
        ctx.save_for_backward(text32, vis1, xq, tq, tkv, v, gp, sp, Mb, MbT, vis1_f32)
        return a

    @staticmethod
    def backward(ctx, da):
        text32, vis1, xq, tq, tkv, v, gp, sp, Mb, MbT, vis1_f32 = ctx.saved_tensors
        ct, (B, H, W) = ctx.ct, ctx.geom
        attn = ct.attn
        D = text32.shape[1]
        M = text32.shape[0]
        pa = attn.packed(vis1.dtype)
        w9 = pa["w9"]
        da = da.contiguous()
        tq4, tkv4 = tq.reshape(B, H, W, D), tkv.reshape(B, H, W, 2 * D)
        with ops.reduce_scope(leaf=True):      # every split partial of this backward is summed by ONE launch when the scope exits
            dtq, dtk, dtv, dwq, dwk, dwv, dtemp, dwo = channel_attention_bwd(
                da, tq4, tkv4[..., :D], tkv4[..., D:], w9[:, :D], w9[:, D:2 * D], w9[:, 2 * D:], v, gp, sp, Mb, MbT,
                attn.temperature, attn.project_out.weight, attn.num_heads, B, H, W)
            dtq2 = dtq.reshape(M, D).contiguous()
            if dtk.data_ptr() + D * dtk.element_size() == dtv.data_ptr() and dtk.stride(2) == 2 * D:      # halves of one buffer
                dkv = torch.as_strided(dtk, (M, 2 * D), (2 * D, 1))
            else:
                dkv = torch.cat([dtk, dtv], dim=-1).reshape(M, 2 * D)
            n11w, n11b = ct.norm11.pair()
            n12w, n12b = ct.norm12.pair()
            # norm11 backward in fp32 on the fp32 text map (see the class docstring); dres = the residual path of `a`.  Where the kernel covers
            # the width, the q conv's data gradient is formed inside the launch (fp32 rows of x, 16-bit dtq / weights / da)
            if ops.ln_bwd_tok_dxn_f32_fits(M, D, dtq2.dtype):
                dtext, d11w, d11b, _ = ops.ln_bwd_tok_dxn(text32, dtq2, pa["wqT"], da, n11w, n11b, want_xn=False)
            else:
                dtext, d11w, d11b, _ = ops.ln_bwd_tok(text32, ops.gemm_tok(dtq2, pa["wqT"]).float(), da.float(), n11w, n11b)
            # the kv side is batch-invariant up to t_kv: its gradient is summed over the batch (fp32 accumulation) and everything behind it
            # -- the 1x1 conv's data gradient, norm12's backward, both parameter gradients -- runs on the ps*ps tokens of the one map
            dkv1 = dkv.reshape(B, H * W, 2 * D).sum(dim=0)
            if ops.ln_bwd_tok_dxn_f32_fits(vis1.shape[0], D, vis1.dtype) and vis1.shape[0] % 64 == 0:
                # fp32 rows of the parameter: its gradient comes out in fp32, the type it is accumulated in
                dvis, d12w, d12b, xv = ops.ln_bwd_tok_dxn(vis1_f32, dkv1, pa["wkvT"], None, n12w, n12b)
            elif ops.ln_bwd_win_dxn_fits(vis1.shape[0], D, vis1.dtype) and vis1.shape[0] % 64 == 0:
                dvis, d12w, d12b, xv = ops.ln_bwd_tok_dxn(vis1, dkv1, pa["wkvT"], None, n12w, n12b)
            else:
                dvis, d12w, d12b, xv = ops.ln_bwd_tok(vis1, ops.gemm_tok(dkv1, pa["wkvT"]), torch.zeros_like(vis1), n12w, n12b)
            d_wq = ops.gemm_tn(dtq2, xq).reshape(D, D, 1, 1)
            d_wkv = ops.gemm_tn(dkv1, xv).reshape(2 * D, D, 1, 1)
        d_prompt = dvis.t().reshape(1, D, H, W)               # (ps*ps, D) tokens -> the parameter's (1, D, ps, ps), a view
        return (None, None, None, dtext, None, None, d_prompt, d11w, d11b, d12w, d12b, d_wq, d_wkv, dwq.reshape(D, 1, 3, 3),
                _join_taps(dwk, dwv).reshape(2 * D, 1, 3, 3), dwo.reshape(D, D, 1, 1), dtemp.reshape(-1, 1, 1))


class _TextMap(torch.autograd.Function):
    """text[b,i,j,:] = L[b,:] * clip[floor(i*B/ps), floor(j*512/ps)]  (ref :575-577)"""

    @staticmethod
    def forward(ctx, L, clip, ps):
        clip = clip.float().contiguous()
        ctx.save_for_backward(clip)
        return ops.tvsp_text_map(L.float().contiguous(), clip, ps)

    @staticmethod
    def backward(ctx, dtext):
        (clip,) = ctx.saved_tensors
        return ops.tvsp_text_map_bwd(dtext.float().contiguous(), clip), None, None


class _Bilinear(torch.autograd.Function):
    """F.interpolate(mode="bilinear", align_corners=False) on channels-last data (ref :580)"""

    @staticmethod
    def forward(ctx, x, H, W):
        ctx.hw = x.shape[1:3]
        return ops.resize_bilinear(x.contiguous(), H, W)

    @staticmethod
    def backward(ctx, dy):
        return ops.resize_bilinear(dy.contiguous(), ctx.hw[0], ctx.hw[1], backward=True), None, None


def tvsp(mod, x, clip_prompt, prompt_weights, out=None):
    """TVSP.forward (ref :572-583) with the batch-coupling broadcast of SURVEY Q1 made explicit:
    text[b,i,j,d] = L[b,d] * clip[floor(i*B/ps), floor(j*512/ps)].  out: see conv3x3."""
    B, H, W, D = x.shape
    ps, dt, dev = mod.prompt_size, x.dtype, x.device
    L = mix_rows(prompt_weights, mod.text_prompt_learnable)                            # (B,D) = (w[..., None] * learnable (T,D)).mean(1)
    text = _TextMap.apply(L, clip_prompt, ps)                                          # fp32: see _CrossChannelAttnRes
    pm = mod.packed(dt)                                     # vis1: the one visual prompt map as tokens (ref :578 expands it to B copies)
    ct = mod.cross_transformer
    at = ct.attn
    a = _CrossChannelAttnRes.apply(ct, (B, ps, ps), dt, text.reshape(-1, D), pm["vis1"], pm["vis1_f32"], mod.visual_prompt, ct.norm11.body.weight,
                                   ct.norm11.body.bias, ct.norm12.body.weight, ct.norm12.body.bias, at.q.weight, at.kv.weight,
                                   at.q_dwconv.weight, at.kv_dwconv.weight, at.project_out.weight, at.temperature)
    y = _gdfn_res_ag(ct.ffn, ct.norm2, a, B, ps, ps).reshape(B, ps, ps, D)
    if (H, W) != (ps, ps):                                                             # ref :580
        y = _Bilinear.apply(y, H, W)
    return conv3x3(y, mod.conv_last, out=out)


class _JoinLeft(torch.autograd.Function):
    """cat([x, p], -1) where p already IS the right half of `buf` (TVSP's last conv wrote it there): x is copied into the left half and the
    buffer is the result -- half the bytes of the cat launch (ref :596)."""

    @staticmethod
    def forward(ctx, x, p, buf):
        C = x.shape[-1]
        assert buf.is_contiguous() and buf.shape[:3] == x.shape[:3] and p.shape == (*x.shape[:3], buf.shape[-1] - C) and p.dtype == buf.dtype == x.dtype
        assert p.data_ptr() == buf.data_ptr() + C * buf.element_size() and p.stride() == buf.stride(), "p is not the right half of buf"
        buf[..., :C].copy_(x)
        ctx.c = C
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, d):
        return d[..., :ctx.c], d[..., ctx.c:], None


def prompt_fusion(mod, x, prompt, out=None, joined=None):
    """out: an uninitialised (B,H,W,out_dim) channel slice of a wider buffer the result is written into (see shuffle_join);
    joined: the (B,H,W,2C) buffer whose right half `prompt` is (TVSP wrote it there, see _JoinLeft), else the two are concatenated"""
    t = torch.cat([x, prompt], dim=-1) if joined is None else _JoinLeft.apply(x, prompt, joined)
    B, H, W, D = t.shape
    tb = mod.transformer
    at = tb.attn
    if not torch.is_grad_enabled():
        a = _self_channel_attn_infer(at, tb.norm1, (B, H, W), t.reshape(-1, D))
    else:
        a = _SelfChannelAttnRes.apply(at, tb.norm1, (B, H, W), t.reshape(-1, D), tb.norm1.body.weight, tb.norm1.body.bias,
                                      at.qkv.weight, at.qkv_dwconv.weight, at.project_out.weight, at.temperature)
    y = _gdfn_res_ag(tb.ffn, tb.norm2, a, B, H, W).reshape(B, H, W, D)
    return conv1x1(y, mod.conv, out=out)


# ---- plain convs / resamplers ----------------------------------------------------------------------
def _conv_cache(conv):
    c = conv.__dict__.get("_mphsir_cache")
    if c is None:
        c = conv.__dict__["_mphsir_cache"] = ops.WeightCache()
    return c


def _packed_conv1x1(conv, w, dtype):
    N, K = w.shape[0], w.shape[1]
    return _conv_cache(conv).get([w], dtype, lambda: dict(w=w.reshape(N, K).to(ops.cdt(dtype)).contiguous(),
                                                           wT=w.reshape(N, K).t().to(ops.cdt(dtype)).contiguous()))


def _packed_conv3x3(conv, w, dtype):
    return _conv_cache(conv).get([w], dtype, lambda: dict(fwd=ops.pack_conv3x3(w, dtype),
                                                           bwd=ops.pack_conv3x3(w, dtype, flip_transpose=True)))


def _row_major(t):
    """a 2-D view the token kernels can read as it is (unit column stride, 16-byte aligned rows), else one contiguous copy"""
    es = t.element_size()
    if t.dim() == 2 and t.stride(1) == 1 and (t.stride(0) * es) % 16 == 0 and t.data_ptr() % 16 == 0 and t.stride(0) >= t.shape[1]:
        return t
    return t.contiguous()


class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, conv, out=None):
        N, K = w.shape[0], w.shape[1]
        pk = _packed_conv1x1(conv, w, x.dtype)
        ctx.save_for_backward(x, w)
        ctx.conv = conv
        if out is not None:       # a channel slice of a wider row-major buffer: written through its row pitch
            assert out.shape == (*x.shape[:-1], N) and out.dtype == x.dtype and out.stride(-1) == 1
            out = out.view(-1, N)
        y = ops.gemm_tok(x.reshape(-1, K), pk["w"], out=out)
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, K = w.shape[0], w.shape[1]
        dy2, x2 = dy.reshape(-1, N), x.reshape(-1, K)
        pk = _packed_conv1x1(ctx.conv, w, x.dtype)
        dy2 = _row_major(dy2)            # a channel slice of a wider gradient (the split of a `cat`) is read in place through its row pitch
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm_tok(dy2, pk["wT"]).reshape(x.shape)
        dw = None
        if ctx.needs_input_grad[1]:
            with ops.reduce_scope(leaf=True):          # w is a leaf: the sum joins the deferred parameter-gradient sums
                dw = ops.gemm_tn(dy2, x2).reshape(w.shape)
        return dx, dw, None, None


def conv1x1(x, conv, out=None):
    """bias-free 1x1 conv / Linear on channels-last data; `conv` is the nn.Conv2d holder (its packed weights are cached on it)."""
    return _Conv1x1.apply(x, conv.weight, conv, out)


class _Conv3x3(torch.autograd.Function):
    """dense 3x3 conv as an implicit GEMM on the HIP kernel (forward, input gradient) + im2col/gemm_tn (weights)."""

    @staticmethod
    def forward(ctx, x, w, conv, keep_pad=False, out=None):
        """x with Cin channels, or already zero-padded to round_up(Cin, 32) (input_head).  keep_pad: the output keeps its round_up(Cout, 16)
        channels (the padded ones are zero: zero weight rows) for a consumer that reads the first Cout through the row pitch (output_head).
        out: an uninitialised (B,H,W,Cout) channel slice of a wider buffer the result is written into (Cout % 16 == 0)."""
        Cout, Cin = w.shape[0], w.shape[1]
        Cp = ops.round_up(Cin, 32)
        assert x.shape[-1] in (Cin, Cp)
        xp = x.contiguous() if x.shape[-1] == Cp else F.pad(x, (0, Cp - Cin)).contiguous()
        pk = _packed_conv3x3(conv, w, x.dtype)
        assert out is None or Cout % 16 == 0
        y = ops.conv3x3_tok(xp, pk["fwd"], out=out)
        ctx.save_for_backward(xp, w)
        ctx.conv, ctx.xc = conv, x.shape[-1]
        if out is not None:
            return y.view(y.shape)
        return y if keep_pad or y.shape[-1] == Cout else y[..., :Cout].contiguous()

    @staticmethod
    def backward(ctx, dy):
        xp, w = ctx.saved_tensors
        Cout, Cin = w.shape[0], w.shape[1]
        Cp, Co32 = xp.shape[-1], ops.round_up(Cout, 32)
        if dy.shape[-1] == Co32:      # (a channel slice of a wider gradient -- the split of a concatenation -- is read in place through its row pitch)
            dyp = dy if ops.pixel_pitch(dy) is not None else dy.contiguous()
        else:
            dyp = F.pad(dy[..., :Cout], (0, Co32 - Cout)).contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            pk = _packed_conv3x3(ctx.conv, w, dy.dtype)
            dx = ops.conv3x3_tok(dyp, pk["bwd"])
            dx = dx if dx.shape[-1] == ctx.xc else dx[..., :ctx.xc].contiguous()
        if ctx.needs_input_grad[1]:
            if dyp.dtype in (torch.bfloat16, torch.float16):      # the gather happens inside the token-reduction GEMM; the
                with ops.reduce_scope(leaf=True):                 # ordered sum writes the (Cout, Cin, 3, 3) layout (w is a leaf)
                    dw = ops.conv3x3_wgrad(dyp.reshape(-1, Co32), xp, cout=Cout, cin=Cin)
            else:
                g = ops.gemm_tn(dyp.reshape(-1, Co32), ops.im2col3x3(xp))[:Cout]
                dw = g.reshape(Cout, 9, Cp)[:, :, :Cin].permute(0, 2, 1).reshape(Cout, Cin, 3, 3)
        return dx, dw, None, None, None


def conv3x3(x, conv, keep_pad=False, out=None):
    """dense 3x3, stride 1, zero padding, no bias on channels-last data; `conv` is the nn.Conv2d holder."""
    return _Conv3x3.apply(x, conv.weight, conv, keep_pad, out)


class _InputHead(torch.autograd.Function):
    """inp_img (B,C,H,W) fp32 -> channels-last, compute dtype, channels zero-padded to a multiple of 32 (ref :824 + the patch embedding's
    input layout): one launch for .to / permute / contiguous / pad"""

    @staticmethod
    def forward(ctx, inp, dt):
        ctx.c = inp.shape[1]
        return ops.nchw_to_cl(inp.contiguous(), dt, ops.round_up(inp.shape[1], 32))

    @staticmethod
    def backward(ctx, d):
        return (ops.cl_to_nchw_add(d.contiguous(), ctx.c) if ctx.needs_input_grad[0] else None), None


def input_head(inp, dt):
    if inp.dtype != torch.float32 or ops.round_up(inp.shape[1], 32) > 252:
        return inp.to(dt).permute(0, 2, 3, 1).contiguous()
    return _InputHead.apply(inp, dt)


class _OutputHead(torch.autograd.Function):
    """restored = output_conv(...) + inp_img (ref :842-843): y (B,H,W,Cy) channels-last with the conv's padded channels kept, inp (B,C,H,W)
    fp32 -> (B,C,H,W) fp32.  One launch each way (slice / permute / .to / add, and pad / permute / .to in the backward)."""

    @staticmethod
    def forward(ctx, y, inp):
        ctx.meta = (y.dtype, y.shape[-1])      # (the conv pads its output to a multiple of 16 and reads a gradient padded to 32: equal for 31 channels)
        return ops.cl_to_nchw_add(y, inp.shape[1], inp.contiguous())

    @staticmethod
    def backward(ctx, d):
        dt, cy = ctx.meta
        return ops.nchw_to_cl(d.contiguous(), dt, cy), (d if ctx.needs_input_grad[1] else None)


def output_head(r, conv, inp):
    """conv3x3(r, conv).permute(0,3,1,2).to(inp.dtype) + inp"""
    if inp.dtype != torch.float32 or ops.round_up(conv.weight.shape[0], 32) > 252:
        return conv3x3(r, conv).permute(0, 3, 1, 2).to(inp.dtype) + inp
    return _OutputHead.apply(conv3x3(r, conv, keep_pad=True), inp)


class _MixRows(torch.autograd.Function):
    """(w.unsqueeze(-1) * table).mean(1) for task weights w (B,T) and a table parameter holding (T,D) values in any singleton-padded shape
    (ref :527, TVSP :574): one launch, and ONE for the table's gradient w^T dO / T, handed back in the parameter's own shape (the
    select / slice backward chain of indexing the parameter first was 2 launches per index)."""

    @staticmethod
    def forward(ctx, w, table):
        T = w.shape[1]
        ctx.save_for_backward(w)
        ctx.tshape = table.shape
        return ops.mix_rows(w, table.reshape(T, -1).contiguous(), 1.0 / T)

    @staticmethod
    def backward(ctx, d):
        (w,) = ctx.saved_tensors
        assert not ctx.needs_input_grad[0]
        return None, ops.mix_rows(w, d.contiguous(), 1.0 / w.shape[1], transA=True).reshape(ctx.tshape)


def mix_rows(w, table):
    return _MixRows.apply(w.to(torch.float32).contiguous(), table)


def pixel_unshuffle2(x):
    B, H, W, Cc = x.shape
    return x.reshape(B, H // 2, 2, W // 2, 2, Cc).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, Cc * 4)


class _ShuffleJoin(torch.autograd.Function):
    """cat([pixel_shuffle2(c), f], -1) where f already IS the right half of `buf` (its producer wrote it there through the row pitch):
    the shuffle writes the left half and the buffer is the result -- no cat launch, the decoder's concatenations (ref :838, :843)."""

    @staticmethod
    def forward(ctx, c, f, buf):
        B, H, W, C4 = c.shape
        Cl = C4 // 4
        Ct = buf.shape[-1]
        assert buf.is_contiguous() and buf.shape[:3] == (B, 2 * H, 2 * W) and f.shape == (B, 2 * H, 2 * W, Ct - Cl) and f.dtype == buf.dtype == c.dtype
        assert f.data_ptr() == buf.data_ptr() + Cl * buf.element_size() and f.stride() == buf.stride(), "f is not the right half of buf"
        dst = buf[..., :Cl].unflatten(1, (H, 2)).unflatten(3, (W, 2))                     # (B,H,2,W,2,Cl) view of the left half
        dst.copy_(c.reshape(B, H, W, Cl, 2, 2).permute(0, 1, 4, 2, 5, 3))               # one strided copy launch
        ctx.cl = Cl
        return buf.view(buf.shape)

    @staticmethod
    def backward(ctx, d):
        Cl = ctx.cl
        return pixel_unshuffle2(d[..., :Cl]), d[..., Cl:], None


def shuffle_join(c, f, buf):
    return _ShuffleJoin.apply(c, f, buf)


def pixel_shuffle2(x):
    B, H, W, C4 = x.shape
    return x.reshape(B, H, W, C4 // 4, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B, H * 2, W * 2, C4 // 4)
