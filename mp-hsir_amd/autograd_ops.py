"""autograd wiring of the fused ops: forward = HIP kernels through the C ABI; backward = gradient
of the same math recomputed with torch ops on the GPU (see _composite.py; interim until the HIP
backward kernels land).  Only the op inputs are saved for backward."""
import torch
import torch.nn.functional as F

from . import _composite as C
from . import ops


class _Recompute(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fwd_fn, comp_fn, names, n_in, *tensors):
        ctx.comp_fn, ctx.names, ctx.n_in = comp_fn, names, n_in
        ctx.save_for_backward(*tensors)
        with torch.no_grad():
            return fwd_fn(*tensors[:n_in])

    @staticmethod
    def backward(ctx, dy):
        tensors = ctx.saved_tensors
        n_in = ctx.n_in
        leaves = [t.detach().requires_grad_(True) for t in tensors]
        with torch.enable_grad():
            y = ctx.comp_fn(*leaves[:n_in], dict(zip(ctx.names, leaves[n_in:])))
        need = [i for i, need_i in enumerate(ctx.needs_input_grad[4:]) if need_i]
        grads = torch.autograd.grad(y, [leaves[i] for i in need], dy.to(y.dtype), allow_unused=True)
        out = [None] * len(tensors)
        for i, g in zip(need, grads):
            out[i] = g
        return (None, None, None, None) + tuple(out)


def _apply(fwd, comp, module, *inputs, only=None):
    named = [(n, p) for n, p in module.named_parameters() if only is None or only(n)]
    names = tuple(n for n, _ in named)
    return _Recompute.apply(fwd, comp, names, len(inputs), *inputs, *[p for _, p in named])


# ---- PGSSTB ---------------------------------------------------------------------------------------
def _pgsstb_attn_forward(blk, x, k1):
    B, H, W, Cc = x.shape
    dt = x.dtype
    pk = blk.packed(dt)
    sp = blk.gobal_spectral_attn.packed(dt)
    heads, shift = blk.num_heads, blk.shift_size
    x2 = x.reshape(-1, Cc)
    sa, gate = ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"],
                                pk["bproj"], pk["pg"], heads, shift)
    sa2 = sa.reshape(-1, Cc)
    t = ops.gemm_tok(sa2, sp["wqkv"])
    w9 = sp["w9"]
    v, gp, spart, _ = ops.dwconv_gram(t[:, :Cc], t[:, Cc:2 * Cc], t[:, 2 * Cc:], w9[:, :Cc], w9[:, Cc:2 * Cc], w9[:, 2 * Cc:],
                                      3 * Cc, B, H, W, Cc, heads)
    Mb = ops.spectral_fold(gp, spart, sp["temp"], sp["wo"], dt)
    y = ops.gemm_tok(v, Mb, epi=2, res=x2, sa=sa2, gate=gate, keep=k1, geom=(H, W, shift))
    return y.reshape(B, H, W, Cc)


class _GatedMlp(torch.autograd.Function):
    """z = y + keep*mlp(LN2(y)): HIP forward and HIP data-gradient; the parameter gradients are the
    token-reduction GEMMs / column sums of the three matrices the backward kernel writes."""

    @staticmethod
    def forward(ctx, blk, k2, y, ln_w, ln_b, fc1_w, fc1_b, fc2_w, fc2_b):
        B, H, W, Cc = y.shape
        pk = blk.packed(y.dtype)
        ctx.blk, ctx.k2 = blk, k2
        ctx.save_for_backward(y)
        z = ops.gated_mlp_fwd(y.reshape(-1, Cc), pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"], pk["W2"], pk["b2"],
                              keep=k2, rows_per_batch=H * W)
        return z.reshape(B, H, W, Cc)

    @staticmethod
    def backward(ctx, dz):
        (y,) = ctx.saved_tensors
        blk, k2 = ctx.blk, ctx.k2
        B, H, W, Cc = y.shape
        dt = y.dtype
        pk = blk.packed(dt)
        dz2 = dz.reshape(-1, Cc).contiguous()
        dm = dz2 if k2 is None else (dz.float() * k2.reshape(B, 1, 1, 1)).to(dt).reshape(-1, Cc)
        dx, xn, h, dpre, part = ops.gated_mlp_bwd(y.reshape(-1, Cc), dz2, dm, pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"],
                                                  pk["W1T"], pk["W2T"])
        hid = blk.mlp.fc2.weight.shape[1]
        HP = h.shape[1]
        dW2 = (dm.t() @ h)[:, :hid].float()
        dW1p = (dpre.t() @ xn).float()
        db1p = dpre.float().sum(0)
        dln = part.sum(0)
        return (None, None, dx.reshape(B, H, W, Cc), dln[0], dln[1],
                torch.cat([dW1p[:hid], dW1p[HP:HP + hid]], 0), torch.cat([db1p[:hid], db1p[HP:HP + hid]]),
                dW2, dm.float().sum(0))


def pgsstb(blk, x, k1, k2):
    heads, shifted = blk.num_heads, blk.shift_size > 0
    y = _apply(lambda x_: _pgsstb_attn_forward(blk, x_, k1), lambda x_, P: C.pgsstb_attn(P, x_, heads, shifted, k1), blk, x,
               only=lambda n: not (n.startswith("mlp.") or n.startswith("norm2.")))
    if x.dtype == torch.float32 and x.shape[-1] > 256:      # fp32 LDS budget of the backward kernel
        pk = blk.packed(x.dtype)

        def fwd(y_):
            B, H, W, Cc = y_.shape
            return ops.gated_mlp_fwd(y_.reshape(-1, Cc), pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"], pk["W2"], pk["b2"],
                                     keep=k2, rows_per_batch=H * W).reshape(B, H, W, Cc)
        return _apply(fwd, lambda y_, P: C.mlp_branch(P, y_, k2), blk, y,
                      only=lambda n: n.startswith("mlp.") or n.startswith("norm2."))
    m = blk.mlp
    return _GatedMlp.apply(blk, k2, y, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias)


# ---- GDFN / channel attention sub-chains ------------------------------------------------------------
def _gdfn_res(ffn, ln, a2, B, H, W):
    pf = ffn.packed(a2.dtype)
    t = ops.gemm_tok(a2, pf["w_in"], ln=ln.pair())
    u = ops.dwconv_gate(t, pf["w9"], B, H, W)
    return ops.gemm_tok(u, pf["w_out"], epi=1, res=a2)


def _cross_transformer_forward(ct, text, vis):
    B, ps, _, D = text.shape
    dt = text.dtype
    pa = ct.attn.packed(dt)
    t2, v2 = text.reshape(-1, D), vis.reshape(-1, D)
    tq = ops.gemm_tok(t2, pa["wq"], ln=ct.norm11.pair())
    tkv = ops.gemm_tok(v2, pa["wkv"], ln=ct.norm12.pair())
    w9 = pa["w9"]
    v, gp, sp, _ = ops.dwconv_gram(tq, tkv[:, :D], tkv[:, D:], w9[:, :D], w9[:, D:2 * D], w9[:, 2 * D:], 3 * D,
                                   B, ps, ps, D, ct.attn.num_heads)
    Mb = ops.spectral_fold(gp, sp, pa["temp"], pa["wo"], dt)
    a = ops.gemm_tok(v, Mb, epi=1, res=t2)
    return _gdfn_res(ct.ffn, ct.norm2, a, B, ps, ps).reshape(B, ps, ps, D)


def _transformer_block_forward(tb, t):
    B, H, W, D = t.shape
    dt = t.dtype
    pa = tb.attn.packed(dt)
    t2 = t.reshape(-1, D)
    q = ops.gemm_tok(t2, pa["wqkv"], ln=tb.norm1.pair())
    w9 = pa["w9"]
    v, gp, sp, _ = ops.dwconv_gram(q[:, :D], q[:, D:2 * D], q[:, 2 * D:], w9[:, :D], w9[:, D:2 * D], w9[:, 2 * D:], 3 * D,
                                   B, H, W, D, tb.attn.num_heads)
    Mb = ops.spectral_fold(gp, sp, pa["temp"], pa["wo"], dt)
    a = ops.gemm_tok(v, Mb, epi=1, res=t2)
    return _gdfn_res(tb.ffn, tb.norm2, a, B, H, W).reshape(B, H, W, D)


def tvsp(mod, x, clip_prompt, prompt_weights):
    """TVSP.forward (ref :572-583) with the batch-coupling broadcast of SURVEY Q1 made explicit:
    text[b,i,j,d] = L[b,d] * clip[floor(i*B/ps), floor(j*512/ps)]."""
    B, H, W, D = x.shape
    ps, dt, dev = mod.prompt_size, x.dtype, x.device
    learn = mod.text_prompt_learnable[0, :, :, 0, 0]                                   # (T,D)
    L = (prompt_weights.to(torch.float32).unsqueeze(-1) * learn.unsqueeze(0)).mean(dim=1)      # (B,D)
    ar = torch.arange(ps, device=dev)
    clip_map = clip_prompt[(ar * B) // ps][:, (ar * 512) // ps]                       # (ps,ps)
    text = (clip_map[None, :, :, None] * L[:, None, None, :]).to(dt).contiguous()
    vis = mod.visual_prompt.permute(0, 2, 3, 1).expand(B, ps, ps, D).to(dt).contiguous()
    ct = mod.cross_transformer
    y = _apply(lambda t_, v_: _cross_transformer_forward(ct, t_, v_),
               lambda t_, v_, P: C.cross_transformer(P, t_, v_, ct.attn.num_heads), ct, text, vis)
    if (H, W) != (ps, ps):                                                             # ref :580
        y = F.interpolate(y.permute(0, 3, 1, 2), (H, W), mode="bilinear").permute(0, 2, 3, 1).contiguous()
    return conv3x3(y, mod.conv_last.weight)


def prompt_fusion(mod, x, prompt):
    t = torch.cat([x, prompt], dim=-1)
    tb = mod.transformer
    y = _apply(lambda t_: _transformer_block_forward(tb, t_),
               lambda t_, P: C.transformer_block(P, t_, tb.attn.num_heads), tb, t)
    return conv1x1(y, mod.conv.weight)


# ---- plain convs / resamplers ----------------------------------------------------------------------
class _Conv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        N, K = w.shape[0], w.shape[1]
        y = ops.gemm_tok(x.reshape(-1, K), w.reshape(N, K).to(x.dtype).contiguous())
        return y.reshape(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        N, K = w.shape[0], w.shape[1]
        dy2, x2 = dy.reshape(-1, N), x.reshape(-1, K)
        dx = ops.gemm_tok(dy2.contiguous(), w.reshape(N, K).t().to(x.dtype).contiguous()).reshape(x.shape) \
            if ctx.needs_input_grad[0] and N % 32 == 0 and K % 16 == 0 else (dy2 @ w.reshape(N, K).to(dy.dtype)).reshape(x.shape)
        dw = (dy2.float().t() @ x2.float()).reshape(w.shape) if ctx.needs_input_grad[1] else None
        return dx, dw


def conv1x1(x, w):
    return _Conv1x1.apply(x, w)


def conv3x3(x, w):
    """dense 3x3, stride 1, zero padding, no bias on channels-last data (MIOpen through PyTorch-ROCm: glue)."""
    return F.conv2d(x.permute(0, 3, 1, 2), w.to(x.dtype), None, 1, 1).permute(0, 2, 3, 1).contiguous()


def pixel_unshuffle2(x):
    B, H, W, Cc = x.shape
    return x.reshape(B, H // 2, 2, W // 2, 2, Cc).permute(0, 1, 3, 5, 2, 4).reshape(B, H // 2, W // 2, Cc * 4)


def pixel_shuffle2(x):
    B, H, W, C4 = x.shape
    return x.reshape(B, H, W, C4 // 4, 2, 2).permute(0, 1, 4, 2, 5, 3).reshape(B, H * 2, W * 2, C4 // 4)
