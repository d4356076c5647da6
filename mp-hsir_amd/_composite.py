"""PyTorch-ROCm (GPU) composites of the fused ops, used ONLY to derive gradients.

Round-1 status: every forward runs on the hand-written HIP kernels; the backward of a fused op is
obtained by re-evaluating the same math with stock torch ops on the GPU under autograd and
differentiating that (recompute-in-backward; nothing but the op inputs is saved).  These composites
are scheduled to be replaced, op by op, by HIP backward kernels (DESIGN.md "backward").  They run
on the GPU in the compute dtype -- this is not a CPU fallback and nothing here imports oracle/.

Shapes: activations channels-last (B,H,W,C); P maps parameter names *relative to the owning
module* to tensors (the autograd leaves of the recomputation).
"""
import math

import torch
import torch.nn.functional as F

from . import ops

WINDOW, SHIFT, PROMPT_LEN = 8, 4, 128


def layer_norm(x, w, b):
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), 1e-5).to(x.dtype)


def pw(x, w4):
    return x @ w4.reshape(w4.shape[0], -1).to(x.dtype).t()


class _DwConv3x3(torch.autograd.Function):
    """depthwise 3x3 through the HIP kernels: forward, backward-data (flipped taps) and weight gradient.
    (MIOpen's bf16 channels-last depthwise backward falls back to naive kernels: 75 % of a training step.)"""

    @staticmethod
    def forward(ctx, x, w4):
        w9 = ops.pack_dw(w4)
        ctx.save_for_backward(x, w9)
        ctx.wshape = w4.shape
        return ops.dwconv3x3(x, w9)

    @staticmethod
    def backward(ctx, dy):
        x, w9 = ctx.saved_tensors
        dy = dy.contiguous()
        dx = ops.dwconv3x3(dy, w9, flip=True) if ctx.needs_input_grad[0] else None
        dw = ops.dwconv3x3_wgrad(x, dy).t().reshape(ctx.wshape) if ctx.needs_input_grad[1] else None
        return dx, dw


def dw3(x, w4):
    return _DwConv3x3.apply(x.contiguous(), w4)


def _pad_halves(w, hp):
    """(2*hid, ...) -> (2*hp, ...): each half zero-padded to hp rows (the kernels' padded GDFN layout)."""
    hid = w.shape[0] // 2
    z = w.new_zeros((hp - hid,) + tuple(w.shape[1:]))
    return torch.cat([w[:hid], z, w[hid:], z], 0)


def to_windows(x):
    B, H, W, C = x.shape
    return x.reshape(B, H // 8, 8, W // 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 64, C)


def from_windows(xw, B, H, W):
    C = xw.shape[-1]
    return xw.reshape(B, H // 8, W // 8, 8, 8, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)


def _rel_index(device):
    t = torch.arange(64, device=device)
    y, x = t // 8, t % 8
    return (y[:, None] - y[None, :] + 7) * 15 + (x[:, None] - x[None, :] + 7)


def _shift_mask(H, W, device):
    def reg(n):
        c = torch.arange(n, device=device)
        return (c >= n - 8).long() + (c >= n - 4).long()
    ids = (3 * reg(H)[:, None] + reg(W)[None, :]).reshape(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
    return (ids[:, None, :] != ids[:, :, None]).float() * -100.0


def channel_attention(q, k, v, temperature, w_out, heads):
    B, H, W, C = q.shape
    hd, dt = C // heads, q.dtype

    def split(t):
        return t.reshape(B, H * W, heads, hd).permute(0, 2, 3, 1).float()
    q, k, v = split(q), split(k), split(v)
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    attn = torch.softmax((q @ k.transpose(-1, -2)) * temperature.reshape(1, heads, 1, 1).float(), dim=-1)
    out = (attn @ v).permute(0, 3, 1, 2).reshape(B, H, W, C).to(dt)
    return pw(out, w_out)


def spectral_attention(P, pre, x, heads):
    C = x.shape[-1]
    t = dw3(pw(x, P[pre + "qkv.weight"]), P[pre + "qkv_dwconv.weight"])
    return channel_attention(t[..., :C], t[..., C:2 * C], t[..., 2 * C:], P[pre + "temperature"],
                             P[pre + "project_out.weight"], heads).to(x.dtype)


def cross_attention(P, pre, xq, xkv, heads):
    C = xq.shape[-1]
    q = dw3(pw(xq, P[pre + "q.weight"]), P[pre + "q_dwconv.weight"])
    kv = dw3(pw(xkv, P[pre + "kv.weight"]), P[pre + "kv_dwconv.weight"])
    return channel_attention(q, kv[..., :C], kv[..., C:], P[pre + "temperature"], P[pre + "project_out.weight"], heads).to(xq.dtype)


def gdfn(P, pre, x):
    w_in, w_dw, w_out = P[pre + "project_in.weight"], P[pre + "dwconv.weight"], P[pre + "project_out.weight"]
    hid = w_in.shape[0] // 2
    hp = (hid + 31) // 32 * 32
    t = dw3(pw(x, _pad_halves(w_in, hp)), _pad_halves(w_dw, hp))
    u = F.gelu(t[..., :hp].float()).to(x.dtype) * t[..., hp:]
    return pw(u, F.pad(w_out.reshape(w_out.shape[0], hid), (0, hp - hid)))


def gated_mlp(P, pre, x):
    h = x @ P[pre + "fc1.weight"].to(x.dtype).t() + P[pre + "fc1.bias"].to(x.dtype)
    hid = h.shape[-1] // 2
    g = h[..., :hid] * F.gelu(h[..., hid:].float()).to(x.dtype)
    return g @ P[pre + "fc2.weight"].to(x.dtype).t() + P[pre + "fc2.bias"].to(x.dtype)


def window_attention(P, pre, xw, heads, mask):
    nwb, N, C = xw.shape
    hd = C // heads
    qkv = xw @ P[pre + "qkv.weight"].to(xw.dtype).t() + P[pre + "qkv.bias"].to(xw.dtype)
    qkv = qkv.reshape(nwb, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = (q @ k.transpose(-1, -2)).float()
    bias = P[pre + "relative_position_bias_table"][_rel_index(xw.device).reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias[None].float()
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.reshape(nwb // nW, nW, heads, N, N) + mask[None, :, None]).reshape(nwb, heads, N, N)
    attn = torch.softmax(attn, dim=-1).to(xw.dtype)
    out = (attn @ v).transpose(1, 2).reshape(nwb, N, C)
    return out @ P[pre + "proj.weight"].to(xw.dtype).t() + P[pre + "proj.bias"].to(xw.dtype)


def pg_gate(P, pre, xw):
    return pg_gate_from_mean(P, pre, xw.float().mean(dim=1))


def pg_gate_from_mean(P, pre, mu):
    """the per-window gate as a function of the window mean (B*nW, C) -- tiny fp32 math."""
    r = P[pre + "linear_down.weight"].shape[0]
    w = torch.softmax(mu @ P[pre + "linear_prompt.weight"].float().t(), dim=-1)
    s = w @ P[pre + "prompt_param"].float().reshape(PROMPT_LEN, r)
    q = s @ P[pre + "q.weight"].float().t()
    kv = (mu @ P[pre + "linear_down.weight"].float().t()) @ P[pre + "kv.weight"].float().t()
    k, v = kv[:, :r], kv[:, r:]
    a = torch.softmax(q[:, :, None] * k[:, None, :] * r ** -0.5, dim=-1)
    o = (a * v[:, None, :]).sum(-1)
    o = o @ P[pre + "proj.weight"].float().t() + P[pre + "proj.bias"].float()
    return o @ P[pre + "linear_up.weight"].float().t()


def pgsstb_attn(P, x, heads, shift, k1=None):
    """first residual branch of the block: x + DropPath(local + global spectral of the window attention)."""
    B, H, W, C = x.shape
    xn = layer_norm(x, P["norm1.weight"], P["norm1.bias"])
    if shift:
        xn = torch.roll(xn, (-SHIFT, -SHIFT), (1, 2))
    mask = _shift_mask(H, W, x.device) if shift else None
    sa = window_attention(P, "attn.", to_windows(xn), heads, mask)
    gate = pg_gate(P, "local_spectral_attn.", sa)
    local = sa * gate[:, None, :].to(sa.dtype)

    def img(t):
        t = from_windows(t, B, H, W)
        return torch.roll(t, (SHIFT, SHIFT), (1, 2)) if shift else t
    sa_img = img(sa)
    branch = img(local) + spectral_attention(P, "gobal_spectral_attn.", sa_img, heads)
    if k1 is not None:
        branch = branch * k1.reshape(B, 1, 1, 1).to(branch.dtype)
    return x + branch


def mlp_branch(P, y, k2=None):
    """second residual branch: y + DropPath(GatedMlp(LN2(y)))."""
    m = gated_mlp(P, "mlp.", layer_norm(y, P["norm2.weight"], P["norm2.bias"]))
    if k2 is not None:
        m = m * k2.reshape(-1, 1, 1, 1).to(m.dtype)
    return y + m


def cross_transformer(P, text, vis, heads=2):
    a = text + cross_attention(P, "attn.", layer_norm(text, P["norm11.body.weight"], P["norm11.body.bias"]),
                               layer_norm(vis, P["norm12.body.weight"], P["norm12.body.bias"]), heads)
    return a + gdfn(P, "ffn.", layer_norm(a, P["norm2.body.weight"], P["norm2.body.bias"]))


def transformer_block(P, t, heads):
    t = t + spectral_attention(P, "attn.", layer_norm(t, P["norm1.body.weight"], P["norm1.body.bias"]), heads)
    return t + gdfn(P, "ffn.", layer_norm(t, P["norm2.body.weight"], P["norm2.body.bias"]))
