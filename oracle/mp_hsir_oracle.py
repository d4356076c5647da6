"""CPU oracle for the MP-HSIR forward/backward hot path.   *** TEST INFRASTRUCTURE ONLY ***

A functional, channels-last, plain-PyTorch (fp32 or fp64, CPU) restatement of what
/root/reference/net/MP_HSIR.py computes, written from the behavioural description in SURVEY.md
§8a / Appendix A -- not a copy of the reference modules: there are no nn.Modules here, weights
arrive as a flat ``{state_dict key: tensor}`` mapping and activations are (B, H, W, C).

Who may import this file: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- as the checker (or as the timed CPU baseline), never as the product path.  The
product (``mp-hsir_amd/``) must not import it and fails loudly without its HIP library.

Parity pin: every function below is checked against golden vectors produced by running the real
reference on CPU in the build container (tests/golden/make_golden.py -> tests/golden/*.npz;
tests/test_oracle_vs_golden.py, rel-L2 <= 1e-6 in fp64).  The reference itself ships no tests or
vectors for this path, so those generated fixtures are the only pin.

All `file:line` citations are relative to /root/reference/.
"""
import math

import torch
import torch.nn.functional as F

WINDOW = 8            # window_size=[8,8,8]                      net/MP_HSIR.py:769
SHIFT = WINDOW // 2   # odd blocks shift by window_size // 2      net/MP_HSIR.py:748
PROMPT_LEN = 128      # prompt_len=128                            net/MP_HSIR.py:791
FFN_FACTOR = 2.66     # ffn_expansion_factor / mlp_ratio          net/MP_HSIR.py:773
LN_EPS = 1e-5         # nn.LayerNorm default / hand-rolled LN      net/MP_HSIR.py:357,618


# ----------------------------------------------------------------------------------------------
# configuration bookkeeping (what the constructor derives, net/MP_HSIR.py:764-808)
# ----------------------------------------------------------------------------------------------
def make_cfg(in_channel=31, out_channel=31, dim=64, num_blocks=(2, 4, 6), window_size=(8, 8, 8),
             task_classes=6, num_refinement_blocks=4, heads=(2, 4, 8), ffn_expansion_factor=2.66,
             bias=False):
    assert tuple(window_size) == (8, 8, 8) and not bias
    if task_classes not in (1, 6, 7):
        raise ValueError("task_classes must be 6 or 7")  # net/MP_HSIR.py:507-508
    nb = list(num_blocks)
    n = sum(nb)
    dpr = [0.1 * i / (n - 1) if n > 1 else 0.0 for i in range(n)]  # linspace(0,0.1,n)  :780
    stages = {
        # name: (C, heads, compress_ratio, depth, drop-path rates)           :791-805
        "encoder_level1": (dim, heads[0], 8, nb[0], dpr[0:nb[0]]),
        "encoder_level2": (dim * 2, heads[1], 16, nb[1], dpr[nb[0]:nb[0] + nb[1]]),
        "latent": (dim * 4, heads[2], 32, nb[2], dpr[nb[0] + nb[1]:n]),
        "decoder_level2": (dim * 2, heads[1], 16, nb[1], dpr[nb[0]:nb[0] + nb[1]]),
        "decoder_level1": (dim * 2, heads[0], 8, nb[0], dpr[0:nb[0]]),
        "refinement": (dim * 2, heads[0], 8, num_refinement_blocks, dpr[nb[0]:nb[0] + nb[1]]),
    }
    return dict(in_channel=in_channel, out_channel=out_channel, dim=dim, task_classes=task_classes,
                stages=stages, ffn=ffn_expansion_factor)


# ----------------------------------------------------------------------------------------------
# small building blocks
# ----------------------------------------------------------------------------------------------
def layer_norm_c(x, weight, bias):
    """LayerNorm over the last (channel) axis: biased variance, eps 1e-5 inside the sqrt.
    Covers both flavours on the path: nn.LayerNorm (net/MP_HSIR.py:618-619) and
    WithBias_LayerNorm (net/MP_HSIR.py:341-357)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * weight + bias


def gelu_erf(x):
    """Exact (erf) GELU, nn.GELU()/F.gelu defaults (net/MP_HSIR.py:72,263,389)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def pointwise(x, w):
    """1x1 conv without bias on channels-last data: w is the (Cout, Cin, 1, 1) conv weight."""
    return x @ w.reshape(w.shape[0], w.shape[1]).t()


def conv_nhwc(x, w, padding, groups=1):
    """Dense / depthwise 3x3 (or 1x1) conv, stride 1, zero padding, no bias, channels-last in/out."""
    return F.conv2d(x.permute(0, 3, 1, 2), w, None, 1, padding, 1, groups).permute(0, 2, 3, 1)


def depthwise3x3(x, w):
    """nn.Conv2d(C, C, 3, padding=1, groups=C, bias=False) (net/MP_HSIR.py:92,227,230,257,382)."""
    return conv_nhwc(x, w, 1, groups=w.shape[0])


def pixel_unshuffle2(x):
    """nn.PixelUnshuffle(2) on channels-last: out[.., c*4+dy*2+dx] = in[2y+dy, 2x+dx, c] (:437)."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4)
    return x.reshape(B, H // 2, W // 2, C * 4)


def pixel_shuffle2(x):
    """nn.PixelShuffle(2), inverse of the above (net/MP_HSIR.py:447)."""
    B, H, W, C4 = x.shape
    C = C4 // 4
    x = x.reshape(B, H, W, C, 2, 2).permute(0, 1, 4, 2, 5, 3)
    return x.reshape(B, H * 2, W * 2, C)


# ----------------------------------------------------------------------------------------------
# window geometry (closed forms)
# ----------------------------------------------------------------------------------------------
def to_windows(x):
    """(B,H,W,C) -> (B*nW, 64, C); windows image-major then row-major (net/MP_HSIR.py:21-30)."""
    B, H, W, C = x.shape
    x = x.reshape(B, H // WINDOW, WINDOW, W // WINDOW, WINDOW, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(-1, WINDOW * WINDOW, C)


def from_windows(xw, B, H, W):
    """inverse of to_windows (net/MP_HSIR.py:33-44)."""
    C = xw.shape[-1]
    x = xw.reshape(B, H // WINDOW, W // WINDOW, WINDOW, WINDOW, C).permute(0, 1, 3, 2, 4, 5)
    return x.reshape(B, H, W, C)


def relative_position_index():
    """idx[i,j] = (y_i - y_j + 7)*15 + (x_i - x_j + 7), tokens row-major (net/MP_HSIR.py:172-182)."""
    t = torch.arange(WINDOW * WINDOW)
    y, x = t // WINDOW, t % WINDOW
    return (y[:, None] - y[None, :] + WINDOW - 1) * (2 * WINDOW - 1) + (x[:, None] - x[None, :] + WINDOW - 1)


def shift_mask(H, W, dtype):
    """(nW,64,64) additive mask of the shifted blocks: 0 inside a region, -100 across regions
    (net/MP_HSIR.py:639-660; -100 not -inf).  Regions are cut at L-8 and L-4."""
    def reg(n):
        c = torch.arange(n)
        return (c >= n - WINDOW).long() + (c >= n - SHIFT).long()
    ids = (3 * reg(H)[:, None] + reg(W)[None, :]).to(dtype).reshape(1, H, W, 1)
    idw = to_windows(ids).squeeze(-1)                      # (nW,64)
    diff = idw[:, None, :] - idw[:, :, None]
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


# ----------------------------------------------------------------------------------------------
# a2  GatedMlp                                                        net/MP_HSIR.py:66-82
# ----------------------------------------------------------------------------------------------
def gated_mlp(P, pre, x):
    h = x @ P[pre + "fc1.weight"].t() + P[pre + "fc1.bias"]
    hid = h.shape[-1] // 2
    val, gate = h[..., :hid], h[..., hid:]       # first half = value, second half gated (Q12)
    return (val * gelu_erf(gate)) @ P[pre + "fc2.weight"].t() + P[pre + "fc2.bias"]


# ----------------------------------------------------------------------------------------------
# a3  Spatial_Attention (8x8 window MSA)                              net/MP_HSIR.py:158-218
# ----------------------------------------------------------------------------------------------
def spatial_attention(P, pre, xw, heads, mask):
    nwb, N, C = xw.shape
    hd = C // heads
    qkv = xw @ P[pre + "qkv.weight"].t() + P[pre + "qkv.bias"]
    qkv = qkv.reshape(nwb, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]                     # q scaled first (:198)
    attn = q @ k.transpose(-1, -2)                                     # (nwb,h,64,64)
    table = P[pre + "relative_position_bias_table"]                    # (225,h)
    bias = table[relative_position_index().reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias[None]
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.reshape(nwb // nW, nW, heads, N, N) + mask[None, :, None]).reshape(nwb, heads, N, N)
    attn = torch.softmax(attn, dim=-1)
    out = (attn @ v).transpose(1, 2).reshape(nwb, N, C)
    return out @ P[pre + "proj.weight"].t() + P[pre + "proj.bias"]


# ----------------------------------------------------------------------------------------------
# a4  PG_Spectral_Attention (local low-rank spectral-prompt gate)     net/MP_HSIR.py:116-155
# ----------------------------------------------------------------------------------------------
def pg_spectral_gate(P, pre, xw):
    """per-window channel gate g (nWB, C); the module output is xw * g[:, None, :] (Q13)."""
    mu = xw.mean(dim=1)                                                # (nwb,C)          :135
    w = torch.softmax(mu @ P[pre + "linear_prompt.weight"].t(), dim=-1)        # (nwb,128)  :136
    r = P[pre + "linear_down.weight"].shape[0]
    s = w @ P[pre + "prompt_param"].reshape(PROMPT_LEN, r)             # (nwb,r)          :139-140
    q = s @ P[pre + "q.weight"].t()                                    #                  :142
    kv = (mu @ P[pre + "linear_down.weight"].t()) @ P[pre + "kv.weight"].t()   #          :137,143
    k, v = kv[:, :r], kv[:, r:]
    a = torch.softmax(q[:, :, None] * k[:, None, :] * r ** -0.5, dim=-1)       # (nwb,r,r) :146-147
    o = (a * v[:, None, :]).sum(-1)                                    #                  :149
    o = o @ P[pre + "proj.weight"].t() + P[pre + "proj.bias"]          #                  :151
    return o @ P[pre + "linear_up.weight"].t()                         #                  :152


# ----------------------------------------------------------------------------------------------
# a5/a6/a8  channel ("spectral") attention over H*W                   net/MP_HSIR.py:85-114,220-249,289-322
# ----------------------------------------------------------------------------------------------
def _channel_attention_core(q, k, v, temperature, w_out, heads):
    """q,k,v (B,H,W,C) after the depthwise conv.  L2-normalise q,k over pixels (eps 1e-12, Q10),
    per-head hd x hd Gram * temperature, softmax over the k-channel axis, apply to v, project."""
    B, H, W, C = q.shape
    hd = C // heads

    def split(t):  # -> (B, heads, hd, N)
        return t.reshape(B, H * W, heads, hd).permute(0, 2, 3, 1)
    q, k, v = split(q), split(k), split(v)
    q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    attn = torch.softmax((q @ k.transpose(-1, -2)) * temperature.reshape(1, heads, 1, 1), dim=-1)
    out = (attn @ v).permute(0, 3, 1, 2).reshape(B, H, W, C)
    return pointwise(out, w_out)


def spectral_attention(P, pre, x, heads):
    """Spectral_Attention (:85-114) == Attention/MDTA (:289-322, dup :394-427)."""
    t = depthwise3x3(pointwise(x, P[pre + "qkv.weight"]), P[pre + "qkv_dwconv.weight"])
    C = x.shape[-1]
    return _channel_attention_core(t[..., :C], t[..., C:2 * C], t[..., 2 * C:], P[pre + "temperature"],
                                   P[pre + "project_out.weight"], heads)


def cross_attention(P, pre, xq, xkv, heads):
    """CrossAttention (:220-249): q from xq, k/v from xkv, separate 1x1 + depthwise convs."""
    q = depthwise3x3(pointwise(xq, P[pre + "q.weight"]), P[pre + "q_dwconv.weight"])
    kv = depthwise3x3(pointwise(xkv, P[pre + "kv.weight"]), P[pre + "kv_dwconv.weight"])
    C = xq.shape[-1]
    return _channel_attention_core(q, kv[..., :C], kv[..., C:], P[pre + "temperature"],
                                   P[pre + "project_out.weight"], heads)


# ----------------------------------------------------------------------------------------------
# a7  FFN == FeedForward (gated depthwise-conv feed-forward)          net/MP_HSIR.py:251-265,374-391
# ----------------------------------------------------------------------------------------------
def gdfn(P, pre, x):
    t = depthwise3x3(pointwise(x, P[pre + "project_in.weight"]), P[pre + "dwconv.weight"])
    hid = t.shape[-1] // 2
    return pointwise(gelu_erf(t[..., :hid]) * t[..., hid:], P[pre + "project_out.weight"])  # gelu(x1)*x2 (Q12)


# ----------------------------------------------------------------------------------------------
# a15/a16  PGSSTB and BaseBlock                                       net/MP_HSIR.py:601-761
# ----------------------------------------------------------------------------------------------
def pgsstb(P, pre, x, heads, shifted, keep=None, intermediates=None):
    """One prompt-guided spatial-spectral transformer block on (B,H,W,C).

    keep: optional (mask1, mask2) per-sample DropPath factors, each (B,) already divided by the
    keep probability (timm DropPath semantics); None = eval mode (identity)."""
    B, H, W, C = x.shape
    xn = layer_norm_c(x, P[pre + "norm1.weight"], P[pre + "norm1.bias"])            # :667
    if shifted:
        xn = torch.roll(xn, shifts=(-SHIFT, -SHIFT), dims=(1, 2))                    # :672
    mask = shift_mask(H, W, x.dtype) if shifted else None                           # :680-683
    sa = spatial_attention(P, pre + "attn.", to_windows(xn), heads, mask)           # (B*nW,64,C)
    gate = pg_spectral_gate(P, pre + "local_spectral_attn.", sa)
    local = sa * gate[:, None, :]                                                    # :153,687

    def to_image(tw):
        t = from_windows(tw, B, H, W)
        return torch.roll(t, shifts=(SHIFT, SHIFT), dims=(1, 2)) if shifted else t   # :694,709
    sa_img, local_img = to_image(sa), to_image(local)
    glob = spectral_attention(P, pre + "gobal_spectral_attn.", sa_img, heads)       # :701
    branch = local_img + glob                                                        # :715
    if keep is not None:
        branch = branch * keep[0].reshape(B, 1, 1, 1)
    y = x + branch                                                                   # :718
    m = gated_mlp(P, pre + "mlp.", layer_norm_c(y, P[pre + "norm2.weight"], P[pre + "norm2.bias"]))
    if keep is not None:
        m = m * keep[1].reshape(B, 1, 1, 1)
    if intermediates is not None:
        intermediates.update(sa=sa, gate=gate, x1=local, x2=glob, mlp=m, sa_img=sa_img)
    return y + m                                                                     # :719


def base_block(P, pre, x, heads, depth, keeps=None):
    y = x
    for i in range(depth):                                                           # :746-759
        y = pgsstb(P, "%sblocks.%d." % (pre, i), y, heads, shifted=(i % 2 == 1),
                   keep=None if keeps is None else keeps[i])
    return y + x                                                                     # :760


# ----------------------------------------------------------------------------------------------
# a12-a14  prompts                                                    net/MP_HSIR.py:481-599
# ----------------------------------------------------------------------------------------------
def text_prompt(task_id, task_classes, clip_table):
    """Text_Prompt.forward (:517-532).  clip_table is the injected (T,512) text-embedding table
    (a plain tensor in the reference).  Returns (clip_prompt (B,512), weights (B,T))."""
    if task_id.dim() > 1:   # training path: mean of one-hots over the id list (float)     :519-523
        w = F.one_hot(task_id, task_classes).to(clip_table.dtype).mean(dim=1)
    else:                   # test path: int64 one-hot                                      :525
        w = F.one_hot(task_id, task_classes)
    clip = (w.unsqueeze(-1) * clip_table.unsqueeze(0)).mean(dim=1)                         # :529-530
    return clip, w


def tvsp(P, pre, x, clip_prompt, weights, ps):
    """TVSP.forward (:572-583) incl. the batch-coupling broadcast."""
    B, H, W, D = x.shape
    learn = P[pre + "text_prompt_learnable"][0, :, :, 0, 0]                # (T,D)
    L = (weights.to(learn.dtype).unsqueeze(-1) * learn.unsqueeze(0)).mean(dim=1)           # (B,D) :575-576
    # (B,D,1,1) * (B,512) broadcasts to (B,D,B,512); nearest-resize to (ps,ps)              :576-577
    rows = torch.div(torch.arange(ps) * B, ps, rounding_mode="floor")
    cols = torch.div(torch.arange(ps) * 512, ps, rounding_mode="floor")
    clip_map = clip_prompt[rows][:, cols]                                  # (ps,ps)
    text = clip_map[None, :, :, None] * L[:, None, None, :]                # (B,ps,ps,D)
    vis = P[pre + "visual_prompt"].permute(0, 2, 3, 1).expand(B, ps, ps, D)                # :578
    ct = pre + "cross_transformer."
    # CrossTransformer (:267-287), cross_residual=True, heads=2 (:565)
    a = text + cross_attention(P, ct + "attn.",
                               layer_norm_c(text, P[ct + "norm11.body.weight"], P[ct + "norm11.body.bias"]),
                               layer_norm_c(vis, P[ct + "norm12.body.weight"], P[ct + "norm12.body.bias"]), 2)
    y = a + gdfn(P, ct + "ffn.", layer_norm_c(a, P[ct + "norm2.body.weight"], P[ct + "norm2.body.bias"]))
    if (H, W) != (ps, ps):                                                 # bilinear, align_corners=False :580
        y = F.interpolate(y.permute(0, 3, 1, 2), (H, W), mode="bilinear").permute(0, 2, 3, 1)
    return conv_nhwc(y, P[pre + "conv_last.weight"], 1)                    # :581


def prompt_fusion(P, pre, x, prompt, heads):
    """PromptFusion.forward (:594-599): cat -> TransformerBlock (:466-479) -> 1x1 conv."""
    t = torch.cat([x, prompt], dim=-1)
    tr = pre + "transformer."
    t = t + spectral_attention(P, tr + "attn.",
                               layer_norm_c(t, P[tr + "norm1.body.weight"], P[tr + "norm1.body.bias"]), heads)
    t = t + gdfn(P, tr + "ffn.", layer_norm_c(t, P[tr + "norm2.body.weight"], P[tr + "norm2.body.bias"]))
    return pointwise(t, P[pre + "conv.weight"])


# ----------------------------------------------------------------------------------------------
# a17  the network                                                    net/MP_HSIR.py:810-844
# ----------------------------------------------------------------------------------------------
def mp_hsir_forward(P, cfg, inp_img, task_id, clip_table, keeps=None):
    """inp_img (B,C,H,W) like the reference surface; returns (B,C,H,W)."""
    st = cfg["stages"]
    T = cfg["task_classes"]
    dim = cfg["dim"]
    x_in = inp_img.permute(0, 2, 3, 1)
    clip, w = text_prompt(task_id, T, clip_table.to(inp_img.dtype))

    def stage(name, t):
        C, heads, _cr, depth, _dpr = st[name]
        return base_block(P, name + ".", t, heads, depth, None if keeps is None else keeps[name])

    def down(name, t):   # Downsample (:432-440)
        return pixel_unshuffle2(conv_nhwc(t, P[name + ".body.0.weight"], 1))

    def up(name, t):     # Upsample (:442-450)
        return pixel_shuffle2(conv_nhwc(t, P[name + ".body.0.weight"], 1))

    e1_in = conv_nhwc(x_in, P["patch_embed.proj.weight"], 1)               # :814
    e1 = stage("encoder_level1", e1_in)
    e2 = stage("encoder_level2", down("down1_2", e1))
    lat = stage("latent", down("down2_3", e2))
    d2_in = up("up3_2", lat)
    p2 = tvsp(P, "prompt2.", e2, clip, w, 32)                              # :825
    f2 = prompt_fusion(P, "fusion2.", e2, p2, 8)                           # :826
    d2 = stage("decoder_level2", pointwise(torch.cat([d2_in, f2], -1), P["reduce_chan_level2.weight"]))
    d1_in = up("up2_1", d2)
    p1 = tvsp(P, "prompt1.", e1, clip, w, 64)                              # :833
    f1 = prompt_fusion(P, "fusion1.", e1, p1, 4)                           # :834
    d1 = stage("decoder_level1", torch.cat([d1_in, f1], -1))              # no channel reduce :835
    r = stage("refinement", d1)
    out = conv_nhwc(r, P["output.weight"], 1) + x_in                       # global residual :841
    return out.permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------------------------
# a18/a20  loss, optimiser, schedule, metric
# ----------------------------------------------------------------------------------------------
def l1_after_clamp(restored, clean):
    """train.py:58-61: clamp to [0,1] then nn.L1Loss (mean)."""
    return (restored.clamp(0, 1) - clean).abs().mean()


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-2):
    """torch.optim.AdamW defaults (train.py:69).  step is 1-based.  Returns new (p, m, v)."""
    p = p * (1.0 - lr * weight_decay)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    denom = (v.sqrt() / math.sqrt(1 - beta2 ** step)) + eps
    return p - (lr / (1 - beta1 ** step)) * m / denom, m, v


def warmup_cosine_lr(epoch, base_lr, epochs, eta_min=1e-6):
    """LinearWarmupCosineAnnealingLR as train.py:71-76 uses it (utils/schedulers.py:332-346 closed
    form), stepped per epoch: warmup = int(0.1*epochs), warmup_start_lr 0 -> lr(0) = 0 (Q19)."""
    wu = int(0.1 * epochs)
    if epoch < wu:
        return epoch * base_lr / (wu - 1) if wu > 1 else 0.0
    return eta_min + 0.5 * (base_lr - eta_min) * (1 + math.cos(math.pi * (epoch - wu) / (epochs - wu)))


def psnr_bandwise(restored, clean):
    """utils/val_utils.py:49-69: clip both to [0,1]; per band 10*log10(1/mse) (data_range=1);
    mean over bands, then over the batch.  (B,C,H,W) tensors -> python float."""
    r = restored.detach().double().clamp(0, 1)
    c = clean.detach().double().clamp(0, 1)
    mse = ((r - c) ** 2).mean(dim=(-1, -2))
    return float((10.0 * torch.log10(1.0 / mse)).mean(dim=1).mean())
