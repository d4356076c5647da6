"""per-kernel timings at the shapes of one refinement-stage block (B=32, 64x64, C=128, 2 heads) and one latent block."""
import sys, time, warnings
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
warnings.filterwarnings("ignore")
import torch
from mp_hsir_amd import ops
from mp_hsir_amd.net.MP_HSIR import PGSSTB

dev = torch.device("cuda")
dt = torch.bfloat16

def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for (B, H, C, heads, cr) in [(32, 64, 128, 2, 8), (32, 32, 128, 4, 16), (32, 16, 256, 8, 32), (32, 64, 64, 2, 8)]:
    W = H
    M = B * H * W
    print("==== B=%d %dx%d C=%d heads=%d  (M=%d tokens)" % (B, H, W, C, heads, M))
    torch.manual_seed(0)
    blk = PGSSTB(C, heads, [64, 64], 8, 4, 0.0, 2.66, cr, 128).to(dev)
    pk, sp = blk.packed(dt), blk.gobal_spectral_attn.packed(dt)
    x = torch.randn(B, H, W, C, device=dev, dtype=dt)
    x2 = x.reshape(-1, C)
    es = 2
    def rep(name, us, flops=None, nbytes=None):
        s = "%-26s %9.1f us" % (name, us)
        if flops: s += "  %7.1f TFLOP/s" % (flops / us / 1e6)
        if nbytes: s += "  %7.2f TB/s (algorithmic)" % (nbytes / us / 1e6)
        print(s)
    f = lambda: ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"], pk["bproj"], pk["pg"], heads, 4)
    rep("win_attn_fwd", t_us(f), M * (8.0 * C * C + 256.0 * C), 2.0 * M * C * es)
    sa, gate, mu, oattn = ops.win_attn_fwd(x, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wproj"], pk["bproj"], pk["pg"], heads, 4, save=True)
    sa2 = sa.reshape(-1, C)
    f = lambda: ops.gemm_tok(sa2, sp["wqkv"])
    rep("gemm_tok qkv (N=3C)", t_us(f), 6.0 * M * C * C, 4.0 * M * C * es)
    t = ops.gemm_tok(sa2, sp["wqkv"]); w9 = sp["w9"]
    f = lambda: ops.dwconv_gram(t[:, :C], t[:, C:2*C], t[:, 2*C:], w9[:, :C], w9[:, C:2*C], w9[:, 2*C:], 3*C, B, H, W, C, heads)
    rep("dwconv_gram", t_us(f), M * (54.0 * C + 2.0 * C * C / heads), 4.0 * M * C * es)
    v, gp, spart, _ = f()
    f = lambda: ops.spectral_fold(gp, spart, sp["temp"], sp["wo"], dt, transposed=True)
    rep("spectral_fold", t_us(f))
    Mb, MbT, gp, spart = f()
    f = lambda: ops.gemm_tok(v, Mb, epi=2, res=x2, sa=sa2, gate=gate, geom=(H, W, 4))
    rep("gemm_tok apply+combine", t_us(f), 2.0 * M * C * C, 4.0 * M * C * es)
    f = lambda: ops.gated_mlp_fwd(x2, pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"], pk["W2"], pk["b2"])
    HP = pk["W2"].shape[1]
    rep("gated_mlp_fwd", t_us(f), 6.0 * M * C * HP, 2.0 * M * C * es)
    dy = torch.randn_like(x)
    f = lambda: ops.gated_mlp_bwd(x2, dy.reshape(-1, C), dy.reshape(-1, C), pk["ln2"][0], pk["ln2"][1], pk["W1"], pk["b1"], pk["W1T"], pk["W2T"])
    rep("gated_mlp_bwd", t_us(f), 12.0 * M * C * HP, (4.0 * M * C + 3.0 * M * HP) * es)
    dx, xn, h, dpre, part = f()
    f = lambda: ops.gemm_tn(dpre, xn)
    rep("gemm_tn dW1 (2HPxC)", t_us(f), 4.0 * M * HP * C, M * (2 * HP + C) * es)
    f = lambda: ops.gemm_tn(dy.reshape(-1, C), h)
    rep("gemm_tn dW2 (CxHP)", t_us(f), 2.0 * M * HP * C, M * (HP + C) * es)
    f = lambda: ops.combine_bwd(dy, sa, gate, None, 4)
    rep("combine_bwd", t_us(f), None, 3.0 * M * C * es)
    f = lambda: ops.gemm_tn(dy.reshape(B, H * W, C), v.reshape(B, H * W, C))
    rep("gemm_tn dM (batched)", t_us(f), 2.0 * M * C * C, 2.0 * M * C * es)
    dM = f()
    f = lambda: ops.spectral_fold_bwd(gp, spart, sp["temp"], sp["wo"], dM, dt)
    rep("spectral_fold_bwd", t_us(f))
    W2, _, _ = f()
    t4 = t.reshape(B, H, W, 3 * C)
    f = lambda: ops.dwconv3x3(t4[..., :2 * C], w9[:, :2 * C])
    rep("dwconv3x3 (2C)", t_us(f), None, 4.0 * M * C * es)
    qk = f()
    dall = torch.empty((M, 3 * C), device=dev, dtype=dt)
    f = lambda: ops.gemm_tok(qk.reshape(M, 2 * C), W2, out=dall[:, :2 * C])
    rep("gemm_tok dq|dk (2Cx2C)", t_us(f), 8.0 * M * C * C, 4.0 * M * C * es)
    f = lambda: ops.gemm_tok(dy.reshape(M, C), MbT, out=dall[:, 2 * C:])
    rep("gemm_tok dv", t_us(f), 2.0 * M * C * C, 2.0 * M * C * es)
    dall4 = dall.reshape(B, H, W, 3 * C)
    f = lambda: ops.dwconv3x3(dall4, w9, flip=True)
    rep("dwconv3x3 flip (3C)", t_us(f), None, 6.0 * M * C * es)
    f = lambda: ops.dwconv3x3_wgrad(t4, dall4)
    rep("dwconv3x3_wgrad (3C)", t_us(f), None, 6.0 * M * C * es)
    f = lambda: ops.win_attn_bwd(x, dy, mu, pk["ln1"][0], pk["ln1"][1], pk["wqkv"], pk["bqkv"], pk["rpb"], pk["wprojT"], heads, 4)
    rep("win_attn_bwd", t_us(f), M * (12.0 * C * C + 640.0 * C), 7.0 * M * C * es)
    dqkv, xnw, dsat, drpb = f()
    f = lambda: dqkv @ pk["wqkv"]
    rep("torch matmul dxn", t_us(f), 6.0 * M * C * C, 4.0 * M * C * es)
    f = lambda: ops.gemm_tn(dqkv, xnw)
    rep("gemm_tn dWqkv (3CxC)", t_us(f), 6.0 * M * C * C, 4.0 * M * C * es)
    f = lambda: ops.ln_bwd_win(x, xnw, dy, pk["ln1"][0], 4)
    rep("ln_bwd_win", t_us(f), None, 4.0 * M * C * es)
    f = lambda: ops.pg_gate_bwd(mu, gate, pk["pg"])
    rep("pg_gate_bwd (+gemm_tn)", t_us(f))
    f = lambda: torch.sum(dqkv, dim=0, dtype=torch.float32)
    rep("torch colsum dqkv", t_us(f), None, 3.0 * M * C * es)
