/* libmphsir -- C ABI of the MI355X-native MP-HSIR forward/backward hot path.
 *
 * The reference (ZhehuiWu/MP-HSIR) has no FFI on this path: everything sits behind the Python
 * nn.Module surface of net/MP_HSIR.py.  This header is the drop-in boundary *beneath* that surface:
 * one entry point per fused stock-op sequence of the reference (cited per function), plain device
 * pointers and sizes only -- no torch types.  The Python package mp-hsir_amd binds it with ctypes
 * (mp-hsir_amd/_lib.py); INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; nothing is allocated or freed here;
 *   - activations are channels-last: a (B,H,W,C) cube is a row-major [B*H*W][C] token matrix;
 *   - dtype: 0 = float32 (exact-f32 parity path), 1 = bfloat16 (storage bf16, fp32 accumulate);
 *     "T*" below means "element type selected by dtype"; float* is always fp32;
 *   - GEMM weights are row-major [N][K] (PyTorch Linear / 1x1-conv layout), already converted to
 *     the compute dtype and zero-padded by the caller to the multiples stated per function;
 *   - every call only enqueues work on `stream` (a hipStream_t) and returns 0, or a negative
 *     MPHSIR_E* code without launching anything; mphsir_last_error() describes the failure;
 *   - no global state except the optional launch profiler; re-entrant; graph-capturable.
 */
#ifndef MPHSIR_H
#define MPHSIR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPHSIR_OK 0
#define MPHSIR_EINVAL (-1)   /* bad shape / alignment / unsupported configuration */
#define MPHSIR_ELAUNCH (-2)  /* hipLaunchKernel reported an error */

#define MPHSIR_F32 0
#define MPHSIR_BF16 1
#define MPHSIR_F16 2   /* IEEE half storage, fp32 accumulation: the reference's 16-mixed precision (train.py:118) */

/* "1.0.0-gfx950"; never NULL. */
const char* mphsir_version(void);
/* Text of the last error on the calling thread ("" if none). */
const char* mphsir_last_error(void);
/* gcnArchName of the current HIP device (e.g. "gfx950:sramecc+:xnack-") into buf[n]. */
int mphsir_device_arch(char* buf, int n);

/* ---- token GEMM with fused prologue/epilogue ------------------------------------------------
 * Y[m][n] = epi( sum_k pro(X)[m][k] * W[n][k] )            m < M, n < N
 * Replaces the reference's nn.Conv2d(k=1) / nn.Linear call sites on the path:
 *   Spectral_Attention.qkv / project_out (net/MP_HSIR.py:91,93), CrossAttention.q/kv (:226,229),
 *   FFN/FeedForward.project_in/project_out (:256,258,380,384), PromptFusion.conv (:592),
 *   reduce_chan_level2 (:799), and -- with a per-sample W -- the folded
 *   softmax(QK^T)V + project_out of the channel attention (:107-113).
 * prologue: ln_w/ln_b != NULL -> LayerNorm over K (biased var, eps 1e-5; :341-357, :618) first.
 * epilogue (epi):
 *   0  Y = acc (+ bias)
 *   1  Y = R + acc (+ bias)                                      residual add (:282,286,476,477)
 *   2  Y = R + keep[b] * (SA * gate[window(m)] + acc)            PGSSTB branch sum (:715-718)
 * M % 64 == 0, N % 16 == 0, K % 32 == 0; ldx/ldy/ldr/ldsa in elements, multiples of 16 bytes.
 * w_batch_stride (elements) != 0 selects W + (m / rows_per_batch) * w_batch_stride per sample.   */
typedef struct mphsir_gemm_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* X; int64_t ldx;
    const void* W; int64_t w_batch_stride; int64_t rows_per_batch;
    const float* bias;
    const float* ln_w; const float* ln_b;
    void* Y; int64_t ldy;
    int64_t M, N, K;
    int epi;
    const void* R; int64_t ldr;
    const void* SA; int64_t ldsa;
    const float* gate;          /* [B*nW][N] fp32, epi 2 */
    const float* keep;          /* [B] fp32 DropPath factor (mask/keep_prob) or NULL */
    int32_t H, Wimg, shift;     /* image geometry for window(m), epi 2 */
    int32_t form;               /* 0 = the library chooses; 1 = one workgroup per 64-token tile; 2 = ring form: persistent workgroups
                                   that walk the token tiles, a loader wave streaming K-chunks by LDS-DMA ahead of four MFMA
                                   waves (16-bit types, no LayerNorm prologue; ignored where it does not apply) */
} mphsir_gemm_args;
int mphsir_gemm_tok(const mphsir_gemm_args* a, int dtype, void* stream);

/* ---- fused LayerNorm + shifted 8x8 window attention; the local spectral-prompt gate -----------------
 * mphsir_win_attn_fwd replaces, inside PGSSTB.forward (net/MP_HSIR.py:662-713): norm1 (:667), torch.roll(-shift) (:672),
 * window_partition (:21-30,:677), Spatial_Attention.forward (:193-218) incl. the relative-position
 * bias gather (:200-203) and the calculate_mask shift mask (:639-660, -100 across regions),
 * window_reverse + roll(+shift) (:690-696), and the window mean of PG_Spectral_Attention.forward (:135).
 * X, SA: (B,H,W,C) channels-last cubes; SA = window-attention output back in image order.
 * mu: [B*nW][C] fp32, the mean of SA over each window's 64 tokens, one row per window in shifted-frame window order
 *       (window_partition order).
 * Wqkv [3C][C] (attn.qkv.weight), bqkv [3C]; rpb = relative_position_bias_table [225][heads] fp32;
 * Wproj [C][heads*HDP]: attn.proj.weight with every head's hd input columns zero-padded to
 * HDP = mphsir_win_attn_hdp(hd, dtype).  H, W multiples of 8; shift 0 or 4; (C, C/heads) in {(32,32),(64,32),(64,64),
 * (128,32),(128,64),(256,32),(96,48),(192,48),(192,96),(384,48)}.
 * mphsir_pg_gate_fwd: the gate of PG_Spectral_Attention.forward (:136-152) for every window, 16 windows per workgroup:
 * gate [nW][C] fp32 from mu [nW][C]; the reference's x1 is SA * gate[window] (:153), applied by mphsir_gemm_tok epi 2.
 * The spectral-prompt weights are the fp32 parameters as stored: Wprompt [128][C], prompt_param [128][r], Wq [r][r],
 * Wkv [2r][r], Wdown [r][C], Wpproj [r][r] + bpproj [r], Wup [C][r]; C % 16 == 0, r <= 32.                          */
typedef struct mphsir_win_attn_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* X;
    const float* ln_w; const float* ln_b;
    const void* Wqkv; const float* bqkv;
    const float* rpb;
    const void* Wproj; const float* bproj;
    void* SA;
    float* mu;      /* [B*nW][C] fp32: per-window mean of SA (input of the gate; also saved for backward)            */
    void* Oattn;    /* optional [B*nW*64][C]: softmax(QK^T)V before proj, window-token order (for backward)  */
    int32_t B, H, W, C, heads, shift;
} mphsir_win_attn_args;
typedef struct mphsir_pg_fwd_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const float* mu;
    const float* Wprompt; const float* prompt_param; const float* Wq; const float* Wkv; const float* Wdown;
    const float* Wpproj; const float* bpproj; const float* Wup;
    float* gate;
    int32_t nW, C, r;
} mphsir_pg_fwd_args;
int mphsir_pg_gate_fwd(const mphsir_pg_fwd_args* a, void* stream);
int mphsir_win_attn_fwd(const mphsir_win_attn_args* a, int dtype, void* stream);
int mphsir_win_attn_hdp(int head_dim, int dtype);

/* ---- global "spectral" channel attention: pass A, fold, (pass B = mphsir_gemm_tok) ----------------
 * Reference: Spectral_Attention.forward (net/MP_HSIR.py:96-114) == Attention.forward (:301-322,
 * dup :406-427) and CrossAttention.forward (:234-249).  With t = conv1x1(x) computed by
 * mphsir_gemm_tok, mphsir_dwconv_gram applies the depthwise 3x3 (qkv_dwconv / q_dwconv / kv_dwconv),
 * writes v, and reduces q,k over the pixels into per-workgroup partials of the per-head Gram
 * G_h = q_h k_h^T (hd x hd) and of sum(q^2), sum(k^2) -- q and k never reach HBM.
 * Tq/Tk/Tv: row-major [B*H*W][ld*] views of the 1x1-conv output whose first channel is the first
 * q / k / v channel (self-attention: t, t+C, t+2C with ld = 3C; cross attention: two tensors).
 * wq/wk/wv: the matching depthwise taps as fp32 [9][ldw] (tap-major, channel-contiguous).
 * V: [B*H*W][ldvo].  Gpart: [B][nsplit][heads][hd][hd] fp32, Spart: [B][nsplit][2][C] fp32;
 * nsplit (workgroups per sample) must divide H*W/64.  Deterministic: no atomics.
 * mphsir_spectral_fold: per (sample, head) sums the partials in split order, applies F.normalize
 * (eps 1e-12, :104-105), temperature (:107), row softmax (:108) and folds project_out (:113):
 * M[b] = Wo * blockdiag_h(softmax(...)), written as [B][C][C] in the compute dtype -- the per-sample
 * weight of the pass-B mphsir_gemm_tok over V.  (C, C/heads) as for win_attn plus (32,16),(64,16),(128,16). */
typedef struct mphsir_gram_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* Tq; int64_t ldq; const void* Tk; int64_t ldk; const void* Tv; int64_t ldv;
    const float* wq; const float* wk; const float* wv; int64_t ldw;
    void* V; int64_t ldvo;
    float* Gpart; float* Spart;
    int32_t B, H, W, C, heads, nsplit;
    void* QK; int64_t ldqk;     /* optional [B*H*W][ldqk >= 2C]: q | k after the depthwise conv (training keeps them for the
                                   backward instead of recomputing them; only the sliding-window kernel form writes it)  */
} mphsir_gram_args;
int mphsir_dwconv_gram(const mphsir_gram_args* a, int dtype, void* stream);
int mphsir_dwconv_gram_keeps_qk(int32_t C, int32_t W, int dtype);     /* 1 if QK can be requested for this shape */

/* ---- fused pass A (inference): LayerNorm (optional) + 1x1 qkv conv + depthwise 3x3 + Gram / norms ------------
 * The same results as mphsir_gemm_tok (X -> t = [q|k|v] sources, 3C per pixel) followed by mphsir_dwconv_gram, with t
 * kept on-chip (Spectral_Attention.forward net/MP_HSIR.py:96-107, Attention.forward :301-313 incl. its norm1 :476):
 * one workgroup = one 8x16-pixel tile + one-pixel halo.  X [B*H*W][ldx]; Wqkv [3C][C] compute dtype; w9 fp32
 * [9][ldw >= 3C] taps of the q | k | v channels; V [B*H*W][ldvo]; Gpart [B][nsplit][heads][hd][hd], Spart
 * [B][nsplit][2][C] fp32 exactly as mphsir_dwconv_gram writes them (nsplit divides (H/8)*(W/16)), so
 * mphsir_spectral_fold and pass B follow unchanged.  Training passes T and QK to keep what its backward needs.
 * mphsir_qkv_dwconv_gram_fits: 1 if (C, heads, H, W, dtype) is covered (H % 8 == 0, W % 16 == 0, instantiated width). */
typedef struct mphsir_fused_gram_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* X; int64_t ldx;
    const float* ln_w; const float* ln_b;   /* both or neither: LayerNorm over C (eps 1e-5) applied to X first */
    const void* Wqkv;
    const float* w9; int64_t ldw;
    void* V; int64_t ldvo;
    float* Gpart; float* Spart;
    int32_t B, H, W, C, heads, nsplit;
    int32_t head_groups;                    /* 0/1: one workgroup runs all heads of its tiles; g > 1 (divides heads): g workgroups
                                               per tile set, heads/g heads each (small images: more workgroups)          */
    void* T; int64_t ldt;                   /* optional, both or neither (training): t = qkv(LN(x)) [B*H*W][ldt >= 3C] and    */
    void* QK; int64_t ldqk;                 /* q | k after the depthwise conv [B*H*W][ldqk >= 2C], kept for the backward      */
    int32_t row_segments;                   /* 0: tile form (8x16-pixel tiles, nsplit divides the tile count).  s > 0: ROW-WALKING
                                               form (mphsir_qkv_dwconv_gram_rows_fits): a workgroup walks H/s >= 4 rows of a 32-pixel
                                               column strip, t stays in registers; needs H % s == 0 and nsplit == (W/32)*s;
                                               head_groups is ignored (the form fixes its own head split)                   */
} mphsir_fused_gram_args;
int mphsir_qkv_dwconv_gram(const mphsir_fused_gram_args* a, int dtype, void* stream);
int mphsir_qkv_dwconv_gram_fits(int32_t C, int32_t heads, int32_t H, int32_t W, int dtype);
int mphsir_qkv_dwconv_gram_rows_fits(int32_t C, int32_t heads, int32_t H, int32_t W, int dtype, int32_t with_ln);   /* row-walking
                                           form: 16-bit dtypes, W % 32 == 0, H >= 4, (C, head width) in {64,128}x{32,64}, {96,192}x48 and
                                           (256,32) (four-slot ring); with the LayerNorm prologue: C in {64,128,256} with 32-wide heads
                                           (spectral_rows.hip: rows_shape / rows_form_fits) */
/* Diagnostics, one entry point: arm (stamps = device buffer of >= 32 uint64) or disarm (NULL) the shader-clock phase stamps
 * that workgroup 0 of the next launches of one kernel family writes (tools/bench/bench_win.py [pg], tools/bench/bench_rows.py).  The only mutable global state of the library besides the launch timer (mphsir_prof_*).               */
enum { MPHSIR_DEBUG_PG_GATE = 0, MPHSIR_DEBUG_WIN_ATTN = 1, MPHSIR_DEBUG_FUSED_PASS_A = 2 };
int mphsir_debug(int kind, void* stamps);
typedef struct mphsir_fold_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const float* Gpart; const float* Spart;
    const float* temperature;   /* [heads] */
    const float* Wo;            /* project_out.weight [C][C] fp32 */
    void* M;                    /* [B][C][C] compute dtype */
    void* MT;                   /* optional [B][C][C]: M transposed (saved for backward) */
    int32_t B, C, heads, nsplit;
    float* Gsum; float* Ssum;   /* optional [B][heads][hd][hd], [B][2][C]: the reduced Gram / sums of squares, i.e.
                                   the same data with nsplit = 1 -- what mphsir_spectral_fold_bwd needs (saved for
                                   backward instead of the partials)                                           */
} mphsir_fold_args;
int mphsir_spectral_fold(const mphsir_fold_args* a, int dtype, void* stream);

/* ---- depthwise 3x3 + GELU gate of the GDFN feed-forward ------------------------------------------
 * U[:, c] = gelu_erf(dw(T)[:, c]) * dw(T)[:, HP + c]   (FFN/FeedForward.forward :261-263, :387-389:
 * gelu on the FIRST half).  T [B*H*W][2*HP] (ldt), w9 fp32 [9][2*HP] (ldw), U [B*H*W][HP] (ldu);
 * HP = hidden width zero-padded to a multiple of 32 on both halves.                                */
typedef struct mphsir_gate_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* T; int64_t ldt;
    const float* w9; int64_t ldw;
    void* U; int64_t ldu;
    int32_t B, H, W, HP;
} mphsir_gate_args;
int mphsir_dwconv_gate(const mphsir_gate_args* a, int dtype, void* stream);

/* ---- the whole GDFN feed-forward block in one launch ---------------------------------------------------
 * Y = X + project_out( gelu_erf(x1) * x2 ),  [x1|x2] = dwconv3x3(project_in(LayerNorm(X)))
 * Replaces `x + self.ffn(self.norm2(x))` of CrossTransformer.forward (net/MP_HSIR.py:286) and
 * TransformerBlock.forward (:477) with FFN / FeedForward.forward (:259-265 == :385-391): the 2*hid-wide
 * project_in output and the hid-wide gate product stay on chip (the three-launch form
 * mphsir_gemm_tok -> mphsir_dwconv_gate -> mphsir_gemm_tok moves 6.5x the bytes).
 * X, Y [B*H*W][D] channels-last (ldx, ldy), Y must not alias X; ln_w, ln_b fp32 [D];
 * Win [2*HP][D]: rows 0..hid-1 = project_in.weight[0:hid] (gelu side), rows HP..HP+hid-1 =
 * project_in.weight[hid:2hid], everything else zero; w9 fp32 [9][ldw >= 2*HP] taps laid out the same
 * way; Wout [D][HP] = project_out.weight zero-padded along K.  HP = hid rounded up to a multiple of 32.
 * 16-bit types, D in {64,128,192,256}, H % 8 == 0, W % 16 == 0 (mphsir_gdfn_fused_fits); nsplit
 * workgroups per sample walk (H/8)*(W/tw)/nsplit pixel tiles of 8 x tw pixels each,
 * tw = mphsir_gdfn_fused_tile_width(D) (16 up to D = 128, 8 above).                                    */
typedef struct mphsir_gdfn_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* X; int64_t ldx;
    const float* ln_w; const float* ln_b;
    const void* Win; const float* w9; int64_t ldw;
    const void* Wout;
    void* Y; int64_t ldy;
    int32_t B, H, W, D, HP, nsplit;
    void* T; int64_t ldt;     /* optional (training): t = project_in(LayerNorm(X)) [B*H*W][ldt >= 2*HP] in the storage type, laid out like the
                                 output of mphsir_gemm_tok with Win -- what the backward of the depthwise conv / the gate needs */
} mphsir_gdfn_args;
int mphsir_gdfn_fused(const mphsir_gdfn_args* a, int dtype, void* stream);
int mphsir_gdfn_fused_fits(int32_t D, int32_t HP, int32_t H, int32_t W, int dtype);
int mphsir_gdfn_fused_tile_width(int32_t D);

/* ---- fused LayerNorm + gated MLP + residual --------------------------------------------------
 * Y = X + keep[b] * ( fc2( value * gelu_erf(gate) ) + b2 ),  [value|gate] = fc1(LayerNorm(X)) + b1
 * Replaces PGSSTB's `x + drop_path(self.mlp(self.norm2(x)))` (net/MP_HSIR.py:719; GatedMlp :66-82:
 * FIRST half of fc1 is the value, SECOND half goes through GELU; nn.LayerNorm :619).
 * W1 is [2*HP][C]: rows 0..hid-1 = fc1.weight[0:hid] (value), rows HP..HP+hid-1 = fc1.weight[hid:2hid]
 * (gate), everything else zero; b1 [2*HP] likewise; W2 is [C][HP] = fc2.weight zero-padded along K.
 * HP = hid rounded up to a multiple of 32.  C in {32,64,96,128,192,256,384}; M % 64 == 0.
 * Y may alias X.                                                                                 */
typedef struct mphsir_mlp_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* X; int64_t ldx;
    const float* ln_w; const float* ln_b;
    const void* W1; const float* b1;
    const void* W2; const float* b2;
    const float* keep; int64_t rows_per_batch;   /* DropPath factor per sample, or NULL */
    void* Y; int64_t ldy;
    int64_t M; int32_t C, HP;
    int32_t tiles_per_wave;                      /* tuning: 0 = auto; four waves with 1 or 2 token tiles (of 16) per wave; 3 / 4 =
                                                    eight waves with 1 / 2 tiles (16-bit types) */
    int32_t hsplit;                              /* > 1: small launches (the latent level): the hidden dimension is dealt to hsplit
                                                    workgroups per token tile, the weights are still read once in total; needs */
    float* ypart;                                /* ... a workspace [hsplit][M][C] fp32 for the partial fc2 products (summed in order) */
    const void* R; int64_t ldr;                  /* optional second residual [M][ldr]: Y = (X + keep * mlp(LN(X))) + R -- the skip of a
                                                    whole BaseBlock (net/MP_HSIR.py:727-761) added by its last block's kernel */
    /* optional FUSED BRANCH SUM (PV != NULL; mphsir_gated_mlp_fwd_fuses): the row the LayerNorm / the residual read is not X but
     *   y = X + pkeep[b] * (PSA * pgate[window(m)] + PV Mb[b]^T)      (PGSSTB.forward net/MP_HSIR.py:715-718 -- exactly mphsir_gemm_tok
     * epi 2 with R = X, element for element), formed in LDS; Yb (optional [M][ldyb]: training keeps y for the backward) receives it.
     * PV [M][ldpv]; PM [B][C][C] per-sample matrices (stride pm_batch_stride elements); PSA [M][ldpsa]; pgate [B*nW][C] fp32;
     * pkeep [B] fp32 or NULL; image geometry H, Wimg (multiples of 8, H*Wimg % 128 == 0, M = B*H*Wimg), shift 0 | 4.        */
    const void* PV; int64_t ldpv; const void* PM; int64_t pm_batch_stride;
    const void* PSA; int64_t ldpsa; const float* pgate; const float* pkeep;
    void* Yb; int64_t ldyb;
    int32_t H, Wimg, shift;
} mphsir_mlp_args;
int mphsir_gated_mlp_fwd(const mphsir_mlp_args* a, int dtype, void* stream);
int mphsir_gated_mlp_fwd_fuses(int32_t C, int64_t M, int dtype);     /* 1 where the fused branch sum exists: 16-bit types, C <= 128, M % 128 == 0 */

/* ---- backward (data) of the fused LayerNorm + gated MLP + residual ---------------------------------
 * Given dY = dL/dY of mphsir_gated_mlp_fwd and DM = keep[b]*dY (= dY when DropPath is off), computes
 * dX = dY + LN_backward(...) and writes the three token matrices the parameter gradients are plain
 * token-reduction GEMMs / column sums of:  XN = LN(X) [M][C],  H = value*gelu(gate) [M][HP],
 * DPRE = [dvalue | dgate] [M][2*HP]  (dW2 = DM^T H, dW1 = DPRE^T XN, db2 = colsum DM, db1 = colsum DPRE),
 * plus per-workgroup partial sums part[M/64][2][C] of d(ln weight) and d(ln bias).
 * W1/b1 as for the forward; W1T = W1 transposed [C][2*HP]; W2T = fc2.weight^T zero-padded [HP][C].
 * fp32 supports C <= 256.  Autograd of train.py:58-67 for net/MP_HSIR.py:719.                       */
typedef struct mphsir_mlp_bwd_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* X; const void* dY; const void* DM;
    const float* ln_w; const float* ln_b;
    const void* W1; const float* b1; const void* W1T; const void* W2T;
    void* dX; void* XN; void* H; void* DPRE; float* part;
    int64_t M; int32_t C, HP;
    int32_t variant;                  /* 0 = library's choice; 1..4 pin a kernel form (tests / tuning), see gated_mlp_bwd.hip */
    const float* keep;                /* optional DropPath factors [M / rows_per_batch]: DM is then an OUTPUT, written */
    int64_t rows_per_batch;           /* here as keep[b] * dY (rounded to the compute dtype) instead of by the caller   */
    int32_t hsplit;                   /* > 1: small launches (the latent level): the hidden dimension dealt to hsplit workgroups per
                                         token tile (weights still read once in total) + a second launch that sums the partial d_xn in
                                         order and finishes; variant 0 or 2; needs */
    float* dxn_part;                  /* ... a workspace [hsplit][M][C] fp32 */
} mphsir_mlp_bwd_args;
int mphsir_gated_mlp_bwd(const mphsir_mlp_bwd_args* a, int dtype, void* stream);

/* ---- the parameter gradients of the gated MLP without token-sized intermediates (16-bit types) ------------------------
 * With H == NULL and DPRE == NULL mphsir_gated_mlp_bwd writes neither of them (XN and, with keep, DM are still written), and
 * this entry point produces dW1, db1, dW2, db2 from XN = LN(X) and DM = keep*dY alone: a workgroup owns a 32*chunks_per_wg-wide
 * slab of the hidden dimension and a range of 64-token tiles, RECOMPUTES value / gate / dh for its slab on the matrix cores
 * and keeps its slice of the gradients as fp32 MFMA accumulators over the whole range (gated_mlp_wgrad.hip).  Outputs are
 * per-range partial sums -- dW1p [ranges][2*HP][C] (value rows j, gate rows HP + j), dW2p [ranges][C][HP],
 * db1p [ranges][2*HP], db2p [ranges][C] -- for mphsir_reduce_parts (fixed order: deterministic).  ranges: a multiple of 8
 * (the slabs of a range share an XCD's L2), every range covers ceil(M/64 / ranges) tiles, trailing ranges may be empty
 * (zeros).  W1 / b1 / W2T as for mphsir_gated_mlp_bwd.  mphsir_gated_mlp_wgrad_fits: which (C, chunks_per_wg, dtype) exist.
 * Autograd of train.py:58-67 for net/MP_HSIR.py:66-82 (fc1, fc2 of GatedMlp).                                            */
typedef struct mphsir_mlp_wgrad_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* XN; const void* DM;
    const void* W1; const float* b1; const void* W2T;
    float* dW1p; float* dW2p; float* db1p; float* db2p;
    int64_t M; int32_t C, HP;
    int32_t ranges;
    int32_t chunks_per_wg;            /* 1: 256-thread workgroups, two per CU; 2: 512 threads, half the tile reads from L2 per FLOP */
} mphsir_mlp_wgrad_args;
int mphsir_gated_mlp_wgrad(const mphsir_mlp_wgrad_args* a, int dtype, void* stream);
int mphsir_gated_mlp_wgrad_fits(int32_t C, int32_t chunks_per_wg, int dtype);

/* The depthwise / gate middle of the GDFN backward in one launch (mphsir_dwconv_gate_bwd + mphsir_dwconv3x3_bwd): from T = project_in(LN(x))
 * [B*H*W][2*HP], the taps w9 fp32 [9][ldw >= 2*HP] and dU [B*H*W][HP] (= dY W_out) it recomputes [x1 | x2] = dwconv3x3(T) on every tile's
 * one-pixel halo, forms [d x1 | d x2] there (never written to HBM), and emits U = gelu(x1) x2 [B*H*W][HP] (for d project_out), dT =
 * dwconv3x3^T([d x1 | d x2]) [B*H*W][2*HP] and the tap-gradient partials [nblk][9][2*HP] (one per tile range; nblk x HP/16 workgroups).
 * round_mid != 0 (tests): [d x1 | d x2] rounded to the storage type as the two-launch path stores it -> U and dT bitwise that path's.
 * 16-bit dtypes, H % 8 == 0, W % 16 == 0, HP % 16 == 0.  Reference: FFN / FeedForward.forward net/MP_HSIR.py:259-265 == :385-391. */
int mphsir_gdfn_dw_bwd(const void* T, const float* w9, int64_t ldw, const void* dU, void* U, void* dT, float* partial, int32_t nblk,
                       int32_t B, int32_t H, int32_t W, int32_t HP, int32_t round_mid, int dtype, void* stream);
int mphsir_gdfn_dw_bwd_fits(int32_t H, int32_t W, int32_t HP, int dtype);

/* ---- backward of the window-attention side of a PGSSTB block --------------------------------------
 * mphsir_combine_bwd: backward of mphsir_gemm_tok epi 2 (y = R + keep*(SA*gate[win] + acc), net/MP_HSIR.py
 *   :715-718,:153): dOut = keep*dY (written only if keep != NULL), dSA = dOut*gate[win],
 *   dgate[win][c] = sum over the window's 64 tokens of dOut*SA.  All cubes (B,H,W,C), image order.
 * mphsir_win_attn_bwd: backward of the attention core of mphsir_win_attn_fwd (Spatial_Attention.forward
 *   :193-218) for d_sa = dSA + dmu[win]/64 (dmu = gradient w.r.t. the window mean that feeds the gate).
 *   Outputs: dQKV [B*nW*64][3C] and XNw = LN(x) [B*nW*64][C], both in window-token order (row = window*64 +
 *   token), dSAt = the total d_sa (window-token order, like Oattn), drpb [B*nW][225][heads] partials of the bias-table gradient.
 *   The caller finishes with: d_xn = dQKV Wqkv (plain GEMM), dWqkv = dQKV^T XNw, dbqkv = colsum dQKV (mphsir_gemm_tn),
 *   dWproj = dSAt^T Oattn, dbproj = colsum dSAt.  WprojT = proj.weight^T [C][C].
 *   mphsir_win_attn_bwd_fits(C, heads, dtype): 1 for every instantiated (width, head_dim) in both dtypes (the per-head
 *   tiles do not grow with C: the d_sa / LN(x) operands are re-read from the rows the workgroup has just written).
 * mphsir_ln_bwd_win: dX = dRes + LayerNorm_backward(dXNw) where dXNw [B*nW*64][C] is in window-token order
 *   (norm1 :667 + roll/partition :672-678 in reverse); part [B*nW][2][C] = partials of d(weight), d(bias).
 *   linear != 0: rows are plain token order (the pre-norms of TransformerBlock / CrossTransformer, :282,:286,:476,:477);
 *   XN (optional, needs ln_b): also writes LN(X) in the row order of dXNw.                                    */
int mphsir_combine_bwd(const void* dY, const void* SA, const float* gate, const float* keep, void* dOut, void* dSA,
                       float* dgate, int32_t B, int32_t H, int32_t W, int32_t C, int32_t shift, int dtype, void* stream);
typedef struct mphsir_win_attn_bwd_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* X; const void* dSA; const float* dmu;
    const float* ln_w; const float* ln_b;
    const void* Wqkv; const float* bqkv; const float* rpb; const void* WprojT;
    void* dQKV; void* XNw; void* dSAt; float* drpb;
    int32_t B, H, W, C, heads, shift;
    int32_t head_split;       /* tuning: 0 = auto; n = the heads of a window are dealt to n workgroups (n divides heads) -- small launches
                                 (the latent level: 128 windows for 256 CUs) otherwise leave most of the chip idle; results identical */
} mphsir_win_attn_bwd_args;
int mphsir_win_attn_bwd(const mphsir_win_attn_bwd_args* a, int dtype, void* stream);
int mphsir_win_attn_bwd_fits(int32_t C, int32_t heads, int dtype);
int mphsir_ln_bwd_win(const void* X, const void* dXNw, const void* dRes, const float* ln_w, void* dX, float* part,
                      int32_t B, int32_t H, int32_t W, int32_t C, int32_t shift, const float* ln_b, void* XN, int32_t linear,
                      int dtype, void* stream);

/* The same with the d_xn GEMM inside: d_xn = dQKV WqkvT^T is formed per window on the matrix cores (dQKV [B*nW*64][3C] in window-token
 * order as mphsir_win_attn_bwd writes it, WqkvT = the [C][3C] weight mphsir_gemm_tok would take for that data gradient) and fed to the
 * LayerNorm backward in LDS: replaces mphsir_gemm_tok + mphsir_ln_bwd_win (autograd of net/MP_HSIR.py:667 norm1 and :193-196 qkv);
 * d_xn never reaches HBM.  dX = dRes + dRes2 + LN_backward(d_xn) in image order (dRes, dRes2 optional: dRes2 is the gradient arriving over
 * the skip of the enclosing BaseBlock, net/MP_HSIR.py:760, when this block is its first); part [B*nW][2][C] as mphsir_ln_bwd_win.  16-bit dtypes. */
int mphsir_ln_bwd_win_dxn(const void* X, const void* dQKV, const void* WqkvT, const void* dRes, const void* dRes2, const float* ln_w, void* dX,
                          float* part, int32_t B, int32_t H, int32_t W, int32_t C, int32_t shift, int dtype, void* stream);
int mphsir_ln_bwd_win_dxn_fits(int32_t C, int dtype);
/* ... and in plain token order for the prompt modules' pre-norm 1x1 convs (autograd of `self.ffn(self.norm2(x))` / `self.attn(self.norm1(x))`,
 * net/MP_HSIR.py:286,476-477 with :256,:303): d_xn = dY WT^T with dY [M][K] (the gradient of the 1x1 conv's output, K = its output
 * channels, K % 32 == 0), WT [C][K] = the conv weight transposed; dX = dRes + LN_backward(d_xn) (dRes optional); XN (optional, needs
 * ln_b) = LN(X), the operand of the conv's weight gradient; part [M/64][2][C].  M % 64 == 0; (C, dtype) as mphsir_ln_bwd_win_dxn_fits.
 * x_f32 != 0: X and dX are fp32 rows while dY / WT / dRes / XN stay `dtype` (TVSP's norm11 on the fp32 text map, net/MP_HSIR.py:282 with
 * :575-577 -- the reference's autocast keeps that LayerNorm in fp32 too); C <= 192.                                                  */
int mphsir_ln_bwd_tok_dxn(const void* X, const void* dY, const void* WT, const void* dRes, const float* ln_w, const float* ln_b, void* dX,
                          void* XN, float* part, int64_t M, int32_t C, int32_t K, int32_t x_f32, int dtype, void* stream);

/* ---- stand-alone LayerNorm over channels (SURVEY 8b `layernorm_nhwc`; net/MP_HSIR.py:341-370) ---------------------------
 * Y[m][:] = LN(X[m][:]) * ln_w + ln_b, biased variance, eps 1e-5, statistics in fp32.  X and Y may have different element
 * types (x_dtype / y_dtype): TVSP's norm11 (:282) reads the fp32 rank-one text map and writes the compute dtype.
 * Xcopy (optional, [M][C] in y_dtype): X cast to the output type by the same pass -- the residual operand of that block (:282).  */
int mphsir_layernorm_tok(const void* X, int x_dtype, const float* ln_w, const float* ln_b, void* Y, void* Xcopy, int y_dtype,
                         int64_t M, int32_t C, void* stream);

/* ---- backward of the two small per-sample / per-window stages -----------------------------------------
 * mphsir_spectral_fold_bwd: backward of mphsir_spectral_fold.  dM [B][C][C] fp32 = gradient w.r.t. M_b.  Outputs:
 *   W2 [B][2C][2C] (compute dtype) such that [dq | dk] = [q | k] W2_b^T is one mphsir_gemm_tok with a per-sample
 *   weight (rows c: [diag(2 d sum q^2) | dG blocks], rows C+c: [dG^T blocks | diag(2 d sum k^2)]);
 *   dWo [B][C][C] fp32 per-sample parts of d project_out.weight (sum over B); dtemp [B][heads] parts of d temperature.
 * mphsir_pg_gate_bwd: backward of the local spectral-prompt gate for every window: dmu [nW][C] and the factor rows
 *   L [nW][KL] = [dgate | d o2 | d kv | d q | w | d logit | d d | 0], R [nW][KR] = [o2 | o | 1 | d | s | d s | mu | 0]
 *   (KL >= C+5r+256, KR >= 5r+1+C, multiples of 4): L^T R over the windows (mphsir_gemm_tn, fp32) holds every
 *   parameter gradient of PG_Spectral_Attention (net/MP_HSIR.py:122-129) as a sub-block.                        */
typedef struct mphsir_fold_bwd_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const float* Gpart; const float* Spart; const float* temperature; const float* Wo; const float* dM;
    void* W2; float* dWo; float* dtemp;
    int32_t B, C, heads, nsplit;
    int32_t dM_nsplit;          /* <= 1: dM is [B][C][C]; > 1: dM is [B][dM_nsplit][C][C], the split partials of mphsir_gemm_tn, summed here in split order */
    /* N > 0 (16-bit dtypes, head width 32 / 48 / 64; meant for small images: N <= 1024 tokens per sample): dM is not read (may be
     * NULL) but FORMED in the kernel, dM_b = d_out_b^T v_b over the sample's N tokens -- DO [B*N][lddo >= C] = d_out, V [B*N][ldv >= C]
     * = v of the forward (16-byte aligned rows): the token-reduction GEMM in front of this launch disappears */
    const void* DO; int64_t lddo; const void* V; int64_t ldv; int32_t N;
    const float* dm_scale;      /* optional [B]: dM_b (given or formed) is multiplied by dm_scale[b] -- the DropPath factor, when d_out is handed over
                                   as the block's incoming gradient without it */
    int32_t w2_blocks;          /* != 0: only the entries mphsir_spectral_dqkv_bwd reads are written -- per head the columns [h hd, (h+1) hd) and
                                   [C + h hd, C + (h+1) hd) of its q and k rows (W2_b is zero elsewhere; 0 writes the zeros too, for the token GEMM
                                   [dq | dk] = [q | k] W2_b^T of the unfused path): 2 hd / 2C of the stores */
} mphsir_fold_bwd_args;
int mphsir_spectral_fold_bwd(const mphsir_fold_bwd_args* a, int dtype, void* stream);
typedef struct mphsir_pg_bwd_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const float* mu; const float* dgate;
    const float* Wprompt; const float* prompt_param; const float* Wq; const float* Wkv; const float* Wdown;
    const float* Wpproj; const float* bpproj; const float* Wup;
    float* dmu; void* L; void* R;     /* L, R: fp32, or bf16 when lr_bf16 != 0 (then KL, KR multiples of 8) */
    int32_t nW, C, r, KL, KR;
    int32_t lr_bf16;
} mphsir_pg_bwd_args;
int mphsir_pg_gate_bwd(const mphsir_pg_bwd_args* a, void* stream);

/* ---- backward of the channel attention between the fold and the 1x1 conv, one launch ------------------------------------------
 * autograd of Spectral_Attention.forward net/MP_HSIR.py:96-113 (== Attention :301-322) through train.py:58-67, in the folded form
 * (SURVEY Appendix A):   dv = d_out M_b,   [dq | dk] = [q | k] W2_b^T,   dt = depthwise3x3^T([dq | dk | dv]),
 *                        d taps[c][tap] = sum_p t[c][p + tap] [dq | dk | dv][c][p].
 * Replaces mphsir_gemm_tok (x2, per-sample weights) + mphsir_dwconv3x3_bwd: [dq | dk | dv] (3C values per token) never reaches HBM --
 * a workgroup produces the halo tile of its channel slab on the matrix cores (fp32, in LDS) and runs the depthwise backward on it.
 * QK  [B*H*W][ldqk >= 2C]  q | k after the depthwise conv, as mphsir_qkv_dwconv_gram / mphsir_dwconv_gram keep them;
 * DO  [B*H*W][lddo >= C]   d_out;      T  [B*H*W][ldt >= 3C]  t = qkv(x) (the depthwise conv's input);
 * W2  [B][2C][2C]          per-sample matrix of mphsir_spectral_fold_bwd;      MbT [B][C][C]  M_b^T of mphsir_spectral_fold;
 * w9  fp32 [9][ldw >= 3C]  taps of q | k | v;      dT [B*H*W][lddt >= 3C];
 * part fp32 [nblk][9][3C]  tap-gradient partials, one per tile range (ordered sum by mphsir_reduce_parts): nblk ranges x
 *                          mphsir_spectral_dqkv_bwd_slabs(C, heads) channel slabs = the workgroups of the launch.
 * round_dall != 0 (tests): [dq | dk | dv] is rounded to the storage type before the depthwise pass -- dT is then bitwise what the
 * three-launch path produces (head widths 32 / 64).  16-bit dtypes; H % 8 == 0, W % 16 == 0; (C, C/heads) as mphsir_..._fits says.  */
typedef struct mphsir_spectral_bwd_args {
    uint32_t struct_size;       /* = sizeof(this struct), set by the caller: any other value is rejected with MPHSIR_EINVAL */
    const void* QK; int64_t ldqk;
    const void* DO; int64_t lddo;
    const void* T; int64_t ldt;
    const void* W2; const void* MbT;
    const float* w9; int64_t ldw;
    void* dT; int64_t lddt;
    float* part;
    int32_t B, H, W, C, heads, nblk, round_dall;
    const float* vscale;        /* optional [B]: dv_b = vscale[b] * DO_b M_b (DO handed over without its DropPath factor) */
} mphsir_spectral_bwd_args;
int mphsir_spectral_dqkv_bwd(const mphsir_spectral_bwd_args* a, int dtype, void* stream);
int mphsir_spectral_dqkv_bwd_fits(int32_t C, int32_t heads, int32_t H, int32_t W, int dtype);
int mphsir_spectral_dqkv_bwd_slabs(int32_t C, int32_t heads);

/* ---- dense 3x3 convolution (implicit GEMM) -----------------------------------------------------------
 * Y[p][n] = sum_tap sum_ci X[p+tap][ci] * W[n][tap*Cin + ci]   stride 1, zero padding, no bias, channels-last.
 * Replaces OverlapPatchEmbed.proj (net/MP_HSIR.py:458), Downsample/Upsample convs (:436,:446), TVSP.conv_last
 * (:566) and `output` (:807).  Cin % 32 == 0, N % 16 == 0 (the caller zero-pads 31 -> 32), B*H*W % 64 == 0.
 * Input gradient = the same call with flipped/transposed weights; weight gradient = mphsir_gemm_tn(dY,
 * Col) with Col [B*H*W][9*Cin] written by mphsir_im2col3x3.                                           */
int mphsir_conv3x3_tok(const void* X, int64_t ldx, const void* W, void* Y, int64_t ldy, int32_t B, int32_t H, int32_t Wd,
                       int32_t Cin, int32_t N, int dtype, void* stream);
int mphsir_im2col3x3(const void* X, int64_t ldx, void* Col, int32_t B, int32_t H, int32_t Wd, int32_t Cin, int dtype, void* stream);
/* Weight gradient of the same conv WITHOUT the im2col matrix: partial[split][co][tap*Cin + ci] = sum over the pixels of the split of
 * dY[p][co] * X[p + tap][ci] (X gathered inside the token-reduction GEMM, zeros outside the image); 16-bit types.  The caller sums
 * the nsplit partials (mphsir_reduce_parts).  Replaces mphsir_im2col3x3 + mphsir_gemm_tn on the 16-bit path. */
int mphsir_conv3x3_wgrad(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* Cpart, int32_t B, int32_t H, int32_t W,
                         int32_t Cout, int32_t Cin, int32_t nsplit, int32_t form /* 1 | 2, as mphsir_gemm_tn's tile128 */,
                         int dtype, void* stream);

/* ---- token-reduction GEMM (weight gradients) --------------------------------------------------------
 * Cpart[b][s][n1][n2] = sum over the s-th token range of A[b][m][n1] * B[b][m][n2]  (fp32 partials;
 * the caller sums the nsplit partials in order).  A: [batch][M][lda], B: [batch][M][ldb] token-major
 * views (batch strides in elements).  This is dW = dY^T X of every Linear / 1x1 conv on the path and the
 * per-sample dM = d_out^T v of the folded channel attention (autograd of net/MP_HSIR.py, train.py:58-67).
 * colsum_part (optional, [batch][nsplit][N1]): partial column sums of A = the matching bias gradient.
 * tile128 != 0 selects the large-tile variant: fp32 128x128 tiles; 16-bit types: 1 = the transposed-LDS-read kernel
 * (ds_read_b64_tr_b16, no transposing stores, two LDS stages, register-staged loads, 256-thread workgroups, ~2 per CU)
 * with a 64- or 128-wide tile per operand; 2 = its ring form: one 512-thread workgroup per CU (size nsplit for ~256
 * workgroups), token rows by LDS-DMA into a ring of 32-token slots, two wave groups whose tiles are combined in LDS:
 * half the partial tiles per launch, written as whole 16-byte row chunks.                                          */
int mphsir_gemm_tn(const void* A, int64_t lda, int64_t a_batch_stride, const void* B, int64_t ldb, int64_t b_batch_stride,
                   float* Cpart, float* colsum_part, int64_t M, int32_t N1, int32_t N2, int32_t nsplit, int32_t batch,
                   int32_t tile128, int dtype, void* stream);
/* Up to MPHSIR_TN_GROUP_MAX independent problems of the bf16 large-tile form in ONE launch (same outputs as n calls
 * of mphsir_gemm_tn with batch = 1): the weight-gradient GEMMs of one backward function issued together.         */
#define MPHSIR_TN_GROUP_MAX 8
typedef struct mphsir_gemm_tn_problem {
    const void* A; int64_t lda; const void* B; int64_t ldb;
    float* Cpart; float* colsum_part;
    int64_t M; int32_t N1, N2, nsplit, pad_;
} mphsir_gemm_tn_problem;
int mphsir_gemm_tn_group(const mphsir_gemm_tn_problem* probs, int32_t n, int32_t form /* 1 | 2 */, int dtype, void* stream);

/* ---- plain depthwise 3x3 (backward building blocks) ----------------------------------------------
 * mphsir_dwconv3x3: Y[p][c] = sum_taps X[p+tap][c] * w9[tap][c] (zero padding); flip=1 uses the spatially
 * flipped taps = gradient w.r.t. the input of the same depthwise conv applied to dY.
 * mphsir_dwconv3x3_wgrad: partial[blk][tap][c] = sum over the pixels of block blk of X[p+tap][c]*dY[p][c];
 * the caller sums the nblk partials (fixed order -> deterministic).  w9 / partial are fp32, tap-major.
 * Gradients of every nn.Conv2d(groups=channels) on the path (net/MP_HSIR.py:92,227,230,257,382).        */
int mphsir_dwconv3x3(const void* X, int64_t ldx, const float* w9, int64_t ldw, void* Y, int64_t ldy,
                     int32_t B, int32_t H, int32_t W, int32_t C, int32_t flip, int dtype, void* stream);
int mphsir_dwconv3x3_wgrad(const void* X, int64_t ldx, const void* dY, int64_t lddy, float* partial, int32_t nblk,
                           int32_t B, int32_t H, int32_t W, int32_t C, int dtype, void* stream);
/* 1 if mphsir_dwconv3x3_wgrad takes its LDS-tile form for this shape (16-bit, H % 8 == 0, W % 16 == 0, C % 32 == 0): workgroup
 * (b, slab) of the grid (nblk, ceil(C/96)) then walks the 8x16-pixel tiles b, b + nblk, ... -- size nblk for about one workgroup
 * per CU instead of one partial block per ~128 pixels. */
int mphsir_dwconv3x3_wgrad_tiled(int32_t H, int32_t W, int32_t C, int dtype);
/* Both gradients of one depthwise 3x3 in one launch, for the shapes mphsir_dwconv3x3_wgrad_tiled accepts:
 * dX = mphsir_dwconv3x3(dY, flip=1) and partial = mphsir_dwconv3x3_wgrad(X, dY) (same layout, same nblk rule) with dY read once.
 * X = the conv's input, dY = the gradient of its output; dX must not alias dY. */
int mphsir_dwconv3x3_bwd(const void* X, int64_t ldx, const void* dY, int64_t lddy, const float* w9, int64_t ldw, void* dX, int64_t lddx,
                         float* partial, int32_t nblk, int32_t B, int32_t H, int32_t W, int32_t C, int dtype, void* stream);

/* backward of the GDFN gate u = gelu_erf(T[:, :HP]) * T[:, HP:] (FFN/FeedForward.forward :263, :389):
 * given dU [M][HP] writes U (recomputed, for d project_out) and dT [M][2*HP].                          */
int mphsir_gdfn_gate_bwd(const void* T, const void* dU, void* U, void* dT, int64_t M, int32_t HP, int dtype, void* stream);
/* The same backward with the depthwise conv recomputed inside (T here is the conv's INPUT t = project_in(LN(x)), [B*H*W][2*HP] contiguous,
 * w9 fp32 [9][ldw >= 2*HP]): u and d(conv output) leave in one pass, the conv output never reaches HBM.  Shapes
 * mphsir_dwconv_gate_bwd_fits accepts (16-bit types, H % 8 == 0, W % 16 == 0); U [B*H*W][HP], dT [B*H*W][2*HP] contiguous. */
int mphsir_dwconv_gate_bwd(const void* T, const float* w9, int64_t ldw, const void* dU, void* U, void* dT, int32_t B, int32_t H, int32_t W,
                           int32_t HP, int dtype, void* stream);
int mphsir_dwconv_gate_bwd_fits(int32_t H, int32_t W, int32_t HP, int dtype);

/* ---- sizes of the caller-allocated partial / workspace buffers (bytes) ----------------------------------------------------------
 * The library never allocates: split partials and factor rows are outputs the caller provides.  These helpers return
 * the byte sizes the entry points expect (what mp-hsir_amd/ops.py allocates), so that a binding in another language
 * does not have to re-derive them from the layouts documented above.                                                     */
int64_t mphsir_gemm_tn_workspace_bytes(int32_t N1, int32_t N2, int32_t nsplit, int32_t batch, int32_t with_colsum);
int64_t mphsir_dwconv_gram_workspace_bytes(int32_t B, int32_t nsplit, int32_t C, int32_t heads);   /* Gpart + Spart */
int64_t mphsir_pg_gate_bwd_workspace_bytes(int32_t nW, int32_t C, int32_t r, int dtype, int32_t* KL, int32_t* KR); /* L + R; widths out */
int64_t mphsir_win_attn_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads, int dtype); /* dQKV+XNw+dSAt+drpb */

/* ---- the resamplers of TVSP.forward (net/MP_HSIR.py:572-583), channels-last ----------------------------------------------
 * mphsir_tvsp_text_map: text [B][ps][ps][D] fp32 = L[b][d] * clip[floor(i*B/ps)][floor(j*512/ps)] -- the reference's broadcast
 *   of (B,D,1,1) x (B,512) followed by F.interpolate(nearest) to (ps,ps) (:575-577; the batch axis lands on image rows).
 *   L [B][D] fp32 (task-weighted mean of text_prompt_learnable), clip [B][512] fp32.
 * mphsir_tvsp_text_map_bwd: part [B][ps][D] fp32, part[b][i][d] = sum_j dtext[b,i,j,d] * clip_map[i][j]; dL = sum over i.
 * mphsir_resize_bilinear: F.interpolate(mode="bilinear", align_corners=False) (:580) X (B,h,w,C) -> Y (B,H,W,C);
 *   backward != 0: X = dY (B,H,W,C) -> Y = dX (B,h,w,C), a gather (deterministic).  C a multiple of 16 bytes of elements.   */
int mphsir_tvsp_text_map(const float* L, const float* clip, float* text, int32_t B, int32_t ps, int32_t D, void* stream);
int mphsir_tvsp_text_map_bwd(const float* dtext, const float* clip, float* part, int32_t B, int32_t ps, int32_t D, void* stream);
int mphsir_resize_bilinear(const void* X, void* Y, int32_t B, int32_t h, int32_t w, int32_t H, int32_t W, int32_t C, int32_t backward,
                           int dtype, void* stream);

/* ---- the two ends of MP_HSIR_Net.forward (net/MP_HSIR.py:822-844) and the task-prompt algebra, one launch each -------------
 * mphsir_nchw_to_cl:      Y [B][HW][Cp] (dtype) = X [B][C][HW] fp32 with channels C..Cp-1 zero -- `inp_img` as the patch embedding's
 *   channels-last, 32-padded input (:824); also the backward of the output head (d conv_out from d restored).  Cp <= 252.
 * mphsir_cl_to_nchw_add:  O [B][C][HW] fp32 = (float) Y [B][HW][ldy] (dtype, first C channels) + R [B][C][HW] fp32 (R may be NULL)
 *   -- `self.output(...) + inp_img` (:842-843).
 * mphsir_task_weights:    w [B][T] fp32 = the mean of the one-hot rows of ids [B][n] (int64 task ids; Text_Prompt.forward :519-523).
 * mphsir_mix_rows:        O [I][D] = scale * A Bm, A [I][J] (transA: stored [J][I]), Bm [J][D], fp32 -- the task-weighted means
 *   (w.unsqueeze(-1) * table).mean(1) of :527 / TVSP :574 (scale 1/T) and their gradient d table = w^T dO / T.                */
int mphsir_nchw_to_cl(const float* X, void* Y, int32_t B, int32_t C, int64_t HW, int32_t Cp, int dtype, void* stream);
int mphsir_cl_to_nchw_add(const void* Y, int64_t ldy, const float* R, float* O, int32_t B, int32_t C, int64_t HW, int dtype, void* stream);
int mphsir_task_weights(const int64_t* ids, float* w, int32_t B, int32_t n, int32_t T, void* stream);
int mphsir_mix_rows(const float* A, const float* Bm, float* O, int32_t I, int32_t J, int32_t D, float scale, int32_t transA, void* stream);

/* ---- fused AdamW over the flat parameter arena ---------------------------------------------------
 * One decoupled-weight-decay Adam step on n contiguous fp32 parameters (n % 4 == 0) with gradient g,
 * moments m, v; g is multiplied by grad_scale first (1/world_size after a sum all-reduce).
 * Semantics = torch.optim.AdamW as the reference configures it (train.py:69); step is 1-based.
 * hyper (optional device pointer to [lr, 1-beta1^step, sqrt(1-beta2^step)]) overrides lr/step so that a captured
 * launch can be replayed with per-step values.                                                      */
int mphsir_flat_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int32_t step, float grad_scale, const float* hyper, void* stream);

/* ---- dynamic loss scaling for the fp16 path (the reference trains precision="16-mixed", train.py:118 = torch GradScaler) ----
 * scaler: device fp32[4] = [loss scale, good steps in a row, found-inf flag of this step, optimizer steps taken].
 * mphsir_grad_check sets scaler[2] = 1 if any of g[0..n) is inf / NaN (run it AFTER the gradient all-reduce: a non-finite
 *   value on one rank reaches every rank through the sum, so all ranks take the same decision without another collective).
 * mphsir_flat_adamw_scaled = mphsir_flat_adamw with g additionally divided by scaler[0]; the whole update is skipped when
 *   scaler[2] != 0; Adam's bias corrections use step = scaler[3] + 1 (skipped steps do not count); hyper (optional) = [lr].
 * mphsir_scaler_update = GradScaler.update(): on overflow scale *= backoff and the streak restarts, else scaler[3] += 1 and
 *   after growth_interval good steps scale *= growth; clears the flag.  All three are plain launches: capturable.       */
int mphsir_grad_check(const float* g, int64_t n, float* scaler, void* stream);
int mphsir_flat_adamw_scaled(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, float grad_scale, const float* hyper, const float* scaler, void* stream);
int mphsir_scaler_update(float* scaler, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream);

/* ---- flat-arena utilities -------------------------------------------------------------------------
 * reduce_parts: dst[b][i] = sum_{s < nsplit} src[b*src_batch_stride + s*stride + i], i < n, b < nbatch, summed in a
 * fixed order (a function of nsplit and n only: deterministic), for up to MPHSIR_REDUCE_MAX_SEGS independent segments in ONE launch.  Replaces the
 * at::sum the reference's autograd runs per parameter gradient (train.py:58-67) for every split-M partial buffer
 * the backward kernels of this library write.  All pointers fp32 device memory.                        */
#define MPHSIR_REDUCE_MAX_SEGS 32
typedef struct mphsir_reduce_seg {
    const float* src; float* dst;
    int64_t n, stride, src_batch_stride, dst_batch_stride;
    int32_t nsplit, nbatch;
    /* optional 2-D form (rows > 1): the segment is a rows x n sub-block of a wider partial matrix and lands in dst as a
     * (possibly transposed) dense block: element (r, i) is read at src + r*src_ld + i (+ s*stride) and written to
     * dst + r*dst_ld + i*dst_col_stride.  Un-padding (hidden 170 -> 192), sub-block extraction and the [9][C] -> [C][9]
     * tap transposes of the parameter gradients happen here instead of in separate copy / cat launches.
     * rows <= 1: plain vector of n elements (src_ld, dst_ld ignored, dst_col_stride 0 or 1).                         */
    int32_t rows, dst_col_stride;
    int64_t src_ld, dst_ld;
} mphsir_reduce_seg;
int mphsir_reduce_parts(const mphsir_reduce_seg* segs, int32_t nseg, void* stream);

/* pack_gather: dst[i] = index[i] >= 0 ? (T)arena[index[i]] : 0 for i < n (n % (16/sizeof(T)) == 0).  One launch
 * rebuilds every kernel-layout weight tensor (cast, zero padding, transposes, value|gate splits) from the flat fp32
 * parameter arena after an optimizer step; `index` is static (built once from the packers on the host side).   */
int mphsir_pack_gather(const float* arena, const int32_t* index, void* dst, int64_t n, int dtype, void* stream);

/* multi_copy: n fp32 tensors copied in ONE launch -- the gradient hand-over (autograd's freshly allocated parameter gradients ->
 * their slots of the flat gradient arena that the all-reduce and mphsir_flat_adamw work on; the reference: DDP's bucket copies,
 * train.py:118).  table_dev: DEVICE memory, nseg rows of four int64 {src pointer, dst pointer, n floats, first block}; a block
 * moves up to 4096 floats, first block = the running sum of ceil(n / 4096); total_blocks = that sum over all rows.            */
int mphsir_multi_copy(const int64_t* table_dev, int32_t nseg, int64_t total_blocks, void* stream);

/* l1_clamp_loss: the reference's training loss and its gradient in one pass (train.py:58-61: clamp(restored, 0, 1) then
 * nn.L1Loss): part[b] = sum over block b's elements of |clamp(y) - clean| / n (the caller sums the nblocks partials in order =
 * the loss), grad (optional) = d loss / d y = sign(clamp(y) - clean) / n where 0 <= y <= 1, else 0.  fp32, contiguous.      */
int mphsir_l1_clamp_loss(const float* y, const float* clean, float* grad, float* part, int64_t n, int32_t nblocks, void* stream);

/* ---- optional per-kernel launch timer (bench.py roofline leg) ----------------------------------
 * When enabled for kernel id `kid`, every launch of that kernel is bracketed by hipEvents on its
 * own stream.  read(): synchronises the recorded events, returns the number of launches and their
 * summed duration in milliseconds, and clears the log.                                            */
#define MPHSIR_K_GEMM_TOK 0
#define MPHSIR_K_WIN_ATTN 1
#define MPHSIR_K_DWCONV_GRAM 2
#define MPHSIR_K_SPECTRAL_FOLD 3
#define MPHSIR_K_GATED_MLP 4
#define MPHSIR_K_DWCONV_GATE 5
#define MPHSIR_K_FLAT_ADAMW 6
#define MPHSIR_K_DWCONV 7
#define MPHSIR_K_DWCONV_WGRAD 8
#define MPHSIR_K_GATED_MLP_BWD 9
#define MPHSIR_K_COMBINE_BWD 10
#define MPHSIR_K_WIN_ATTN_BWD 11
#define MPHSIR_K_LN_BWD_WIN 12
#define MPHSIR_K_GEMM_TN 13
#define MPHSIR_K_GDFN_GATE_BWD 14
#define MPHSIR_K_FOLD_BWD 15
#define MPHSIR_K_PG_GATE_BWD 16
#define MPHSIR_K_CONV3X3 17
#define MPHSIR_K_IM2COL 18
#define MPHSIR_K_REDUCE_PARTS 19
#define MPHSIR_K_PACK_GATHER 20
#define MPHSIR_K_LAYERNORM 21
#define MPHSIR_K_PG_GATE 22
#define MPHSIR_K_RESAMPLE 23
#define MPHSIR_K_QKV_DWCONV_GRAM 24
#define MPHSIR_K_GDFN_FUSED 25
#define MPHSIR_K_DWCONV_BWD 26
#define MPHSIR_K_MULTI_COPY 27
#define MPHSIR_K_L1_LOSS 28
#define MPHSIR_K_GATED_MLP_WGRAD 29
#define MPHSIR_K_SPECTRAL_DQKV_BWD 30
#define MPHSIR_K_LAYOUT 31
#define MPHSIR_K_COUNT 32
int mphsir_prof_enable(int kid);   /* kid < 0 disables */
int mphsir_prof_read(int* launches, float* total_ms);
const char* mphsir_kernel_name(int kid);

#ifdef __cplusplus
}
#endif
#endif
