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
} mphsir_gemm_args;
int mphsir_gemm_tok(const mphsir_gemm_args* a, int dtype, void* stream);

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
#define MPHSIR_K_COUNT 16
int mphsir_prof_enable(int kid);   /* kid < 0 disables */
int mphsir_prof_read(int* launches, float* total_ms);
const char* mphsir_kernel_name(int kid);

#ifdef __cplusplus
}
#endif
#endif
