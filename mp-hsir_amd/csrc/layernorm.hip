// layernorm_tok: stand-alone LayerNorm over channels of a token matrix, with independent input / output element types.
//
// Everywhere else LayerNorm is the prologue of the kernel that consumes it (gemm_tok, win_attn, gated_mlp).  The one
// exception is norm11 of TVSP's CrossTransformer (net/MP_HSIR.py:282): its input, the text map
// text[b,i,j,:] = clip[i,j] * L[b,:] (:575-577), is rank one, so the gradients that flow back through it are small
// residuals of large cancelling sums -- rounding the INPUT to bf16 first changes d(norm11.weight) and d(q.weight) by
// 15 % even in exact arithmetic (measured with the fp64 oracle).  The reference under autocast keeps that LayerNorm in
// fp32 on the fp32 input; so does this kernel: fp32 in, statistics in fp32, output in the compute dtype.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

template <class TI, class TO>
__global__ __launch_bounds__(256) void layernorm_tok_kernel(const TI* __restrict__ X, const float* __restrict__ w,
                                                            const float* __restrict__ b, TO* __restrict__ Y, TO* __restrict__ Xc, long M, int C) {
    // 4 adjacent lanes per token, 64 tokens per workgroup.  C % 16 == 0: each lane walks the row in 4-element vectors (a quad reads 16
    // consecutive elements per step); else element-wise (strided by 4) so any C % 4 == 0 works.  Xc (optional): the input cast to TO.
    const int tid = threadIdx.x, q = tid & 3;
    const long t0 = (long)blockIdx.x * 64 + (tid >> 2);
    const bool live = t0 < M;        // rows past the end redo the last row and store nothing (every lane reaches the shuffles)
    const long t = live ? t0 : M - 1;
    const TI* x = X + t * C;
    TO* y = Y + t * C;
    if (C % 16 == 0) {
        float s = 0.f;
        for (int c = 4 * q; c < C; c += 16) { const f32x4 v = load4<TI>(x + c); s += (v[0] + v[1]) + (v[2] + v[3]); }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float mean = s / (float)C;
        float d2 = 0.f;
        for (int c = 4 * q; c < C; c += 16) {
            const f32x4 v = load4<TI>(x + c);
            for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; d2 += d * d; }
        }
        d2 += __shfl_xor(d2, 1);
        d2 += __shfl_xor(d2, 2);
        const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
        for (int c = 4 * q; c < C; c += 16) {
            const f32x4 v = load4<TI>(x + c), wv = load4<float>(w + c), bv = load4<float>(b + c);
            f32x4 o;
            for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * wv[e] + bv[e];
            if (live) store4<TO>(y + c, o);
            if (live && Xc) store4<TO>(Xc + t * C + c, v);
        }
        return;
    }
    float s = 0.f;
    for (int c = q; c < C; c += 4) s += to_f32(x[c]);
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    const float mean = s / (float)C;
    float d2 = 0.f;
    for (int c = q; c < C; c += 4) { const float d = to_f32(x[c]) - mean; d2 += d * d; }
    d2 += __shfl_xor(d2, 1);
    d2 += __shfl_xor(d2, 2);
    const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
    for (int c = q; c < C; c += 4) {
        if (live) y[c] = from_f32<TO>((to_f32(x[c]) - mean) * rstd * w[c] + b[c]);
        if (live && Xc) Xc[t * C + c] = from_f32<TO>(to_f32(x[c]));
    }
}

}  // namespace mphsir

extern "C" int mphsir_layernorm_tok(const void* X, int x_dtype, const float* ln_w, const float* ln_b, void* Y, void* Xcopy, int y_dtype,
                                    int64_t M, int32_t C, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && ln_w && ln_b && Y, "layernorm_tok: null pointer");
    MPHSIR_REQUIRE(M > 0 && C > 0 && C % 4 == 0, "layernorm_tok: M > 0 and C %% 4 == 0 required");
    MPHSIR_REQUIRE(C % 16 != 0 || (aligned16(X) && aligned16(Y) && aligned16(Xcopy) && aligned16(ln_w) && aligned16(ln_b)), "layernorm_tok: 16-byte alignment required");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(x_dtype) && MPHSIR_DTYPE_OK(y_dtype), "layernorm_tok: dtype unsupported");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((M + 63) / 64)), block(256);
#define MPHSIR_LN_CASE(xi, TI, yo, TO)                                                                          \
    if (x_dtype == xi && y_dtype == yo) {                                                                      \
        MPHSIR_LAUNCH(MPHSIR_K_LAYERNORM, (layernorm_tok_kernel<TI, TO>), grid, block, 0, s,                    \
                      reinterpret_cast<const TI*>(X), ln_w, ln_b, reinterpret_cast<TO*>(Y), reinterpret_cast<TO*>(Xcopy), (long)M, (int)C); \
        return MPHSIR_OK;                                                                                      \
    }
    MPHSIR_LN_CASE(MPHSIR_F32, float, MPHSIR_F32, float)
    MPHSIR_LN_CASE(MPHSIR_F32, float, MPHSIR_BF16, bf16_t)
    MPHSIR_LN_CASE(MPHSIR_BF16, bf16_t, MPHSIR_BF16, bf16_t)
    MPHSIR_LN_CASE(MPHSIR_BF16, bf16_t, MPHSIR_F32, float)
    MPHSIR_LN_CASE(MPHSIR_F32, float, MPHSIR_F16, f16_t)
    MPHSIR_LN_CASE(MPHSIR_F16, f16_t, MPHSIR_F16, f16_t)
    MPHSIR_LN_CASE(MPHSIR_F16, f16_t, MPHSIR_F32, float)
#undef MPHSIR_LN_CASE
    set_error("layernorm_tok: dtype pair (%d -> %d) not instantiated", x_dtype, y_dtype);
    return MPHSIR_EINVAL;
}
