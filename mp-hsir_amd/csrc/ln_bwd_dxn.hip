// ln_bwd_win_dxn: the last two launches of a PGSSTB block's attention backward as one -- the data gradient of the window
// attention's qkv Linear, d_xn = dQKV Wqkv (a token GEMM, K = 3C), and the backward of norm1 with the residual path
// (autograd of net/MP_HSIR.py:667 `self.norm1(x)` and :193-196 `self.qkv(x)` through train.py:58-67).
//
// Until round 6: mphsir_gemm_tok wrote d_xn [M][C] and mphsir_ln_bwd_win read it back next to x and d_res.  At the lower pyramid
// levels both are 12-18 us launches on the critical path for a few MB; at the full-resolution level they are two HBM-bound passes
// (266 MB) of which the d_xn round trip is a quarter.  Here a 256-thread workgroup owns the 64 rows of one window (dQKV is in
// window-token order) and
//   (a) forms d_xn[64][C] on the matrix cores: a wave = 16 token rows x all C outputs; its dQKV fragments come straight from HBM / L2
//       (16 bytes per lane and K chunk), the rows of Wqkv^T go through LDS in K groups (the next group requested before this group's
//       barrier), transposed accumulators (lane = 4 channels x 1 token);
//   (b) parks the fp32 tile [64][C + 1] in LDS -- the tile ln_bwd_win_kernel stages anyway for the column sums -- and runs that kernel's
//       LayerNorm backward on it: 4 lanes per token, x / d_res rows requested at the very top (their latency runs under the GEMM),
//       dx = d_res + rstd (g - mean(g) - xhat mean(g xhat)), then d beta = sum_t d_xn and d gamma = sum_t d_xn xhat per workgroup.
// d_xn never reaches HBM and is not rounded to 16 bits on the way.  16-bit types.
#include <type_traits>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct LnDxnDev {
    const void* X; const void* dQKV; const void* WT; const void* dRes; const float* ln_w;
    void* dX; float* part;
    int B, H, W, shift;
    int K;                       // reduction width of the GEMM (3C for the window attention's qkv; 2 HP / 3 D for the prompt modules), % 32 == 0
    int linear;                  // 1: rows in plain token order (the prompt modules' LayerNorms): row = 64 blockIdx + t, no window gather
    const float* ln_b; void* XN; // optional (linear form): also emit LN(x), the operand of the 1x1 conv's weight gradient
    const void* dRes2;           // optional: a second residual gradient added to dX (the skip of the enclosing BaseBlock, whose backward ends here)
};

// 8 consecutive channels of a token row of X / dX: the 16-bit types hold them packed (one 16-byte vector), fp32 rows (the text map of TVSP's
// norm11, which the reference's autocast keeps in fp32 too) as two
template <class TX> struct Row8 {
    Vec16<TX> v;
    __device__ __forceinline__ void load(const TX* p) { v = load16<TX>(p); }
    __device__ __forceinline__ void store(TX* p) const { store16<TX>(p, v); }
    __device__ __forceinline__ float get(int e) const { return v.get(e); }
    __device__ __forceinline__ void set(int e, float x) { v.set(e, x); }
};
template <> struct Row8<float> {
    f32x4 a, b;
    __device__ __forceinline__ void load(const float* p) { a = *reinterpret_cast<const f32x4*>(p); b = *reinterpret_cast<const f32x4*>(p + 4); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<f32x4*>(p) = a; *reinterpret_cast<f32x4*>(p + 4) = b; }
    __device__ __forceinline__ float get(int e) const { return e < 4 ? a[e] : b[e - 4]; }
    __device__ __forceinline__ void set(int e, float x) { if (e < 4) a[e] = x; else b[e - 4] = x; }
};

template <class T, int C> struct LnDxnCfg {
    static constexpr int NB = C / 16;
    static constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    // K chunks (of 32) per weight stage: the stage [C][32 KGC + pad] next to the fp32 tile [64][C + 1] must leave two workgroups per CU
    // where it can (C <= 192)
    static constexpr int KGC = C <= 192 ? 3 : (C <= 256 ? 2 : 1);
    static constexpr int LDW = 32 * KGC + PAD, LDF = C + 1;
    static constexpr int MAXV = C / 32;                               // 16-byte vectors of a token row per lane (4 lanes per token)
    static constexpr size_t f_floats = ((size_t)64 * LDF + 3) & ~(size_t)3;
    // the weight stage is dead when the fp32 tile is written (one barrier between them): they share the LDS, and the workgroup's footprint is
    // the larger of the two -- 33 KB at C = 128 instead of 62: three workgroups per CU (148 registers) where there were two
    static constexpr size_t w_bytes = (size_t)C * LDW * sizeof(T);
    static constexpr size_t bytes = f_floats * 4 > w_bytes ? f_floats * 4 : w_bytes;
    static constexpr int MINWG = C <= 128 ? 3 : 2;                    // workgroups per CU the register budget is held to
    static constexpr int NWV = (C * KGC * 4 + 255) / 256;             // 16-byte weight vectors per thread and stage
    static_assert(C % 32 == 0 && bytes <= 160 * 1024, "shape");
};

template <class T, int C, bool XF = false>     // XF: X and dX are fp32 (dY, W, d_res, XN stay T)
__global__ __launch_bounds__(256, (LnDxnCfg<T, C>::MINWG)) void ln_bwd_win_dxn_kernel(LnDxnDev a) {
    typedef typename std::conditional<XF, float, T>::type TX;
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    typedef LnDxnCfg<T, C> CF;
    constexpr int VEC = 8, NB = CF::NB, KGC = CF::KGC, LDW = CF::LDW, LDF = CF::LDF, MAXV = CF::MAXV, NWV = CF::NWV;
    const int K = a.K, NKC = K / 32, NGRP = (NKC + KGC - 1) / KGC;
    static_assert(sizeof(T) == 2, "16-bit types only");
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* Fs = reinterpret_cast<float*>(smem_v);                      // [64][LDF] d_xn, then d_xn * xhat
    T* Ws = reinterpret_cast<T*>(smem_v);                              // [C][LDW]  rows of Wqkv^T, one K group (the same LDS, before the tile exists)
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const TX* X = reinterpret_cast<const TX*>(a.X);
    const T* dRes = reinterpret_cast<const T*>(a.dRes);         // (may be null: no residual path)
    const T* dRes2 = reinterpret_cast<const T*>(a.dRes2);
    const T* WT = reinterpret_cast<const T*>(a.WT);
    TX* dX = reinterpret_cast<TX*>(a.dX);
    const int t = tid >> 2, q = tid & 3;
    // token t of this window in image order (cyclic shift + window partition as address arithmetic, net/MP_HSIR.py:672-677)
    const int nwx = a.W >> 3, nW = (a.H >> 3) * nwx;
    const int b = blockIdx.x / nW, wi = blockIdx.x % nW, wy = wi / nwx, wx = wi % nwx;
    const long pix = a.linear ? (long)blockIdx.x * 64 + t
                              : ((long)b * a.H + (wy * 8 + (t >> 3) + a.shift) % a.H) * a.W + (wx * 8 + (t & 7) + a.shift) % a.W;

    // ---- the lane's share of the token row (x, d_res): requested now, used after the GEMM
    constexpr bool DR_LATE = C >= 256;       // wide rows: d_res is requested after the GEMM (32 registers fewer held across it: no spills)
    Row8<TX> xv[MAXV];
    Vec16<T> dr[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        xv[k].load(X + pix * C + (q + 4 * k) * VEC);
        if (!DR_LATE && dRes) dr[k] = load16<T>(dRes + pix * C + (q + 4 * k) * VEC);
    }

    // ---- (a) d_xn = dQKV Wqkv: wave wv owns window rows 16 wv .. 16 wv + 15
    const T* drow = reinterpret_cast<const T*>(a.dQKV) + ((long)blockIdx.x * 64 + wv * 16 + (lane & 15)) * K + 8 * (lane >> 4);
    Vec16<T> wreg[NWV];
    auto wload = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < C * KGC * 4) {
                const int col = g * 32 * KGC + (idx % (KGC * 4)) * VEC;
                if (col < K) wreg[i] = load16<T>(WT + (long)(idx / (KGC * 4)) * K + col);
            }
        }
    };
    auto wstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < C * KGC * 4) store16<T>(Ws + (idx / (KGC * 4)) * LDW + (idx % (KGC * 4)) * VEC, wreg[i]);
        }
    };
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
    wload(0);
#pragma unroll 1
    for (int g = 0; g < NGRP; ++g) {
        frag_t xf[KGC];
#pragma unroll
        for (int kc = 0; kc < KGC; ++kc)
            if (g * KGC + kc < NKC) xf[kc] = *reinterpret_cast<const frag_t*>(drow + (g * KGC + kc) * 32);
        __syncthreads();                                   // the previous group's fragments have been read
        wstore();
        __syncthreads();
        if (g + 1 < NGRP) wload(g + 1);
#pragma unroll
        for (int kc = 0; kc < KGC; ++kc)
            if (g * KGC + kc < NKC) {                      // (uniform) the last group of a K that is not a multiple of 32 KGC is short
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) mma(acc[nb], load_frag<T>(Ws, LDW, nb * 16, kc * 32), xf[kc]);
            }
    }
    __syncthreads();                                       // every wave has read the last weight group: the tile takes its place
    {
        float* frow = Fs + (wv * 16 + (lane & 15)) * LDF + (lane >> 4) * 4;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) frow[nb * 16 + r] = acc[nb][r];
    }
    __syncthreads();

    // ---- (b) LayerNorm backward on the tile (ln_bwd_win_kernel's arithmetic, d_xn read from the fp32 tile)
    if (DR_LATE && dRes) {
#pragma unroll
        for (int k = 0; k < MAXV; ++k) dr[k] = load16<T>(dRes + pix * C + (q + 4 * k) * VEC);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
        for (int e = 0; e < VEC; ++e) s += xv[k].get(e);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float mean = s / (float)C;
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
        for (int e = 0; e < VEC; ++e) { const float d = xv[k].get(e) - mean; d2 += d * d; }
    d2 += __shfl_xor(d2, 1); d2 += __shfl_xor(d2, 2);
    const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int c0 = (q + 4 * k) * VEC;
        for (int e = 0; e < VEC; ++e) {
            const float gw = Fs[t * LDF + c0 + e] * a.ln_w[c0 + e], xh = (xv[k].get(e) - mean) * rstd;
            s1 += gw; s2 += gw * xh;
        }
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    s1 *= 1.0f / (float)C; s2 *= 1.0f / (float)C;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int c0 = (q + 4 * k) * VEC;
        Row8<TX> o;
        Vec16<T> r2;
        if (dRes2) r2 = load16<T>(dRes2 + pix * C + c0);
        for (int e = 0; e < VEC; ++e) {
            const float dxn = Fs[t * LDF + c0 + e], xh = (xv[k].get(e) - mean) * rstd;
            float v = rstd * (dxn * a.ln_w[c0 + e] - s1 - xh * s2);
            if (dRes) v += dr[k].get(e);
            if (dRes2) v += r2.get(e);
            o.set(e, v);
        }
        o.store(dX + pix * C + c0);
        if (a.XN) {
            Vec16<T> n;
            for (int e = 0; e < VEC; ++e) n.set(e, (xv[k].get(e) - mean) * rstd * a.ln_w[c0 + e] + a.ln_b[c0 + e]);
            store16<T>(reinterpret_cast<T*>(a.XN) + pix * C + c0, n);
        }
    }
    __syncthreads();
    float* part = a.part + (long)blockIdx.x * 2 * C;
    for (int c = tid; c < C; c += 256) {
        float sum = 0.f;
        for (int tt = 0; tt < 64; ++tt) sum += Fs[tt * LDF + c];
        part[C + c] = sum;                                 // d beta
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int c0 = (q + 4 * k) * VEC;
        for (int e = 0; e < VEC; ++e) Fs[t * LDF + c0 + e] *= (xv[k].get(e) - mean) * rstd;
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float sum = 0.f;
        for (int tt = 0; tt < 64; ++tt) sum += Fs[tt * LDF + c];
        part[c] = sum;                                     // d gamma
    }
}

template <class T> struct LnDxnShapes {
    static int run(const LnDxnDev& d, int C, bool xf, hipStream_t s) {
        const int nblk = d.linear ? d.B : d.B * (d.H / 8) * (d.W / 8);      // linear: B = number of 64-row blocks
#define MPHSIR_LD_CASE(c, XF)                                                                                             \
    if (C == c && xf == XF) {                                                                                              \
        constexpr size_t shmem = LnDxnCfg<T, c>::bytes;                                                                     \
        allow_big_lds(ln_bwd_win_dxn_kernel<T, c, XF>, shmem);                                                              \
        MPHSIR_LAUNCH(MPHSIR_K_LN_BWD_WIN, (ln_bwd_win_dxn_kernel<T, c, XF>), dim3(nblk), dim3(256), shmem, s, d);          \
        return MPHSIR_OK;                                                                                                  \
    }
        MPHSIR_LD_CASE(32, false) MPHSIR_LD_CASE(64, false) MPHSIR_LD_CASE(96, false) MPHSIR_LD_CASE(128, false) MPHSIR_LD_CASE(192, false)
        MPHSIR_LD_CASE(256, false)
        MPHSIR_LD_CASE(32, true) MPHSIR_LD_CASE(64, true) MPHSIR_LD_CASE(96, true) MPHSIR_LD_CASE(128, true) MPHSIR_LD_CASE(192, true)
#undef MPHSIR_LD_CASE
        return MPHSIR_EINVAL;
    }
};
template <> struct LnDxnShapes<float> {
    static int run(const LnDxnDev&, int, bool, hipStream_t) { return MPHSIR_EINVAL; }
};

}  // namespace mphsir

extern "C" int mphsir_ln_bwd_win_dxn_fits(int32_t C, int dtype) {
    return ((dtype == MPHSIR_BF16 || dtype == MPHSIR_F16) && (C == 32 || C == 64 || C == 96 || C == 128 || C == 192 || C == 256)) ? 1 : 0;      /* (C = 384: 90 spilled registers -- the remote-sensing latent level keeps the two launches) */
}

extern "C" int mphsir_ln_bwd_win_dxn(const void* X, const void* dQKV, const void* WqkvT, const void* dRes, const void* dRes2, const float* ln_w, void* dX,
                                     float* part, int32_t B, int32_t H, int32_t W, int32_t C, int32_t shift, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && dQKV && WqkvT && ln_w && dX && part, "ln_bwd_win_dxn: null pointer");
    MPHSIR_REQUIRE(mphsir_ln_bwd_win_dxn_fits(C, dtype), "ln_bwd_win_dxn: (C=%d, dtype=%d) not covered (16-bit types; ask mphsir_ln_bwd_win_dxn_fits)", C, dtype);
    MPHSIR_REQUIRE(B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && (shift == 0 || shift == 4), "ln_bwd_win_dxn: bad geometry");
    MPHSIR_REQUIRE(aligned16(X) && aligned16(dQKV) && aligned16(WqkvT) && aligned16(dRes) && aligned16(dRes2) && aligned16(dX),
                   "ln_bwd_win_dxn: 16-byte alignment required");
    LnDxnDev d{X, dQKV, WqkvT, dRes, ln_w, dX, part, B, H, W, shift, 3 * C, 0, nullptr, nullptr, dRes2};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, (LnDxnShapes<T_>::run(d, C, false, s)));
}

extern "C" int mphsir_ln_bwd_tok_dxn(const void* X, const void* dY, const void* WT, const void* dRes, const float* ln_w, const float* ln_b, void* dX,
                                     void* XN, float* part, int64_t M, int32_t C, int32_t K, int32_t x_f32, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && dY && WT && ln_w && dX && part && (!XN || ln_b), "ln_bwd_tok_dxn: null pointer (XN needs ln_b)");
    MPHSIR_REQUIRE(mphsir_ln_bwd_win_dxn_fits(C, dtype) && (!x_f32 || C <= 192),
                   "ln_bwd_tok_dxn: (C=%d, dtype=%d, x_f32=%d) not covered (16-bit types; ask mphsir_ln_bwd_win_dxn_fits; fp32 rows: C <= 192)", C, dtype, x_f32);
    MPHSIR_REQUIRE(M > 0 && M % 64 == 0 && K > 0 && K % 32 == 0, "ln_bwd_tok_dxn: M %% 64 == 0 and K %% 32 == 0 required (M=%ld, K=%d)", (long)M, K);
    MPHSIR_REQUIRE(aligned16(X) && aligned16(dY) && aligned16(WT) && aligned16(dRes) && aligned16(dX) && aligned16(XN), "ln_bwd_tok_dxn: 16-byte alignment required");
    LnDxnDev d{X, dY, WT, dRes, ln_w, dX, part, (int)(M / 64), 8, 8, 0, K, 1, ln_b, XN, nullptr};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, (LnDxnShapes<T_>::run(d, C, x_f32 != 0, s)));
}
