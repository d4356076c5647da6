// gated_mlp_fwd: Y = X + keep * ( fc2( value * gelu(gate) ) ),  [value|gate] = fc1( LayerNorm(X) )
//
// The whole second half of a PGSSTB block (net/MP_HSIR.py:719 with GatedMlp :66-82 and norm2 :619)
// in one kernel: the 2*hid-wide hidden activation never leaves the CU.  One 256-thread workgroup
// owns 64 tokens; the LayerNorm-ed rows sit in LDS; each wave owns 16 tokens and walks the hidden
// dimension in chunks of 32: two MFMA passes give value/gate for the chunk (D[hid][tok] layout, so
// four consecutive hidden units of one token are one lane's accumulator -> gate applied in
// registers, one 8/16-byte LDS store), then the chunk is immediately consumed as the K-slice of
// fc2 into C/16 persistent fp32 accumulator tiles.  Weights stream from L2 as MFMA fragments.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct MlpDev {
    const void* X; long ldx;
    const float* ln_w; const float* ln_b;
    const void* W1; const float* b1;     // [2*HP][C], [2*HP]  (value rows then gate rows, zero padded)
    const void* W2; const float* b2;     // [C][HP], [C]
    const float* keep; long rpb;
    void* Y; long ldy;
    int M, HP;
    int tpw;      // token tiles per wave: 0 = auto, 1, 2
    int hsplit;   // > 1 (LDS form): the hidden dimension is dealt to hsplit workgroups per token tile (grid.y); each writes its fp32
    float* Ypart; // partial fc2 product [hsplit][M][C]; mlp_combine_kernel adds them, the bias and the residual
    const void* R; long ldr;   // optional second residual: Y = (X + keep*mlp(LN(X))) + R  (the BaseBlock skip, net/MP_HSIR.py:727-761)
    // optional FUSED BRANCH SUM (round 5; the LDS forms with one token tile per wave, 16-bit types, C <= 128): the kernel's input row
    //   y = X + pkeep[b] * (PSA * pgate[window] + PV Mb[b]^T)          (PGSSTB.forward net/MP_HSIR.py:715-718: pass B of the channel
    // attention + the local gate + the first residual, what mphsir_gemm_tok epi 2 computes) is formed in LDS from the PV tile and the
    // sample's C x C matrix instead of being read from X; Yb (optional: training keeps y for the backward) receives it.
    const void* PV; long ldpv; const void* PM; long pms;
    const void* PSA; long ldpsa; const float* pgate; const float* pkeep;
    void* Yb; long ldyb;
    int H, Wimg, shift;
};

template <class T, int C>
__global__ __launch_bounds__(256) void gated_mlp_kernel(MlpDev a) {
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr int LDX = C + PAD;
    constexpr int LDH = 32 + PAD;
    constexpr int VEC = Vec16<T>::N;
    constexpr int NCT = C / 16;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* Xs = reinterpret_cast<T*>(smem_v);               // [64][LDX]  LN(x); later the output stage
    T* Hs = Xs + 64 * LDX;                              // [4][16][LDH] per-wave hidden chunk

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int m0 = blockIdx.x * 64;
    const T* X = reinterpret_cast<const T*>(a.X);

    // ---- LayerNorm into LDS: 4 adjacent lanes per token ------------------------------------
    {
        constexpr int NV = C / VEC;            // 16-byte vectors per row
        constexpr int VPT = NV / 4;            // vectors per thread
        const int r = tid >> 2, q = tid & 3;
        const T* row = X + (long)(m0 + r) * a.ldx;
        Vec16<T> xv[VPT];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            xv[i] = load16<T>(row + (q + 4 * i) * VEC);
            for (int e = 0; e < VEC; ++e) s += xv[i].get(e);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float mean = s / (float)C;
        float d2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i)
            for (int e = 0; e < VEC; ++e) { float d = xv[i].get(e) - mean; d2 += d * d; }
        d2 += __shfl_xor(d2, 1);
        d2 += __shfl_xor(d2, 2);
        const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            Vec16<T> o;
            for (int e = 0; e < VEC; ++e) o.set(e, (xv[i].get(e) - mean) * rstd * a.ln_w[c0 + e] + a.ln_b[c0 + e]);
            store16<T>(Xs + r * LDX + c0, o);
        }
    }
    __syncthreads();

    const T* W1 = reinterpret_cast<const T*>(a.W1);
    const T* W2 = reinterpret_cast<const T*>(a.W2);
    T* Hw = Hs + wv * 16 * LDH;
    const int HP = a.HP;
    f32x4 out[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) out[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int j = 0; j < HP; j += 32) {
        // fc1 for hidden units j..j+31 (value) and HP+j.. (gate), this wave's 16 tokens
        f32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0}, g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
#pragma unroll 4
        for (int kk = 0; kk < C; kk += TR::KCHUNK) {
            const frag_t bx = load_frag<T>(Xs, LDX, wv * 16, kk);
            mma(v0, load_frag<T>(W1, C, j, kk), bx);
            mma(v1, load_frag<T>(W1, C, j + 16, kk), bx);
            mma(g0, load_frag<T>(W1, C, HP + j, kk), bx);
            mma(g1, load_frag<T>(W1, C, HP + j + 16, kk), bx);
        }
        // lane: token lane&15, hidden rows (lane>>4)*4 + r
        const int hr = (lane >> 4) * 4;
        f32x4 h0, h1;
        for (int r = 0; r < 4; ++r) {
            h0[r] = (v0[r] + a.b1[j + hr + r]) * Math<T>::gelu(g0[r] + a.b1[HP + j + hr + r]);
            h1[r] = (v1[r] + a.b1[j + 16 + hr + r]) * Math<T>::gelu(g1[r] + a.b1[HP + j + 16 + hr + r]);
        }
        store4<T>(Hw + (lane & 15) * LDH + hr, h0);
        store4<T>(Hw + (lane & 15) * LDH + 16 + hr, h1);
        __syncthreads();
        // fc2 K-slice: out[co][tok] += W2[co][j..j+31] * H[tok][0..31]
#pragma unroll
        for (int kk = 0; kk < 32; kk += TR::KCHUNK) {
            const frag_t bh = load_frag<T>(Hw, LDH, 0, kk);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) mma(out[ct], load_frag<T>(W2, HP, ct * 16, j + kk), bh);
        }
        __syncthreads();
    }

    // ---- epilogue: (acc + b2) -> LDS stage (this wave's own rows), then coalesced residual + store
    {
        const int tok = wv * 16 + (lane & 15), cr = (lane >> 4) * 4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 o = out[ct];
            for (int r = 0; r < 4; ++r) o[r] += a.b2[ct * 16 + cr + r];
            store4<T>(Xs + tok * LDX + ct * 16 + cr, o);
        }
    }
    __syncthreads();
    T* Y = reinterpret_cast<T*>(a.Y);
    constexpr int NV = C / VEC;
    for (int idx = tid; idx < 64 * NV; idx += 256) {
        const int r = idx / NV, c0 = (idx % NV) * VEC, m = m0 + r;
        const float kf = a.keep ? a.keep[m / a.rpb] : 1.f;
        const Vec16<T> x = load16<T>(X + (long)m * a.ldx + c0);
        const Vec16<T> h = load16<T>(Xs + r * LDX + c0);
        Vec16<T> o;
        if (a.R) {
            const Vec16<T> rr = load16<T>(reinterpret_cast<const T*>(a.R) + (long)m * a.ldr + c0);
            for (int e = 0; e < VEC; ++e) o.set(e, (x.get(e) + kf * h.get(e)) + rr.get(e));
        } else {
            for (int e = 0; e < VEC; ++e) o.set(e, x.get(e) + kf * h.get(e));
        }
        store16<T>(Y + (long)m * a.ldy + c0, o);
    }
}

// ---- version with the weight chunks staged through LDS (shared by the 4 waves) ------------------------
// The direct version above streams every weight fragment from L1/L2 per wave (4x redundant, L1-bandwidth
// bound).  Here each 32-wide hidden chunk of fc1 (32 value + 32 gate rows, K = C) and of fc2 (C rows, K = 32)
// is loaded once per workgroup with coalesced 16-byte loads, and each wave owns TT token tiles (BM = 64*TT)
// so every weight fragment read from LDS feeds TT MFMAs.
// WF = 1: four waves (one per SIMD); WF = 2: eight waves, two per SIMD -- per 32-wide hidden chunk a wave issues ~48 MFMAs
// (768 cycles of its SIMD's matrix pipe) and then ~320 VALU instructions of GELU (1280 cycles): with one wave per SIMD the
// two never overlap (MFMA-busy 9.6 %); with two, one wave's GELU runs beside the other's MFMAs, and each staged weight
// chunk feeds twice the tokens.
// NWV = waves per workgroup: 4 (one per SIMD), 8 (two per SIMD, see above) or 2 (32 tokens per workgroup: on request only, see launch_mlp).
template <class T, int C, int TT, int NWV = 4> struct MlpLdsCfg {
    static constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    static constexpr int BM = 16 * TT * NWV, NTHR = 64 * NWV;
    static constexpr int LDX = C + PAD, LDH = 32 + PAD;
    static constexpr size_t ELEMS = (size_t)BM * LDX + 64 * LDX + (size_t)C * LDH + NWV * 16 * TT * LDH;
    static constexpr size_t BYTES = ELEMS * sizeof(T);
    static constexpr bool FITS = BYTES <= 160 * 1024;
};

// (the fused-sum form is held to 128 registers -- four waves per SIMD, two workgroups per CU like the plain form -- it asked for 154)
template <class T, int C, int TT, int NWV, bool PB = false>
__global__ __launch_bounds__(64 * NWV, PB ? 4 : (NWV == 8 ? 2 : 1)) void gated_mlp_lds_kernel(MlpDev a) {
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    typedef MlpLdsCfg<T, C, TT, NWV> CF;
    constexpr int LDX = CF::LDX, LDH = CF::LDH, BM = CF::BM, NTHR = CF::NTHR;
    constexpr int VEC = Vec16<T>::N;
    constexpr int NCT = C / 16;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* Xs = reinterpret_cast<T*>(smem_v);               // [BM][LDX]
    T* W1s = Xs + BM * LDX;                             // [64][LDX]  value rows 0..31, gate rows 32..63
    T* W2s = W1s + 64 * LDX;                            // [C][LDH]
    T* Hs = W2s + C * LDH;                              // [NWV][16*TT][LDH]
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int m0 = blockIdx.x * BM;
    const T* X = reinterpret_cast<const T*>(a.X);

    // ---- fused branch sum (PB): the tile's rows y = X + pkeep * (PSA * gate[window] + PV Mb^T) are formed here: in Xs for the
    // LayerNorm below (normalised in place), in registers for the residual of the epilogue, in Yb for the backward ------------------
    // (the rows y are NOT kept in registers for the epilogue's residual -- 16 registers across the whole hidden loop pushed this form
    // 18 registers over the 128 that two workgroups per CU leave: it spilled -- but re-read there by the thread that stored them, from
    // Yb, or, when the caller keeps no y, from the output rows Y, which receive y first and z later)
    constexpr int NVR = C / VEC, NIT = PB ? BM * NVR / NTHR : 1;
    T* Ykeep = nullptr;
    if constexpr (PB) {
        static_assert(TT == 1 && NWV == 8 && sizeof(T) == 2 && (BM * NVR) % NTHR == 0, "fused branch sum: eight waves, one token tile each, 16-bit types");
        constexpr int LDF = C + 4, NKC0 = C / TR::KCHUNK;
        static_assert((size_t)C * LDX <= (size_t)64 * LDX + (size_t)C * LDH + NWV * 16 * LDH, "the sample's matrix is staged over the weight tiles");
        static_assert((size_t)BM * LDF * 4 <= CF::BYTES, "fp32 stage of the product");
        const int hw = a.H * a.Wimg, bsmp = m0 / hw;     // a token tile never straddles samples (H W % BM == 0, checked by the host)
        const T* V = reinterpret_cast<const T*>(a.PV);
        const T* Mb = reinterpret_cast<const T*>(a.PM) + (long)bsmp * a.pms;
        T* Ms = W1s;                                     // [C][LDX] over W1s | W2s | Hs;  the V tile goes to Xs
        // (requesting the epilogue's rows of X and PSA here, under the staging and the product, measured level in round 5 and held 32
        // registers across the product: they are read where they are used)
        const T* SA = reinterpret_cast<const T*>(a.PSA);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + NTHR * it, r = idx / NVR, c0 = (idx % NVR) * VEC;
            store16<T>(Xs + r * LDX + c0, load16<T>(V + (long)(m0 + r) * a.ldpv + c0));
        }
        for (int idx = tid; idx < C * NVR; idx += NTHR) {
            const int r = idx / NVR, c0 = (idx % NVR) * VEC;
            store16<T>(Ms + r * LDX + c0, load16<T>(Mb + (long)r * C + c0));
        }
        __syncthreads();
        f32x4 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kc = 0; kc < NKC0; ++kc) {
            const frag_t bf = load_frag<T>(Xs, LDX, wv * 16, kc * TR::KCHUNK);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) mma(acc[ct], load_frag<T>(Ms, LDX, ct * 16, kc * TR::KCHUNK), bf);
        }
        __syncthreads();                                 // every wave has read the V tile and the matrix: the fp32 stage goes over them
        float* Cf = reinterpret_cast<float*>(smem_v);    // [BM][LDF]
        {
            const int tok = wv * 16 + (lane & 15), cr = (lane >> 4) * 4;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(Cf + tok * LDF + ct * 16 + cr) = acc[ct];
        }
        __syncthreads();
        // a thread finishes 8 consecutive channels of a token: 16-byte loads of X and PSA, one rounding -- the arithmetic of
        // gemm_tok's epilogue 2, element for element
        Ykeep = a.Yb ? reinterpret_cast<T*>(a.Yb) : reinterpret_cast<T*>(a.Y);
        const long ldk = a.Yb ? a.ldyb : a.ldy;
        Vec16<T> yreg[NIT];
        const float kf1 = a.pkeep ? a.pkeep[bsmp] : 1.f;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + NTHR * it, r = idx / NVR, c = (idx % NVR) * VEC;
            const long m = (long)m0 + r;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(Cf + r * LDF + c), v1 = *reinterpret_cast<const f32x4*>(Cf + r * LDF + c + 4);
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const int p = (int)(m - (long)bsmp * hw), y = p / a.Wimg, x = p % a.Wimg;
            const int ys = (y - a.shift + a.H) % a.H, xs = (x - a.shift + a.Wimg) % a.Wimg;          // shifted-frame coordinates
            const float* gp = a.pgate + ((long)bsmp * (hw / 64) + (ys >> 3) * (a.Wimg >> 3) + (xs >> 3)) * C + c;
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
            const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
            const Vec16<T> rr = load16<T>(X + m * a.ldx + c), sa = load16<T>(SA + m * a.ldpsa + c);
            Vec16<T> o;
            for (int e = 0; e < 8; ++e) o.set(e, rr.get(e) + kf1 * (sa.get(e) * g[e] + v[e]));
            yreg[it] = o;
            store16<T>(Ykeep + m * ldk + c, o);
        }
        __syncthreads();                                 // the stage is consumed: the rows go where the LayerNorm expects them
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + NTHR * it, r = idx / NVR, c = (idx % NVR) * VEC;
            store16<T>(Xs + r * LDX + c, yreg[it]);
        }
        __syncthreads();
    }

    // ---- LayerNorm into LDS: 4 adjacent lanes per token, TT passes ---------------------------------
    for (int pass = 0; pass < TT; ++pass) {
        constexpr int NV = C / VEC, VPT = NV / 4;
        const int r = pass * 16 * NWV + (tid >> 2), q = tid & 3;
        const T* row = PB ? Xs + r * LDX : X + (long)(m0 + r) * a.ldx;
        Vec16<T> xv[VPT];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            xv[i] = load16<T>(row + (q + 4 * i) * VEC);
            for (int e = 0; e < VEC; ++e) s += xv[i].get(e);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float mean = s / (float)C;
        float d2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i)
            for (int e = 0; e < VEC; ++e) { float d = xv[i].get(e) - mean; d2 += d * d; }
        d2 += __shfl_xor(d2, 1);
        d2 += __shfl_xor(d2, 2);
        const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            Vec16<T> o;
            for (int e = 0; e < VEC; ++e) o.set(e, (xv[i].get(e) - mean) * rstd * a.ln_w[c0 + e] + a.ln_b[c0 + e]);
            store16<T>(Xs + r * LDX + c0, o);
        }
    }

    __syncthreads();
    // the wave's LN(x) fragments stay in registers for the whole hidden-chunk loop (Xs is only reused by the epilogue)
    constexpr int NKC = C / TR::KCHUNK;
    frag_t bx[TT][NKC];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) bx[t][kc] = load_frag<T>(Xs, LDX, (wv * TT + t) * 16, kc * TR::KCHUNK);

    const T* W1 = reinterpret_cast<const T*>(a.W1);
    const T* W2 = reinterpret_cast<const T*>(a.W2);
    T* Hw = Hs + wv * 16 * TT * LDH;
    const int HP = a.HP;
    // weight slices of one 32-wide hidden chunk: global -> registers (prefetched one chunk ahead) -> LDS
    constexpr int VPR = C / VEC, VPH = 32 / VEC;
    constexpr int NW1 = 64 * VPR, NW2 = C * VPH, NVW = NW1 + NW2, NPT = (NVW + NTHR - 1) / NTHR;
    Vec16<T> wreg[NPT];
    auto wload = [&](int j) {
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const int v = tid + NTHR * it;
            if (v < NW1) {
                const int r = v / VPR, c = (v % VPR) * VEC;
                wreg[it] = load16<T>(W1 + (long)(r < 32 ? j + r : HP + j + (r - 32)) * C + c);
            } else if (v < NVW) {
                const int u = v - NW1, r = u / VPH, c = (u % VPH) * VEC;
                wreg[it] = load16<T>(W2 + (long)r * HP + j + c);
            }
        }
    };
    auto wstore = [&]() {
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const int v = tid + NTHR * it;
            if (v < NW1) {
                const int r = v / VPR, c = (v % VPR) * VEC;
                store16<T>(W1s + r * LDX + c, wreg[it]);
            } else if (v < NVW) {
                const int u = v - NW1, r = u / VPH, c = (u % VPH) * VEC;
                store16<T>(W2s + r * LDH + c, wreg[it]);
            }
        }
    };
    // hidden split (small launches: the latent level has 128 / 64 token tiles for 256 CUs and every workgroup streams all weight
    // chunks through LDS): workgroup (tile, blockIdx.y) takes hidden columns [j0, j1) -- the weights are still read ONCE in total
    const int hsp = a.hsplit > 1 ? a.hsplit : 1, j0 = (int)blockIdx.y * (HP / hsp), j1 = j0 + HP / hsp;
    wload(j0);
    f32x4 out[TT][NCT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int i = 0; i < NCT; ++i) out[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int j = j0; j < j1; j += 32) {
        __syncthreads();                                   // previous chunk's weight tiles fully consumed
        wstore();
        __syncthreads();
        if (j + 32 < j1) wload(j + 32);                    // in flight during the MFMAs below
        f32x4 vv[TT][2], gg[TT][2];
#pragma unroll
        for (int t = 0; t < TT; ++t)
            for (int u = 0; u < 2; ++u) { vv[t][u] = f32x4{0.f, 0.f, 0.f, 0.f}; gg[t][u] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        // (requesting the four weight fragments of K chunk kc + 1 before the MFMAs of chunk kc, with scheduling barriers, was measured
        // SLOWER here -- M = 131072, C = 128: 65.8 -> 72.7 us -- and is gone; the backward and the weight-gradient kernel keep it)
        {
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int kk = kc * TR::KCHUNK;
                const frag_t wv0 = load_frag<T>(W1s, LDX, 0, kk), wv1 = load_frag<T>(W1s, LDX, 16, kk);
                const frag_t wg0 = load_frag<T>(W1s, LDX, 32, kk), wg1 = load_frag<T>(W1s, LDX, 48, kk);
#pragma unroll
                for (int t = 0; t < TT; ++t) {
                    mma(vv[t][0], wv0, bx[t][kc]);
                    mma(vv[t][1], wv1, bx[t][kc]);
                    mma(gg[t][0], wg0, bx[t][kc]);
                    mma(gg[t][1], wg1, bx[t][kc]);
                }
            }
        }
        const int hr = (lane >> 4) * 4;
#pragma unroll
        for (int t = 0; t < TT; ++t)
            for (int u = 0; u < 2; ++u) {
                f32x4 h;
                for (int r = 0; r < 4; ++r)
                    h[r] = (vv[t][u][r] + a.b1[j + u * 16 + hr + r]) * Math<T>::gelu(gg[t][u][r] + a.b1[HP + j + u * 16 + hr + r]);
                store4<T>(Hw + (t * 16 + (lane & 15)) * LDH + u * 16 + hr, h);
            }
        wave_barrier();                                    // Hw is wave-private
        {
#pragma unroll
            for (int kk = 0; kk < 32; kk += TR::KCHUNK) {
                frag_t bh[TT];
#pragma unroll
                for (int t = 0; t < TT; ++t) bh[t] = load_frag<T>(Hw, LDH, t * 16, kk);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const frag_t w2 = load_frag<T>(W2s, LDH, ct * 16, kk);
#pragma unroll
                    for (int t = 0; t < TT; ++t) mma(out[t][ct], w2, bh[t]);
                }
            }
        }
        wave_barrier();
    }
    __syncthreads();

    if (hsp > 1) {            // partial product of this hidden range: fp32, 4 consecutive channels of a token per lane
        float* Yp = a.Ypart + ((long)blockIdx.y * a.M + m0) * C;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int tok = (wv * TT + t) * 16 + (lane & 15), cr = (lane >> 4) * 4;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(Yp + (long)tok * C + ct * 16 + cr) = out[t][ct];
        }
        return;
    }
    // ---- epilogue: (acc + b2) -> LDS stage (own rows), then coalesced residual + store -----------------
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int tok = (wv * TT + t) * 16 + (lane & 15), cr = (lane >> 4) * 4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 o = out[t][ct];
            for (int r = 0; r < 4; ++r) o[r] += a.b2[ct * 16 + cr + r];
            store4<T>(Xs + tok * LDX + ct * 16 + cr, o);
        }
    }
    __syncthreads();
    T* Y = reinterpret_cast<T*>(a.Y);
    constexpr int NV = C / VEC;
    constexpr int NEP = BM * NV / NTHR;                 // = TT * NV / 4: whole
    static_assert((BM * NV) % NTHR == 0, "epilogue: whole passes");
#pragma unroll PB ? 2 : 1
    for (int it = 0; it < NEP; ++it) {
        const int idx = tid + it * NTHR, r = idx / NV, c0 = (idx % NV) * VEC, m = m0 + r;
        const float kf = a.keep ? a.keep[m / a.rpb] : 1.f;
        Vec16<T> x;
        if constexpr (PB) x = load16<T>(Ykeep + (long)m * (a.Yb ? a.ldyb : a.ldy) + c0);      // this thread's own store of the prologue
        else x = load16<T>(X + (long)m * a.ldx + c0);
        const Vec16<T> h = load16<T>(Xs + r * LDX + c0);
        Vec16<T> o;
        if (a.R) {
            const Vec16<T> rr = load16<T>(reinterpret_cast<const T*>(a.R) + (long)m * a.ldr + c0);
            for (int e = 0; e < VEC; ++e) o.set(e, (x.get(e) + kf * h.get(e)) + rr.get(e));
        } else {
            for (int e = 0; e < VEC; ++e) o.set(e, x.get(e) + kf * h.get(e));
        }
        store16<T>(Y + (long)m * a.ldy + c0, o);
    }
}

// y = x + keep * (sum of the hidden-split partials + b2): 8 (fp32: 4) channels of a token per thread
template <class T>
__global__ __launch_bounds__(256) void mlp_combine_kernel(MlpDev a, int C) {
    constexpr int VEC = Vec16<T>::N;
    const long nv = (long)a.M * C / VEC;
    const T* X = reinterpret_cast<const T*>(a.X);
    T* Y = reinterpret_cast<T*>(a.Y);
    for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < nv; v += (long)gridDim.x * 256) {
        const long e0 = v * VEC, m = e0 / C;
        const int c0 = (int)(e0 - m * C);
        float acc[VEC];
        for (int e = 0; e < VEC; ++e) acc[e] = a.b2[c0 + e];
        for (int sp = 0; sp < a.hsplit; ++sp) {            // fixed order: deterministic
            const float* p = a.Ypart + ((long)sp * a.M + m) * C + c0;
            for (int q = 0; q < VEC; q += 4) {
                const f32x4 pv = *reinterpret_cast<const f32x4*>(p + q);
                for (int r = 0; r < 4; ++r) acc[q + r] += pv[r];
            }
        }
        const float kf = a.keep ? a.keep[m / a.rpb] : 1.f;
        const Vec16<T> x = load16<T>(X + m * a.ldx + c0);
        Vec16<T> o;
        if (a.R) {
            const Vec16<T> rr = load16<T>(reinterpret_cast<const T*>(a.R) + m * a.ldr + c0);
            for (int e = 0; e < VEC; ++e) o.set(e, (x.get(e) + kf * acc[e]) + rr.get(e));
        } else {
            for (int e = 0; e < VEC; ++e) o.set(e, x.get(e) + kf * acc[e]);
        }
        store16<T>(Y + m * a.ldy + c0, o);
    }
}

template <class T, int C, int TT, int NWV = 4>
static int launch_mlp_lds(const MlpDev& d, hipStream_t s) {
    typedef MlpLdsCfg<T, C, TT, NWV> CF;
    if constexpr (!CF::FITS) {
        return 1;   // caller falls back
    } else {
        if constexpr (sizeof(T) == 2 && TT == 1 && C <= 128 && NWV == 8) {
            if (d.PV) {                       // fused branch sum: same LDS, the y rows live in the X tile / in registers
                allow_big_lds(gated_mlp_lds_kernel<T, C, TT, NWV, true>, CF::BYTES);
                MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP, (gated_mlp_lds_kernel<T, C, TT, NWV, true>), dim3(d.M / CF::BM, 1), dim3(CF::NTHR), CF::BYTES, s, d);
                return MPHSIR_OK;
            }
        }
        if (d.PV) { set_error("gated_mlp: the fused branch sum needs the eight-wave LDS form"); return MPHSIR_EINVAL; }
        allow_big_lds(gated_mlp_lds_kernel<T, C, TT, NWV>, CF::BYTES);
        const int hsp = d.hsplit > 1 ? d.hsplit : 1;
        MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP, (gated_mlp_lds_kernel<T, C, TT, NWV>), dim3(d.M / CF::BM, hsp), dim3(CF::NTHR), CF::BYTES, s, d);
        if (hsp > 1) {
            long blocks = ((long)d.M * C / Vec16<T>::N + 255) / 256;
            MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP, (mlp_combine_kernel<T>), dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, s, d, C);
        }
        return MPHSIR_OK;
    }
}

template <class T, int C>
static int launch_mlp(const MlpDev& d, hipStream_t s) {
    // preferred: LDS-staged weights; two token tiles per wave when there are enough tokens to fill the chip
    int rc = 1;
    // Eight waves with one 16-token tile each (two waves per SIMD) wherever that still leaves a workgroup per CU -- measured at
    // M = 131072, C = 128: 68 us against 92 (four waves, one tile) and 112 (four waves, two tiles: the round-2 default);
    // tpw 1 / 2 force the four-wave forms, 3 / 4 the eight-wave ones (tests, tools/bench/bench_mlp_fwd.py).  (A two-wave form, 32 tokens
    // per workgroup, was measured slower wherever it was tried -- round 4 -- and is gone; small launches split the HIDDEN dimension.)
    if (sizeof(T) == 2 && d.M % 128 == 0 && (d.tpw == 3 || (d.tpw == 0 && d.M / 128 >= 256))) rc = launch_mlp_lds<T, C, 1, 8>(d, s);
    if (rc == 1 && sizeof(T) == 2 && d.M % 256 == 0 && d.tpw == 4) rc = launch_mlp_lds<T, C, 2, 8>(d, s);
    if (rc == 1 && d.M % 128 == 0 && d.tpw == 2) rc = launch_mlp_lds<T, C, 2>(d, s);
    if (rc == 1) rc = launch_mlp_lds<T, C, 1>(d, s);
    if (rc != 1) return rc;
    if (d.hsplit > 1) { set_error("gated_mlp: the hidden split needs the LDS-staged form, which does not fit C=%d in this element type", C); return MPHSIR_EINVAL; }
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    const size_t shmem = (64 * (C + PAD) + 4 * 16 * (32 + PAD)) * sizeof(T);
    allow_big_lds(gated_mlp_kernel<T, C>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP, (gated_mlp_kernel<T, C>), dim3(d.M / 64), dim3(256), shmem, s, d);
    return MPHSIR_OK;
}

template <class T>
static int dispatch_mlp(const MlpDev& d, int C, hipStream_t s) {
    switch (C) {
        case 32: return launch_mlp<T, 32>(d, s);
        case 64: return launch_mlp<T, 64>(d, s);
        case 96: return launch_mlp<T, 96>(d, s);
        case 128: return launch_mlp<T, 128>(d, s);
        case 192: return launch_mlp<T, 192>(d, s);
        case 256: return launch_mlp<T, 256>(d, s);
        case 384: return launch_mlp<T, 384>(d, s);
    }
    set_error("gated_mlp: C=%d not instantiated (32,64,96,128,192,256,384)", C);
    return MPHSIR_EINVAL;
}

}  // namespace mphsir

extern "C" int mphsir_gated_mlp_fwd_fuses(int32_t C, int64_t M, int dtype) {
    return ((dtype == MPHSIR_BF16 || dtype == MPHSIR_F16) && (C == 32 || C == 64 || C == 96 || C == 128) && M > 0 && M % 128 == 0) ? 1 : 0;
}

extern "C" int mphsir_gated_mlp_fwd(const mphsir_mlp_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "gated_mlp_fwd");
    MPHSIR_REQUIRE(a && a->X && a->Y && a->W1 && a->W2 && a->b1 && a->b2 && a->ln_w && a->ln_b, "gated_mlp: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "gated_mlp: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2;
    MPHSIR_REQUIRE(a->M > 0 && a->M % 64 == 0, "gated_mlp: M must be a positive multiple of 64");
    MPHSIR_REQUIRE(a->HP > 0 && a->HP % 32 == 0, "gated_mlp: padded hidden width must be a multiple of 32");
    MPHSIR_REQUIRE(aligned16(a->X) && aligned16(a->Y) && aligned16(a->W1) && aligned16(a->W2) &&
                       (a->ldx * esz) % 16 == 0 && (a->ldy * esz) % 16 == 0, "gated_mlp: 16-byte alignment required");
    if (a->keep) MPHSIR_REQUIRE(a->rows_per_batch > 0, "gated_mlp: keep needs rows_per_batch");
    MlpDev d{a->X, (long)a->ldx, a->ln_w, a->ln_b, a->W1, a->b1, a->W2, a->b2, a->keep,
             (long)(a->rows_per_batch > 0 ? a->rows_per_batch : a->M), a->Y, (long)a->ldy, (int)a->M, a->HP, a->tiles_per_wave,
             a->hsplit, a->ypart, a->R, (long)a->ldr,
             a->PV, (long)a->ldpv, a->PM, (long)a->pm_batch_stride, a->PSA, (long)a->ldpsa, a->pgate, a->pkeep, a->Yb, (long)a->ldyb,
             a->H, a->Wimg, a->shift};
    if (a->PV) {
        d.tpw = 3;                              // the eight-wave form, one token tile per wave
        MPHSIR_REQUIRE(mphsir_gated_mlp_fwd_fuses(a->C, a->M, dtype) && a->hsplit <= 1 && (a->tiles_per_wave == 0 || a->tiles_per_wave == 3),
                       "gated_mlp: the fused branch sum exists for 16-bit types, C in {32,64,96,128}, one token tile per wave, no hidden split");
        MPHSIR_REQUIRE(a->PM && a->PSA && a->pgate && aligned16(a->PV) && aligned16(a->PM) && aligned16(a->PSA) && (a->ldpv * esz) % 16 == 0 &&
                           (a->ldpsa * esz) % 16 == 0 && (a->pm_batch_stride * esz) % 16 == 0 && (!a->Yb || (aligned16(a->Yb) && (a->ldyb * esz) % 16 == 0)),
                       "gated_mlp: fused branch sum: PM, PSA, pgate required, 16-byte alignment");
        MPHSIR_REQUIRE(a->H > 0 && a->Wimg > 0 && a->H % 8 == 0 && a->Wimg % 8 == 0 && ((int64_t)a->H * a->Wimg) % 128 == 0 && a->M % ((int64_t)a->H * a->Wimg) == 0 &&
                           (a->shift == 0 || a->shift == 4), "gated_mlp: fused branch sum: H, W multiples of 8 with H*W %% 128 == 0, M = B*H*W, shift 0 or 4");
    }
    if (a->R) MPHSIR_REQUIRE(aligned16(a->R) && (a->ldr * esz) % 16 == 0, "gated_mlp: second residual: 16-byte alignment required");
    if (a->hsplit > 1)
        MPHSIR_REQUIRE(a->ypart && aligned16(a->ypart) && a->HP % (32 * a->hsplit) == 0 && a->tiles_per_wave <= 2,
                       "gated_mlp: hsplit needs a workspace ypart [hsplit][M][C] fp32, HP %% (32 hsplit) == 0 and a four-wave form");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, (dispatch_mlp<T_>(d, a->C, s)));
}
