// gated_mlp_bwd: data gradient of  y = x + keep * fc2(value * gelu(gate)),  [value|gate] = fc1(LN(x)).
//
// Backward of PGSSTB's second residual branch (reference forward: net/MP_HSIR.py:719, GatedMlp :66-82,
// norm2 :619).  Same tile structure as gated_mlp_fwd (64 tokens per workgroup, 16 per wave, hidden
// dimension walked in chunks of 32, everything after the LayerNorm staging is wave-local):
//   recompute value/gate for the chunk (fc1 on the LN-ed tile in LDS),
//   dh   = dm W2            (dm = keep * dy staged in LDS, W2^T streamed from L2),
//   dval = dh * gelu(gate),  dgate = dh * value * gelu'(gate)             (registers),
//   dxn += [dval|dgate] W1   (C/16 persistent fp32 accumulator tiles per wave),
// then LayerNorm backward per token and dx = dy + that.  h = value*gelu(gate), [dval|dgate] and LN(x)
// are written out once so that the four weight gradients are plain token-reduction GEMMs
// (dW2 = dm^T h, dW1 = [dval|dgate]^T xn, done by the caller with mphsir_gemm_tn) and the bias / LN
// parameter gradients are column sums (per-workgroup partials here, no atomics).
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct MlpBwdDev {
    const void* X; const void* dY; const void* DM;   // [M][C] each (ld = C)
    const float* ln_w; const float* ln_b;
    const void* W1; const float* b1;                 // [2*HP][C], [2*HP]
    const void* W1T;                                 // [C][2*HP]
    const void* W2T;                                 // [HP][C]
    void* dX;                                        // [M][C]
    void* XN; void* H; void* DPRE;                   // [M][C], [M][HP], [M][2*HP]
    float* part;                                     // [M/64][2][C]: d(ln weight), d(ln bias) partial sums
    int M, HP;
    const float* keep; long rpb;                     // optional: DM is written here as keep[row / rpb] * dY
    int hsplit;                                      // > 1 (second form, four waves): the hidden dimension is dealt to hsplit workgroups per
    float* dxn_part;                                 // token tile (grid.y); each writes its fp32 partial d_xn [hsplit][M][C]; mlp_bwd_combine_kernel
};                                                   // sums them in order and finishes (LayerNorm backward, residual, parameter partials)

// STAGE = true: the fc1 rows of the hidden chunk ([64][C]) and the matching columns of W1^T ([C][64]) are loaded
// once per workgroup into LDS (coalesced) instead of every wave streaming its own fragments through L1.
// GX = true (widths whose two [64][C] token tiles do not fit LDS, i.e. fp32 at C = 384): the LN(x) / dm operands are
// not kept in LDS; their fragments are read back from the XN / DM rows this workgroup has just written.
template <class T, int C, bool STAGE, bool GX = false>
__global__ __launch_bounds__(256) void gated_mlp_bwd_kernel(MlpBwdDev a) {
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr int LDX = C + PAD;
    constexpr int LDH = 64 + PAD;
    constexpr int LDF = C + 4;                          // fp32 staging of dxn
    constexpr int VEC = Vec16<T>::N;
    constexpr int NCT = C / 16;
    constexpr int NV = C / VEC, VPT = NV / 4;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* Xs = reinterpret_cast<T*>(smem_v);               // [64][LDX]  LN(x)
    T* Ds = Xs + 64 * LDX;                              // [64][LDX]  dm = keep*dy
    T* Hs = GX ? Xs + 64 * LDF * 4 / sizeof(T) : Ds + 64 * LDX;   // [4][16][LDH]  per wave: [dval(32) | dgate(32)], also h staging
    float* stat = reinterpret_cast<float*>(Hs + 4 * 16 * LDH);   // mean[64], rstd[64]
    T* W1s = reinterpret_cast<T*>(stat + 128);          // [64][LDX]   (STAGE) value rows 0..31, gate rows 32..63
    T* W1Ts = W1s + 64 * LDX;                           // [C][LDH]    (STAGE) W1^T columns: value 0..31 | gate 32..63
    float* Fs = reinterpret_cast<float*>(smem_v);       // [64][LDF] fp32 dxn (aliases Xs|Ds after the main loop)
    static_assert(GX || 64 * LDF * 4 <= 2 * 64 * LDX * sizeof(T), "fp32 dxn stage must fit in the two token tiles");
    static_assert(!(GX && STAGE), "the global-operand form streams its weights");

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int m0 = blockIdx.x * 64;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* DM = reinterpret_cast<const T*>(a.DM);
    const T* dY = reinterpret_cast<const T*>(a.dY);

    // ---- stage LN(x) (also written to XN) and dm; 4 adjacent lanes per token --------------------
    {
        const int r = tid >> 2, q = tid & 3;
        const T* row = X + (long)(m0 + r) * C;
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
        if (q == 0) { stat[r] = mean; stat[64 + r] = rstd; }
        T* XN = reinterpret_cast<T*>(a.XN) + (long)(m0 + r) * C;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            Vec16<T> o;
            for (int e = 0; e < VEC; ++e) o.set(e, (xv[i].get(e) - mean) * rstd * a.ln_w[c0 + e] + a.ln_b[c0 + e]);
            if (!GX) store16<T>(Xs + r * LDX + c0, o);
            store16<T>(XN + c0, o);
            Vec16<T> dmv;
            if (a.keep) {                             // DropPath backward fused: dm = keep[b] * dy, also kept for dW2 / db2
                const float kf = a.keep[(m0 + r) / a.rpb];
                const Vec16<T> dyv = load16<T>(dY + (long)(m0 + r) * C + c0);
                for (int e = 0; e < VEC; ++e) dmv.set(e, kf * dyv.get(e));
                store16<T>(const_cast<T*>(DM) + (long)(m0 + r) * C + c0, dmv);
            } else {
                dmv = load16<T>(DM + (long)(m0 + r) * C + c0);
            }
            if (!GX) store16<T>(Ds + r * LDX + c0, dmv);
        }
    }
    __syncthreads();
    const T* Xop = GX ? reinterpret_cast<const T*>(a.XN) + (long)m0 * C : Xs;      // B operands of the chunk loop
    const T* Dop = GX ? DM + (long)m0 * C : Ds;
    constexpr int LDO = GX ? C : LDX;

    const T* W1 = reinterpret_cast<const T*>(a.W1);
    const T* W1T = reinterpret_cast<const T*>(a.W1T);
    const T* W2T = reinterpret_cast<const T*>(a.W2T);
    T* Hw = Hs + wv * 16 * LDH;
    const int HP = a.HP;
    T* Hout = reinterpret_cast<T*>(a.H);
    T* Pout = reinterpret_cast<T*>(a.DPRE);
    f32x4 out[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) out[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int j = 0; j < HP; j += 32) {
        if (STAGE) {
            constexpr int VPR = C / VEC, VPH = 32 / VEC;
            for (int v = tid; v < 64 * VPR; v += 256) {
                const int r = v / VPR, c = (v % VPR) * VEC;
                const long srow = r < 32 ? j + r : HP + j + (r - 32);
                store16<T>(W1s + r * LDX + c, load16<T>(W1 + srow * C + c));
            }
            for (int v = tid; v < C * 2 * VPH; v += 256) {
                const int r = v / (2 * VPH), seg = (v / VPH) & 1, c = (v % VPH) * VEC;
                store16<T>(W1Ts + r * LDH + seg * 32 + c, load16<T>(W1T + (long)r * 2 * HP + seg * HP + j + c));
            }
            __syncthreads();
        }
        f32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0}, g0 = {0, 0, 0, 0}, g1 = {0, 0, 0, 0};
        f32x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};          // dh tiles
#pragma unroll 4
        for (int kk = 0; kk < C; kk += TR::KCHUNK) {
            const frag_t bx = load_frag<T>(Xop, LDO, wv * 16, kk);
            const frag_t bd = load_frag<T>(Dop, LDO, wv * 16, kk);
            if (STAGE) {
                mma(v0, load_frag<T>(W1s, LDX, 0, kk), bx);
                mma(v1, load_frag<T>(W1s, LDX, 16, kk), bx);
                mma(g0, load_frag<T>(W1s, LDX, 32, kk), bx);
                mma(g1, load_frag<T>(W1s, LDX, 48, kk), bx);
            } else {
                mma(v0, load_frag<T>(W1, C, j, kk), bx);
                mma(v1, load_frag<T>(W1, C, j + 16, kk), bx);
                mma(g0, load_frag<T>(W1, C, HP + j, kk), bx);
                mma(g1, load_frag<T>(W1, C, HP + j + 16, kk), bx);
            }
            mma(e0, load_frag<T>(W2T, C, j, kk), bd);
            mma(e1, load_frag<T>(W2T, C, j + 16, kk), bd);
        }
        const int hr = (lane >> 4) * 4, tk = lane & 15;
        f32x4 h0, h1, dv0, dv1, dg0, dg1;
        for (int r = 0; r < 4; ++r) {
            const float va = v0[r] + a.b1[j + hr + r], ga = g0[r] + a.b1[HP + j + hr + r];
            const float vb = v1[r] + a.b1[j + 16 + hr + r], gb = g1[r] + a.b1[HP + j + 16 + hr + r];
            float ea, eb, da, db;
            Math<T>::gelu_pair(ga, ea, da);
            Math<T>::gelu_pair(gb, eb, db);
            h0[r] = va * ea;            h1[r] = vb * eb;
            dv0[r] = e0[r] * ea;        dv1[r] = e1[r] * eb;
            dg0[r] = e0[r] * va * da;
            dg1[r] = e1[r] * vb * db;
        }
        // h chunk -> HBM through the per-wave LDS buffer (whole 64-byte row segments per token); not when the parameter gradients
        // come from mphsir_gated_mlp_wgrad (H == DPRE == NULL)
        if (Hout) {
            store4<T>(Hw + tk * LDH + hr, h0);
            store4<T>(Hw + tk * LDH + 16 + hr, h1);
            __syncthreads();
            {
                constexpr int VPR = 32 / VEC;                    // vectors per 32-wide row segment
                for (int i = lane; i < 16 * VPR; i += 64) {
                    const int t = i / VPR, c = (i % VPR) * VEC;
                    store16<T>(Hout + (long)(m0 + wv * 16 + t) * HP + j + c, load16<T>(Hw + t * LDH + c));
                }
            }
            __syncthreads();
        }
        store4<T>(Hw + tk * LDH + hr, dv0);
        store4<T>(Hw + tk * LDH + 16 + hr, dv1);
        store4<T>(Hw + tk * LDH + 32 + hr, dg0);
        store4<T>(Hw + tk * LDH + 48 + hr, dg1);
        __syncthreads();
        if (Pout) {
            constexpr int VPR = 32 / VEC;
            for (int i = lane; i < 16 * 2 * VPR; i += 64) {
                const int t = i / (2 * VPR), seg = (i / VPR) & 1, c = (i % VPR) * VEC;
                store16<T>(Pout + (long)(m0 + wv * 16 + t) * 2 * HP + seg * HP + j + c, load16<T>(Hw + t * LDH + seg * 32 + c));
            }
        }
        // dxn[c][tok] += W1T[c][j..j+31] * dval + W1T[c][HP+j..] * dgate
#pragma unroll
        for (int kk = 0; kk < 32; kk += TR::KCHUNK) {
            const frag_t bv = load_frag<T>(Hw, LDH, 0, kk);
            const frag_t bg = load_frag<T>(Hw, LDH, 0, 32 + kk);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                if (STAGE) {
                    mma(out[ct], load_frag<T>(W1Ts, LDH, ct * 16, kk), bv);
                    mma(out[ct], load_frag<T>(W1Ts, LDH, ct * 16, 32 + kk), bg);
                } else {
                    mma(out[ct], load_frag<T>(W1T, 2 * HP, ct * 16, j + kk), bv);
                    mma(out[ct], load_frag<T>(W1T, 2 * HP, ct * 16, HP + j + kk), bg);
                }
            }
        }
        __syncthreads();
    }

    // ---- dxn -> fp32 LDS stage; LayerNorm backward; dx = dy + ...; parameter-gradient partials -----
    {
        const int tok = wv * 16 + (lane & 15), cr = (lane >> 4) * 4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
            *reinterpret_cast<f32x4*>(Fs + tok * LDF + ct * 16 + cr) = out[ct];
    }
    __syncthreads();
    float* part = a.part + (long)blockIdx.x * 2 * C;
    for (int c = tid; c < C; c += 256) {                 // d(ln bias)[c] = sum_tok dxn
        float s = 0.f;
        for (int t = 0; t < 64; ++t) s += Fs[t * LDF + c];
        part[C + c] = s;
    }
    __syncthreads();
    {
        const int r = tid >> 2, q = tid & 3;
        const float mean = stat[r], rstd = stat[64 + r];
        const T* xrow = X + (long)(m0 + r) * C;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            const Vec16<T> xv = load16<T>(xrow + c0);
            for (int e = 0; e < VEC; ++e) {
                const float gw = Fs[r * LDF + c0 + e] * a.ln_w[c0 + e], xh = (xv.get(e) - mean) * rstd;
                s1 += gw;
                s2 += gw * xh;
            }
        }
        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
        s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
        s1 *= 1.0f / (float)C;
        s2 *= 1.0f / (float)C;
        T* dxrow = reinterpret_cast<T*>(a.dX) + (long)(m0 + r) * C;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            const Vec16<T> xv = load16<T>(xrow + c0);
            const Vec16<T> dy = load16<T>(dY + (long)(m0 + r) * C + c0);
            Vec16<T> o;
            for (int e = 0; e < VEC; ++e) {
                const float dxn = Fs[r * LDF + c0 + e], xh = (xv.get(e) - mean) * rstd;
                o.set(e, dy.get(e) + rstd * (dxn * a.ln_w[c0 + e] - s1 - xh * s2));
                Fs[r * LDF + c0 + e] = dxn * xh;          // for d(ln weight)
            }
            store16<T>(dxrow + c0, o);
        }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {                 // d(ln weight)[c] = sum_tok dxn * xhat
        float s = 0.f;
        for (int t = 0; t < 64; ++t) s += Fs[t * LDF + c];
        part[c] = s;
    }
}

// d_xn (fp32, [TOK][LDF] in LDS) -> d(LN bias) partials, LayerNorm backward + residual -> dx, d(LN weight) partials.  Shared by the
// second form's own epilogue and by the combine kernel of the hidden split.  `grp0` = index of the first partial group (of GS tokens).
template <class T, int C, int TOK, int NTHR, int TT, int RPP>
__device__ __forceinline__ void mlp_bwd_ln_epilogue(const MlpBwdDev& a, float* Fs, const float* stat, long m0, long grp0, int tid) {
    constexpr int VEC = Vec16<T>::N, NV = C / VEC, VPT = NV / 4, LDF = C + 4, GS = TOK < 64 ? TOK : 64, NG = TOK / GS;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* dY = reinterpret_cast<const T*>(a.dY);
    float* part = a.part + grp0 * 2 * C;
    for (int i = tid; i < NG * C; i += NTHR) {                      // d(ln bias)[c] = sum_tok dxn, per group of GS tokens
        const int g = i / C, c = i % C;
        float s = 0.f;
        for (int t = 0; t < GS; ++t) s += Fs[(g * GS + t) * LDF + c];
        part[(g * 2 + 1) * C + c] = s;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < TT; ++pass) {
        const int r = pass * RPP + (tid >> 2), q = tid & 3;
        const float mean = stat[r], rstd = stat[TOK + r];
        const T* xrow = X + (m0 + r) * C;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            const Vec16<T> xv = load16<T>(xrow + c0);
            for (int e = 0; e < VEC; ++e) {
                const float gw = Fs[r * LDF + c0 + e] * a.ln_w[c0 + e], xh = (xv.get(e) - mean) * rstd;
                s1 += gw;
                s2 += gw * xh;
            }
        }
        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
        s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
        s1 *= 1.0f / (float)C;
        s2 *= 1.0f / (float)C;
        T* dxrow = reinterpret_cast<T*>(a.dX) + (m0 + r) * C;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            const Vec16<T> xv = load16<T>(xrow + c0);
            const Vec16<T> dy = load16<T>(dY + (m0 + r) * C + c0);
            Vec16<T> o;
            for (int e = 0; e < VEC; ++e) {
                const float dxn = Fs[r * LDF + c0 + e], xh = (xv.get(e) - mean) * rstd;
                o.set(e, dy.get(e) + rstd * (dxn * a.ln_w[c0 + e] - s1 - xh * s2));
                Fs[r * LDF + c0 + e] = dxn * xh;          // for d(ln weight)
            }
            store16<T>(dxrow + c0, o);
        }
    }
    __syncthreads();
    for (int i = tid; i < NG * C; i += NTHR) {                // d(ln weight)[c] = sum_tok dxn * xhat
        const int g = i / C, c = i % C;
        float s = 0.f;
        for (int t = 0; t < GS; ++t) s += Fs[(g * GS + t) * LDF + c];
        part[(g * 2) * C + c] = s;
    }
}

// ---- second form ------------------------------------------------------------------------------------------------
// The first form is bound by LDS traffic and barriers (measured 140-170 TFLOP/s): one 16-token tile per wave means
// every weight fragment read from LDS feeds exactly one MFMA, the token fragments are re-read from LDS for each of
// the HP/32 chunks, the next chunk's weights are fetched only after the previous chunk is done, and the wave-private
// h / dpre staging goes through five workgroup barriers per chunk.  Here
//   * the wave's LN(x) and dm fragments live in REGISTERS for the whole chunk loop (the token tiles leave LDS),
//   * a wave owns TT 16-token tiles, so one weight fragment feeds TT MFMAs,
//   * all three weight slices of a chunk (fc1 rows, W2^T rows, W1^T columns) are staged in LDS once per workgroup
//     and the NEXT chunk's slices are already in flight in registers during the MFMAs,
//   * wave-private staging uses wave barriers: two workgroup barriers per chunk remain.
// NWV waves per workgroup, one (TT) 16-token tile each: 8 = two per SIMD (one wave's GELU / stores beside the other's MFMAs), 4 = the
// original form.  (A two-wave form, 32 tokens per workgroup, was measured slower at the latent level in round 4 -- every workgroup
// streams all weight slices through LDS by itself, twice the workgroups are twice that traffic -- and is gone: small launches split
// the HIDDEN dimension instead.)  The LayerNorm-parameter partials are per group of 64 tokens.
template <class T, int C, int TT, int NWV>
__global__ __launch_bounds__(64 * NWV) void gated_mlp_bwd2_kernel(MlpBwdDev a) {
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T), LDX = C + PAD, LDH = 64 + PAD, LDF = C + 4, VEC = Vec16<T>::N;
    constexpr int NCT = C / 16, NV = C / VEC, VPT = NV / 4;
    constexpr int TOK = 16 * TT * NWV, WT = 16 * TT, NTHR = 64 * NWV, GS = TOK < 64 ? TOK : 64, NG = TOK / GS;
    constexpr int NKC = C / TR::KCHUNK, NKH = 32 / TR::KCHUNK;
    constexpr size_t P0 = 2 * (size_t)TOK * LDX * sizeof(T);
    constexpr size_t P1 = ((size_t)96 * LDX + (size_t)C * LDH + NWV * WT * LDH) * sizeof(T);
    constexpr size_t P2 = (size_t)TOK * LDF * 4;
    constexpr size_t REG = ((P0 > P1 ? (P0 > P2 ? P0 : P2) : (P1 > P2 ? P1 : P2)) + 15) / 16 * 16;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    char* smem = reinterpret_cast<char*>(smem_v);
    T* Xs = reinterpret_cast<T*>(smem);                  // phase 0: [TOK][LDX] LN(x)
    T* Ds = Xs + TOK * LDX;                              //          [TOK][LDX] dm
    T* W1s = reinterpret_cast<T*>(smem);                 // loop: [64][LDX] fc1 rows: value 0..31 | gate 32..63
    T* W2Ts = W1s + 64 * LDX;                            //       [32][LDX] W2^T rows of the chunk
    T* W1Ts = W2Ts + 32 * LDX;                           //       [C][LDH]  W1^T columns: value 0..31 | gate 32..63
    T* Hs = W1Ts + C * LDH;                              //       [NWV][WT][LDH] per wave: h, then [dval | dgate]
    float* Fs = reinterpret_cast<float*>(smem);          // end:  [TOK][LDF] fp32 dxn
    float* stat = reinterpret_cast<float*>(smem + REG);  // mean[TOK], rstd[TOK]

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const long m0 = (long)blockIdx.x * TOK;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* DM = reinterpret_cast<const T*>(a.DM);
    const T* dY = reinterpret_cast<const T*>(a.dY);
    const int HP = a.HP;

    // ---- phase 0: LN(x) (also written to XN) and dm into LDS, 4 adjacent lanes per token; then into fragments ----
#pragma unroll
    for (int pass = 0; pass < TT; ++pass) {
        const int r = pass * 16 * NWV + (tid >> 2), q = tid & 3;
        const T* row = X + (m0 + r) * C;
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
        if (q == 0) { stat[r] = mean; stat[TOK + r] = rstd; }
        T* XN = reinterpret_cast<T*>(a.XN) + (m0 + r) * C;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            Vec16<T> o;
            for (int e = 0; e < VEC; ++e) o.set(e, (xv[i].get(e) - mean) * rstd * a.ln_w[c0 + e] + a.ln_b[c0 + e]);
            store16<T>(Xs + r * LDX + c0, o);
            if (blockIdx.y == 0) store16<T>(XN + c0, o);          // hidden split: every slab needs LN(x), the first one stores it
            Vec16<T> dmv;
            if (a.keep) {                             // DropPath backward fused: dm = keep[b] * dy, also kept for dW2 / db2
                const float kf = a.keep[(m0 + r) / a.rpb];
                const Vec16<T> dyv = load16<T>(dY + (m0 + r) * C + c0);
                for (int e = 0; e < VEC; ++e) dmv.set(e, kf * dyv.get(e));
                if (blockIdx.y == 0) store16<T>(const_cast<T*>(DM) + (m0 + r) * C + c0, dmv);
            } else {
                dmv = load16<T>(DM + (m0 + r) * C + c0);
            }
            store16<T>(Ds + r * LDX + c0, dmv);
        }
    }
    __syncthreads();
    frag_t bx[TT][NKC], bd[TT][NKC];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            bx[t][kc] = load_frag<T>(Xs, LDX, wv * WT + 16 * t, kc * TR::KCHUNK);
            bd[t][kc] = load_frag<T>(Ds, LDX, wv * WT + 16 * t, kc * TR::KCHUNK);
        }

    // ---- weight slices of one chunk: global -> registers (prefetch) -> LDS ----------------------------------------
    const T* W1 = reinterpret_cast<const T*>(a.W1);
    const T* W1T = reinterpret_cast<const T*>(a.W1T);
    const T* W2T = reinterpret_cast<const T*>(a.W2T);
    constexpr int VPR = C / VEC, VPH = 32 / VEC;
    constexpr int NW1 = 64 * VPR, NW2 = 32 * VPR, NWT = C * 2 * VPH, NVW = NW1 + NW2 + NWT, NPT = (NVW + NTHR - 1) / NTHR;
    Vec16<T> wreg[NPT];
    auto wload = [&](int j) {
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const int v = tid + NTHR * it;
            if (v < NW1) {
                const int r = v / VPR, c = (v % VPR) * VEC;
                wreg[it] = load16<T>(W1 + (long)(r < 32 ? j + r : HP + j + (r - 32)) * C + c);
            } else if (v < NW1 + NW2) {
                const int u = v - NW1, r = u / VPR, c = (u % VPR) * VEC;
                wreg[it] = load16<T>(W2T + (long)(j + r) * C + c);
            } else if (v < NVW) {
                const int u = v - NW1 - NW2, r = u / (2 * VPH), seg = (u / VPH) & 1, c = (u % VPH) * VEC;
                wreg[it] = load16<T>(W1T + (long)r * 2 * HP + seg * HP + j + c);
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
            } else if (v < NW1 + NW2) {
                const int u = v - NW1, r = u / VPR, c = (u % VPR) * VEC;
                store16<T>(W2Ts + r * LDX + c, wreg[it]);
            } else if (v < NVW) {
                const int u = v - NW1 - NW2, r = u / (2 * VPH), seg = (u / VPH) & 1, c = (u % VPH) * VEC;
                store16<T>(W1Ts + r * LDH + seg * 32 + c, wreg[it]);
            }
        }
    };
    // hidden split (small launches, see MlpBwdDev): this workgroup takes hidden columns [j0, j1); h and [dval | dgate] columns are
    // disjoint between slabs, d_xn is a sum over them
    const int hsp = a.hsplit > 1 ? a.hsplit : 1, j0 = (int)blockIdx.y * (HP / hsp), j1 = j0 + HP / hsp;
    wload(j0);

    T* Hw = Hs + wv * WT * LDH;
    T* Hout = reinterpret_cast<T*>(a.H);
    T* Pout = reinterpret_cast<T*>(a.DPRE);
    f32x4 out[TT][NCT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int i = 0; i < NCT; ++i) out[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int hr = (lane >> 4) * 4, tk = lane & 15;

    for (int j = j0; j < j1; j += 32) {
        __syncthreads();                 // every wave is done with the previous chunk's slices (and, first time, with Xs/Ds)
        wstore();
        __syncthreads();
        if (j + 32 < j1) wload(j + 32);  // in flight during the MFMAs below
        f32x4 pv[2][TT], pg[2][TT], pe[2][TT];
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int t = 0; t < TT; ++t) pv[f][t] = pg[f][t] = pe[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // PIPE (eight waves, C <= 128: registers to spare at two waves per SIMD): the six weight fragments of K chunk kc + 1 are
        // requested before the MFMAs of chunk kc and the scheduler is kept from sinking each read down to its use -- left alone it
        // serialises read -> wait -> MFMA with one or two reads in flight, the LDS latency exposed in front of most MFMAs.
        #ifndef MPHSIR_MLP_BWD_PIPE
#define MPHSIR_MLP_BWD_PIPE 1
#endif
        constexpr bool PIPE = MPHSIR_MLP_BWD_PIPE && NWV == 8 && TT == 1 && C <= 128 && sizeof(T) == 2;
        if constexpr (PIPE) {
            frag_t wq[2][6];
            auto fetch = [&](int kc) {
                const int kk = kc * TR::KCHUNK, b = kc & 1;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    wq[b][3 * f + 0] = load_frag<T>(W1s, LDX, 16 * f, kk);
                    wq[b][3 * f + 1] = load_frag<T>(W1s, LDX, 32 + 16 * f, kk);
                    wq[b][3 * f + 2] = load_frag<T>(W2Ts, LDX, 16 * f, kk);
                }
            };
            fetch(0);
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int b = kc & 1;
                if (kc + 1 < NKC) fetch(kc + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    mma(pv[f][0], wq[b][3 * f + 0], bx[0][kc]);
                    mma(pg[f][0], wq[b][3 * f + 1], bx[0][kc]);
                    mma(pe[f][0], wq[b][3 * f + 2], bd[0][kc]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int kk = kc * TR::KCHUNK;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const frag_t wv_ = load_frag<T>(W1s, LDX, 16 * f, kk), wg_ = load_frag<T>(W1s, LDX, 32 + 16 * f, kk);
                    const frag_t w2_ = load_frag<T>(W2Ts, LDX, 16 * f, kk);
#pragma unroll
                    for (int t = 0; t < TT; ++t) {
                        mma(pv[f][t], wv_, bx[t][kc]);
                        mma(pg[f][t], wg_, bx[t][kc]);
                        mma(pe[f][t], w2_, bd[t][kc]);
                    }
                }
            }
        }
        f32x4 hh[2][TT], dv[2][TT], dg[2][TT];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            float bv[4], bg[4];
            for (int r = 0; r < 4; ++r) { bv[r] = a.b1[j + 16 * f + hr + r]; bg[r] = a.b1[HP + j + 16 * f + hr + r]; }
#pragma unroll
            for (int t = 0; t < TT; ++t)
                for (int r = 0; r < 4; ++r) {
                    const float va = pv[f][t][r] + bv[r], ga = pg[f][t][r] + bg[r];
                    float ea, da;
                    Math<T>::gelu_pair(ga, ea, da);
                    hh[f][t][r] = va * ea;
                    dv[f][t][r] = pe[f][t][r] * ea;
                    dg[f][t][r] = pe[f][t][r] * va * da;
                }
        }
        // h chunk -> HBM through the wave's staging rows (whole 64-byte row segments per token); not when the parameter gradients
        // come from mphsir_gated_mlp_wgrad (H == DPRE == NULL)
        if (Hout) {
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                store4<T>(Hw + (16 * t + tk) * LDH + hr, hh[0][t]);
                store4<T>(Hw + (16 * t + tk) * LDH + 16 + hr, hh[1][t]);
            }
            wave_barrier();
            for (int i = lane; i < WT * VPH; i += 64) {
                const int t = i / VPH, c = (i % VPH) * VEC;
                store16<T>(Hout + (m0 + wv * WT + t) * HP + j + c, load16<T>(Hw + t * LDH + c));
            }
            wave_barrier();
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            store4<T>(Hw + (16 * t + tk) * LDH + hr, dv[0][t]);
            store4<T>(Hw + (16 * t + tk) * LDH + 16 + hr, dv[1][t]);
            store4<T>(Hw + (16 * t + tk) * LDH + 32 + hr, dg[0][t]);
            store4<T>(Hw + (16 * t + tk) * LDH + 48 + hr, dg[1][t]);
        }
        wave_barrier();
        if (Pout)
            for (int i = lane; i < WT * 2 * VPH; i += 64) {
                const int t = i / (2 * VPH), seg = (i / VPH) & 1, c = (i % VPH) * VEC;
                store16<T>(Pout + (m0 + wv * WT + t) * 2 * HP + seg * HP + j + c, load16<T>(Hw + t * LDH + seg * 32 + c));
            }
        // dxn[c][tok] += W1T[c][j..j+31] * dval + W1T[c][HP+j..] * dgate
        if constexpr (PIPE) {
            // the same for d_xn: W1^T fragments in groups of four, the next group requested before the current group's MFMAs
            constexpr int GW = NCT % 4 == 0 ? 4 : 2, NG = 2 * NKH * NCT / GW;
            static_assert(NCT % GW == 0, "groups of W1^T fragments");
            frag_t bh[2 * NKH], wt[2][GW];
#pragma unroll
            for (int kh = 0; kh < 2 * NKH; ++kh) bh[kh] = load_frag<T>(Hw, LDH, 0, kh * TR::KCHUNK);
            auto fetch = [&](int g) {
                const int kh = g / (NCT / GW), c0 = (g % (NCT / GW)) * GW;
#pragma unroll
                for (int i = 0; i < GW; ++i) wt[g & 1][i] = load_frag<T>(W1Ts, LDH, (c0 + i) * 16, kh * TR::KCHUNK);
            };
            fetch(0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int kh = g / (NCT / GW), c0 = (g % (NCT / GW)) * GW;
                if (g + 1 < NG) fetch(g + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GW; ++i) mma(out[0][c0 + i], wt[g & 1][i], bh[kh]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int kh = 0; kh < 2 * NKH; ++kh) {
                frag_t bh[TT];
#pragma unroll
                for (int t = 0; t < TT; ++t) bh[t] = load_frag<T>(Hw, LDH, 16 * t, kh * TR::KCHUNK);
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const frag_t wt_ = load_frag<T>(W1Ts, LDH, ct * 16, kh * TR::KCHUNK);
#pragma unroll
                    for (int t = 0; t < TT; ++t) mma(out[t][ct], wt_, bh[t]);
                }
            }
        }
        wave_barrier();
    }
    if (hsp > 1) {           // this slab's share of d_xn: fp32, 4 consecutive channels of a token per lane
        float* Dp = a.dxn_part + ((long)blockIdx.y * a.M + m0) * C;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const int tok = wv * WT + 16 * t + (lane & 15), cr = (lane >> 4) * 4;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(Dp + (long)tok * C + ct * 16 + cr) = out[t][ct];
        }
        return;
    }
    __syncthreads();

    // ---- dxn -> fp32 LDS stage; LayerNorm backward; dx = dy + ...; parameter-gradient partials per 64 tokens -----
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const int tok = wv * WT + 16 * t + (lane & 15), cr = (lane >> 4) * 4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) *reinterpret_cast<f32x4*>(Fs + tok * LDF + ct * 16 + cr) = out[t][ct];
    }
    __syncthreads();
    mlp_bwd_ln_epilogue<T, C, TOK, NTHR, TT, 16 * NWV>(a, Fs, stat, m0, (long)blockIdx.x * NG, tid);
}

// Hidden split, second launch: d_xn = the slabs' partials summed in order (deterministic) -> the same LayerNorm-backward epilogue,
// one workgroup per 64 tokens.
template <class T, int C>
__global__ __launch_bounds__(256) void mlp_bwd_combine_kernel(MlpBwdDev a) {
    constexpr int VEC = Vec16<T>::N, NV = C / VEC, VPT = NV / 4, LDF = C + 4;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* Fs = reinterpret_cast<float*>(smem_v);        // [64][LDF]
    float* stat = Fs + 64 * LDF;                         // mean[64], rstd[64]
    const int tid = threadIdx.x;
    const long m0 = (long)blockIdx.x * 64;
    const T* X = reinterpret_cast<const T*>(a.X);
    {
        const int r = tid >> 2, q = tid & 3;
        const T* row = X + (m0 + r) * C;
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
            for (int e = 0; e < VEC; ++e) { const float d = xv[i].get(e) - mean; d2 += d * d; }
        d2 += __shfl_xor(d2, 1);
        d2 += __shfl_xor(d2, 2);
        if (q == 0) { stat[r] = mean; stat[64 + r] = rsqrtf(d2 / (float)C + 1e-5f); }
    }
    for (int idx = tid; idx < 64 * (C / 4); idx += 256) {
        const int t = idx / (C / 4), c = (idx % (C / 4)) * 4;
        f32x4 acc = *reinterpret_cast<const f32x4*>(a.dxn_part + (m0 + t) * C + c);
        for (int sp = 1; sp < a.hsplit; ++sp) acc += *reinterpret_cast<const f32x4*>(a.dxn_part + ((long)sp * a.M + m0 + t) * C + c);
        *reinterpret_cast<f32x4*>(Fs + t * LDF + c) = acc;
    }
    __syncthreads();
    mlp_bwd_ln_epilogue<T, C, 64, 256, 1, 64>(a, Fs, stat, m0, (long)blockIdx.x, tid);
}

template <class T, int C, int TT, int NWV = 4>
constexpr size_t mlp_bwd2_lds() {
    constexpr size_t PAD = LDS_PAD_BYTES / sizeof(T), LDX = C + PAD, LDH = 64 + PAD, LDF = C + 4, TOK = 16 * TT * NWV, WT = 16 * TT;
    constexpr size_t P0 = 2 * TOK * LDX * sizeof(T), P1 = (96 * LDX + C * LDH + NWV * WT * LDH) * sizeof(T), P2 = TOK * LDF * 4;
    constexpr size_t REG = ((P0 > P1 ? (P0 > P2 ? P0 : P2) : (P1 > P2 ? P1 : P2)) + 15) / 16 * 16;
    return REG + 2 * TOK * sizeof(float);
}

template <class T, int C, int TT, int NWV = 4>
constexpr bool mlp_bwd2_fits() { return mlp_bwd2_lds<T, C, TT, NWV>() <= 160 * 1024 && (C / Vec16<T>::N) % 4 == 0 && (NWV == 4 || sizeof(T) == 2); }

template <class T, int C, int TT, int NWV = 4>
static int launch_mlp_bwd2(const MlpBwdDev& d, hipStream_t s) {
    if constexpr (mlp_bwd2_fits<T, C, TT, NWV>()) {
        constexpr size_t lds = mlp_bwd2_lds<T, C, TT, NWV>();
        allow_big_lds(gated_mlp_bwd2_kernel<T, C, TT, NWV>, lds);
        const int hsp = d.hsplit > 1 ? d.hsplit : 1;
        MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP_BWD, (gated_mlp_bwd2_kernel<T, C, TT, NWV>), dim3(d.M / (16 * TT * NWV), hsp), dim3(64 * NWV), lds, s, d);
        if (hsp > 1) {
            const size_t cl = (64 * (size_t)(C + 4) + 128) * sizeof(float);
            allow_big_lds(mlp_bwd_combine_kernel<T, C>, cl);
            MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP_BWD, (mlp_bwd_combine_kernel<T, C>), dim3(d.M / 64), dim3(256), cl, s, d);
        }
    }
    return MPHSIR_OK;
}

// variant: 0 = choose (second form when it fits LDS; eight waves with one tile each once that still leaves a workgroup per
// CU), 1 = first form, 2 / 3 = second form, four waves with one / two 16-token tiles per wave, 4 = second form, eight waves,
template <class T, int C>
static int launch_mlp_bwd(const MlpBwdDev& d, int variant, hipStream_t s) {
    if (d.hsplit > 1) {          // the hidden split is built on the four-wave second form
        if constexpr (mlp_bwd2_fits<T, C, 1>()) return launch_mlp_bwd2<T, C, 1>(d, s);
        set_error("gated_mlp_bwd: the hidden split needs the second form, which does not fit C=%d in this element type", C);
        return MPHSIR_EINVAL;
    }
    if (variant != 1) {
        if constexpr (mlp_bwd2_fits<T, C, 1, 8>()) {
            // C = 192 (remote-sensing dec1 / refinement): the eight-wave form spills 11 registers there and is still 1.7x the four-wave
            // form (M = 65536: 238 -> 141 us, tools/bench/bench_mlp_bwd.py); at 16384 tokens the two are level
            if (d.M % 128 == 0 && (variant == 4 || (variant == 0 && C <= 192 && d.M / 128 >= 256))) return launch_mlp_bwd2<T, C, 1, 8>(d, s);
        }
        if constexpr (mlp_bwd2_fits<T, C, 2>()) {
            if (d.M % 128 == 0 && variant == 3) return launch_mlp_bwd2<T, C, 2>(d, s);
        }
        if constexpr (mlp_bwd2_fits<T, C, 1>()) {
            if (variant == 0 || variant == 2) return launch_mlp_bwd2<T, C, 1>(d, s);
        }
    }
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr size_t base = (2 * 64 * (C + PAD) + 4 * 16 * (64 + PAD)) * sizeof(T) + 128 * sizeof(float);
    constexpr size_t staged = base + (64 * (size_t)(C + PAD) + (size_t)C * (64 + PAD)) * sizeof(T);
    if constexpr (staged <= 160 * 1024) {
        allow_big_lds(gated_mlp_bwd_kernel<T, C, true>, staged);
        MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP_BWD, (gated_mlp_bwd_kernel<T, C, true>), dim3(d.M / 64), dim3(256), staged, s, d);
    } else if constexpr (base > 160 * 1024) {
        constexpr size_t gx = 64 * (size_t)(C + 4) * 4 + 4 * 16 * (64 + PAD) * sizeof(T) + 128 * sizeof(float);
        static_assert(gx <= 160 * 1024, "gated_mlp_bwd: width does not fit LDS");
        allow_big_lds(gated_mlp_bwd_kernel<T, C, false, true>, gx);
        MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP_BWD, (gated_mlp_bwd_kernel<T, C, false, true>), dim3(d.M / 64), dim3(256), gx, s, d);
    } else {
        allow_big_lds(gated_mlp_bwd_kernel<T, C, false>, base);
        MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP_BWD, (gated_mlp_bwd_kernel<T, C, false>), dim3(d.M / 64), dim3(256), base, s, d);
    }
    return MPHSIR_OK;
}

template <class T>
static int dispatch_mlp_bwd(const MlpBwdDev& d, int C, int variant, hipStream_t s) {
    switch (C) {
        case 32: return launch_mlp_bwd<T, 32>(d, variant, s);
        case 64: return launch_mlp_bwd<T, 64>(d, variant, s);
        case 96: return launch_mlp_bwd<T, 96>(d, variant, s);
        case 128: return launch_mlp_bwd<T, 128>(d, variant, s);
        case 192: return launch_mlp_bwd<T, 192>(d, variant, s);
        case 256: return launch_mlp_bwd<T, 256>(d, variant, s);
        case 384: return launch_mlp_bwd<T, 384>(d, variant, s);
    }
    set_error("gated_mlp_bwd: C=%d not instantiated", C);
    return MPHSIR_EINVAL;
}

}  // namespace mphsir

extern "C" int mphsir_gated_mlp_bwd(const mphsir_mlp_bwd_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "gated_mlp_bwd");
    MPHSIR_REQUIRE(a && a->X && a->dY && a->DM && a->ln_w && a->ln_b && a->W1 && a->b1 && a->W1T && a->W2T && a->dX && a->XN &&
                       a->part && (!a->H == !a->DPRE), "gated_mlp_bwd: null pointer (H and DPRE: both or neither)");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "gated_mlp_bwd: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(a->M > 0 && a->M % 64 == 0 && a->HP > 0 && a->HP % 32 == 0, "gated_mlp_bwd: M %% 64 and HP %% 32 must be 0");
    MPHSIR_REQUIRE(aligned16(a->X) && aligned16(a->dY) && aligned16(a->DM) && aligned16(a->dX) && aligned16(a->XN) && aligned16(a->H) &&
                       aligned16(a->DPRE) && aligned16(a->W1) && aligned16(a->W1T) && aligned16(a->W2T), "gated_mlp_bwd: 16-byte alignment required");
    MlpBwdDev d{a->X, a->dY, a->DM, a->ln_w, a->ln_b, a->W1, a->b1, a->W1T, a->W2T, a->dX, a->XN, a->H, a->DPRE, a->part,
                (int)a->M, a->HP, a->keep, (long)a->rows_per_batch, a->hsplit, a->dxn_part};
    if (a->hsplit > 1)
        MPHSIR_REQUIRE(a->dxn_part && aligned16(a->dxn_part) && a->HP % (32 * a->hsplit) == 0 && (a->variant == 0 || a->variant == 2),
                       "gated_mlp_bwd: hsplit needs a workspace dxn_part [hsplit][M][C] fp32, HP %% (32 hsplit) == 0 and variant 0 or 2");
    MPHSIR_REQUIRE(!a->keep || (a->rows_per_batch > 0 && a->M % a->rows_per_batch == 0), "gated_mlp_bwd: keep needs rows_per_batch dividing M");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MPHSIR_REQUIRE(a->variant >= 0 && a->variant <= 4, "gated_mlp_bwd: variant must be 0..4");
    return MPHSIR_DISPATCH_T(dtype, (dispatch_mlp_bwd<T_>(d, a->C, a->variant, s)));
}
