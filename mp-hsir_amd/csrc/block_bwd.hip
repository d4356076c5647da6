// Backward kernels of the first residual branch of a PGSSTB block (window attention side).
//
//   combine_bwd   backward of gemm_tok epi 2,  y = x + keep*(sa*gate[win] + out):
//                 d_out = keep*dy, d_sa = d_out*gate[win], d_gate[win] = sum_tok d_out*sa   (ref :715-718, :153)
//   win_attn_bwd  backward of the 8x8 window attention core (ref Spatial_Attention.forward :193-218):
//                 per window and head recompute q,k,v and the probabilities from LN(x); dO = d_sa W_proj;
//                 dP^T = V dO^T; dS = P o (dP - rowsum(dP o P)); dQ = scale dS K, dK = dS^T Q, dV = P^T dO.
//                 Writes d[q|k|v] (window-token order), LN(x) (same order) and the total d_sa (image order),
//                 so that every parameter gradient is a plain token-reduction GEMM / column sum done by
//                 the caller, plus per-window partials of the relative-position-bias gradient.
//   ln_bwd_win    d_x = d_res + LayerNorm_backward(d_xn) with d_xn in window-token order (un-shift /
//                 un-window by address arithmetic), plus per-window partials of d(norm1 weight/bias).
// All reductions are ordered (no atomics): results are bitwise reproducible.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct WinGeom {
    int B, H, W, shift;
};

__device__ __forceinline__ long win_pixel(const WinGeom& g, int blk, int t) {
    const int nwx = g.W >> 3, nW = (g.H >> 3) * nwx;
    const int b = blk / nW, wi = blk % nW, wy = wi / nwx, wx = wi % nwx;
    const int ys = wy * 8 + (t >> 3), xs = wx * 8 + (t & 7);
    return ((long)b * g.H + (ys + g.shift) % g.H) * g.W + (xs + g.shift) % g.W;
}

// ---------------------------------------------------------------------------------------------------
struct CombBwdDev {
    const void* dY; const void* SA; const float* gate; const float* keep;
    void* dOut;      // optional: keep*dy (written only when keep != NULL)
    void* dSA;       // d_out * gate[win]
    float* dgate;    // [B*nW][C]
    WinGeom g; int C;
};

template <class T>
__global__ __launch_bounds__(256) void combine_bwd_kernel(CombBwdDev a) {
    // one pass over the window's 64 x C tile of dy and sa (16-byte vectors): a thread owns one channel vector and walks
    // every `groups`-th token, writing d_out / d_sa and accumulating dy*sa per channel; the `groups` partial sums per
    // channel meet in LDS in a fixed order.  (The first form re-read both tiles with 2-byte loads for the sums.)
    constexpr int VEC = Vec16<T>::N;
    __shared__ float red[256 * VEC];                 // [groups][C] partial sums (groups * C <= 256 * VEC)
    const int C = a.C, tid = threadIdx.x, nv = C / VEC;
    const int groups = 256 / nv < 64 ? 256 / nv : 64, grp = tid / nv, cv = tid % nv, c0 = cv * VEC;
    const int nW = (a.g.H >> 3) * (a.g.W >> 3), b = blockIdx.x / nW;
    const float kf = a.keep ? a.keep[b] : 1.f;
    const T* dY = reinterpret_cast<const T*>(a.dY);
    const T* SA = reinterpret_cast<const T*>(a.SA);
    T* dOut = reinterpret_cast<T*>(a.dOut);
    T* dSA = reinterpret_cast<T*>(a.dSA);
    const float* g = a.gate + (long)blockIdx.x * C;
    float acc[VEC];
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (grp < groups) {
        float gv[VEC];
        for (int e = 0; e < VEC; ++e) gv[e] = g[c0 + e];
        for (int t = grp; t < 64; t += groups) {
            const long p = win_pixel(a.g, blockIdx.x, t) * C + c0;
            const Vec16<T> dy = load16<T>(dY + p), sa = load16<T>(SA + p);
            Vec16<T> o, s;
            for (int e = 0; e < VEC; ++e) {
                const float d = to_f32(from_f32<T>(kf * dy.get(e)));     // rounded exactly as the stored d_out
                o.set(e, d);
                s.set(e, d * gv[e]);
                acc[e] += d * sa.get(e);
            }
            if (a.keep && dOut) store16<T>(dOut + p, o);
            store16<T>(dSA + p, s);
        }
        for (int e = 0; e < VEC; ++e) red[grp * C + c0 + e] = acc[e];
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float sum = 0.f;
        for (int gi = 0; gi < groups; ++gi) sum += red[gi * C + c];
        a.dgate[(long)blockIdx.x * C + c] = sum;
    }
}

// ---------------------------------------------------------------------------------------------------
struct WinBwdDev {
    const void* X; const void* dSA; const float* dmu;
    const float* ln_w; const float* ln_b;
    const void* Wqkv; const float* bqkv; const float* rpb;
    const void* WprojT;                 // [C][C]: WprojT[ci][co] = proj.weight[co][ci]
    void* dQKV;                         // [B*nW*64][3C]  window-token order
    void* XNw;                          // [B*nW*64][C]   LN(x), window-token order
    void* dSAt;                         // [B*nW*64][C]   total d_sa (adds dmu/64), window-token order
    float* drpb;                        // [B*nW][225][HEADS]
    WinGeom g;
    int hsplit;                         // the heads of a window are dealt to hsplit workgroups (grid.y): small launches (the latent level has
                                        // 128 windows for 256 CUs) get one workgroup per (window, head group) instead of half a chip idle
};

// LDS plan.  Per head the kernel holds four row-major [64 tok][HDP] tiles (q, k, v, dO_h) and the two 64x64 matrices
// P and dS, each stored ONCE as [query][key]: every product that needs an operand "the other way round" (dQ = dS K,
// dK = dS^T Q, dV = P^T dO) reads it through load_frag_tr (ds_read_b64_tr_b16 for bf16, four strided reads for f32),
// so there are no transposed copies and no scalar transposing stores.  The LN(x) tile (the B operand of the q/k/v
// recompute, read 3x per head) stays in LDS when XL; otherwise its fragments come back from the XNw rows this
// workgroup has just written (L1/L2 hits).  The total d_sa (read once per head) always comes from the dSAt rows.
// With that the footprint no longer grows with 2*C: every shipped width fits in both dtypes.
template <class T, int C, int HD, bool XL> struct WinBwdCfg {
    static constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    static constexpr int HEADS = C / HD;
    static constexpr int KC = ElemTraits<T>::KCHUNK;
    static constexpr int HDP = (HD + KC - 1) / KC * KC;
    static constexpr int LDX = C + PAD;
    static constexpr int LDQ = HDP + PAD;        // row-major [64][HDP]
    static constexpr int LDT = 64 + PAD;         // the 64x64 tiles
    static constexpr bool ALIAS = HDP >= 64;     // P reuses the V tile (v is dead once dP is formed)
    static constexpr size_t XS = XL ? 64 * LDX : 0, QS = 64 * LDQ, SS = 64 * LDT;
    static constexpr size_t T_ELEMS = XS + 4 * QS + (ALIAS ? 1 : 2) * SS;
    static constexpr size_t BYTES = T_ELEMS * sizeof(T) + (225 + 64) * 4;
    static constexpr bool FITS = BYTES <= 160 * 1024;
    // weight stage of phase (a): SR rows x C of q_h / k_h / v_h / proj rows, placed in the dS tile (free until the softmax
    // backward) -- no LDS growth.  SR = the largest of HD, 32, 16 that divides HD and fits; none: fragments straight from L2.
    static constexpr int LDWS = C + PAD;
    static constexpr int SR = (size_t)HD * LDWS <= SS ? HD : (HD % 32 == 0 && (size_t)32 * LDWS <= SS) ? 32 : ((size_t)16 * LDWS <= SS ? 16 : 0);
    static constexpr bool STAGE = SR > 0;
};

// X tile in LDS whenever the whole footprint then still allows two workgroups per CU (80 KB each); beyond that the
// tile is dropped (occupancy beats the L2 re-reads: measured on MI355X, tools/bench_win.py).
template <class T, int C, int HD> struct WinBwdPick {
    static constexpr bool XL = WinBwdCfg<T, C, HD, true>::BYTES <= 80 * 1024;
};

// XR (C <= 128, 16-bit): the LN(x) fragments of the 32 tokens a wave multiplies in every q/k/v unit (half `wv & 1` of the
// window) live in REGISTERS for the whole kernel -- the tile passes through the (still unused) q|k tiles once -- so the
// footprint is the one without the X tile (52 KB at C = 128: three workgroups per CU instead of two) at no re-read.
template <class T, int C, int HD, bool XL, bool XR = false>
__global__ __launch_bounds__(256, XR ? 3 : 1) void win_attn_bwd_kernel(WinBwdDev a) {
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    typedef WinBwdCfg<T, C, HD, XL> CF;
    static_assert(!XR || (!XL && CF::HDP == HD && CF::STAGE && 64 * (C + CF::PAD) <= 4 * CF::QS), "XR: no X tile, no K padding, staged weights");
    constexpr int VEC = Vec16<T>::N;
    constexpr int TPW = HD / 16;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* Xs = reinterpret_cast<T*>(smem_v);        // LN(x)            [64][LDX]   (XL only)
    T* Qr = Xs + CF::XS;                         // q (scaled)       [64][LDQ]
    T* Kr = Qr + CF::QS;                         // k                [64][LDQ]
    T* Vr = Kr + CF::QS;                         // v                [64][LDQ]
    T* Or = Vr + CF::QS;                         // dO_h             [64][LDQ]
    T* dS = Or + CF::QS;                         // dS   [q][key]    [64][LDT]
    T* Ps = CF::ALIAS ? Vr : dS + CF::SS;        // P    [q][key]    [64][LDT]
    float* rpbs = reinterpret_cast<float*>(Xs + CF::T_ELEMS);
    int* reg = reinterpret_cast<int*>(rpbs + 225);

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* dSA = reinterpret_cast<const T*>(a.dSA);
    const long row0 = (long)blockIdx.x * 64;
    const T* Xg = reinterpret_cast<const T*>(a.XNw) + row0 * C;      // this window's rows of the two side outputs,
    const T* Dg = reinterpret_cast<const T*>(a.dSAt) + row0 * C;     // read back as MFMA operands after the barrier

    // ---- LN(x) and the total d_sa: both side outputs (window-token order); LN(x) also into LDS when XL ----------
    {
        constexpr int NV = C / VEC, VPT = NV / 4;
        const int t = tid >> 2, q = tid & 3;
        const long pix = win_pixel(a.g, blockIdx.x, t);
        const T* row = X + pix * C;
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
        T* xnw = reinterpret_cast<T*>(a.XNw) + (row0 + t) * C;
        T* dst = reinterpret_cast<T*>(a.dSAt) + (row0 + t) * C;
        const float* dmu = a.dmu + (long)blockIdx.x * C;
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c0 = (q + 4 * i) * VEC;
            Vec16<T> o;
            for (int e = 0; e < VEC; ++e) o.set(e, (xv[i].get(e) - mean) * rstd * a.ln_w[c0 + e] + a.ln_b[c0 + e]);
            if (XL) store16<T>(Xs + t * CF::LDX + c0, o);
            if (XR) store16<T>(Qr + t * CF::LDX + c0, o);      // transit through the q|k tiles (written only from phase (a) on)
            store16<T>(xnw + c0, o);
            Vec16<T> d = load16<T>(dSA + pix * C + c0);
            for (int e = 0; e < VEC; ++e) d.set(e, d.get(e) + dmu[c0 + e] * (1.0f / 64.0f));
            store16<T>(dst + c0, d);
        }
        if (tid < 64) {
            const int nwx = a.g.W >> 3, nW = (a.g.H >> 3) * nwx, wi = blockIdx.x % nW;
            const int ys = (wi / nwx) * 8 + (tid >> 3), xs = (wi % nwx) * 8 + (tid & 7);
            const int ry = (ys >= a.g.H - 8) + (ys >= a.g.H - 4), rx = (xs >= a.g.W - 8) + (xs >= a.g.W - 4);
            reg[tid] = a.g.shift ? 3 * ry + rx : 0;
        }
        if (CF::HDP != HD)   // zero K-padding columns of the four row-major tiles once
            for (int i = tid; i < 4 * 64 * (CF::HDP - HD); i += 256) {
                const int rr = i / (CF::HDP - HD), cc = HD + i % (CF::HDP - HD);
                Qr[rr * CF::LDQ + cc] = from_f32<T>(0.f);
            }
    }

    const T* Wqkv = reinterpret_cast<const T*>(a.Wqkv);
    const T* WpT = reinterpret_cast<const T*>(a.WprojT);
    T* dQKV = reinterpret_cast<T*>(a.dQKV);
    const float scale = rsqrtf((float)HD);

    // weight stage of phase (a): SR rows x C, 16-byte vectors over all 256 threads, register prefetch one stage ahead
    constexpr int WSR = CF::STAGE ? CF::SR : 16;
    constexpr int WVT = WSR * (C / VEC), NWV = (WVT + 255) / 256;
    Vec16<T> wpre[NWV];
    auto wload = [&](int h, int sub) __attribute__((always_inline)) {
        const int which = sub / (HD / WSR), r0 = (sub % (HD / WSR)) * WSR;
        const T* Wsrc = (which < 3 ? Wqkv + (long)(which * C) * C : WpT) + (long)(h * HD + r0) * C;
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < WVT) wpre[i] = load16<T>(Wsrc + (long)(idx / (C / VEC)) * C + (idx % (C / VEC)) * VEC);
        }
    };
    auto wstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < WVT) store16<T>(dS + (idx / (C / VEC)) * CF::LDWS + (idx % (C / VEC)) * VEC, wpre[i]);
        }
    };
    // heads h_lo .. h_hi-1 of this window (every workgroup of the window writes the same XNw / dSAt rows: identical values)
    const int hpb = CF::HEADS / a.hsplit, h_lo = blockIdx.y * hpb, h_hi = h_lo + hpb;
    if (CF::STAGE) wload(h_lo, 0);
    constexpr int NKX = XR ? C / TR::KCHUNK : 1;
    frag_t xf[2][NKX];

    for (int h = h_lo; h < h_hi; ++h) {
        __syncthreads();      // first head: the side outputs / X tile are visible to the whole workgroup; later: tiles free
        if constexpr (XR) {
            if (h == h_lo) {  // (the first stage's barriers order these reads before the first q rows are written)
#pragma unroll
                for (int kc = 0; kc < NKX; ++kc) {
                    xf[0][kc] = load_frag<T>(Qr, CF::LDX, (wv & 1) * 32, kc * TR::KCHUNK);
                    xf[1][kc] = load_frag<T>(Qr, CF::LDX, (wv & 1) * 32 + 16, kc * TR::KCHUNK);
                }
            }
        }
        if (CF::ALIAS && CF::HDP != HD)   // P overwrote the zero padding of v
            for (int i = tid; i < 64 * (CF::HDP - HD); i += 256) {
                const int rr = i / (CF::HDP - HD), cc = HD + i % (CF::HDP - HD);
                Vr[rr * CF::LDQ + cc] = from_f32<T>(0.f);
            }
        if (tid < 225) rpbs[tid] = a.rpb[tid * CF::HEADS + h];
        if constexpr (CF::STAGE) {
        // ---- (a) recompute q,k,v and dO_h = d_sa W_proj[:, head] (row-major [tok][hd]) -------------------------------
        // The weight rows go through LDS (the dS tile, free in this phase) SR rows at a time: all 256 threads fetch the NEXT
        // stage from L2 (coalesced, all in flight) while the current one is multiplied -- as in win_attn.hip, where per-unit
        // fragment fetches from L2 were 58 % of a head's time.  unit u = (16-channel tile of the stage, half of the tokens).
        constexpr int NKC = C / TR::KCHUNK, SR = CF::SR, SPP = HD / SR, NSUB = 4 * SPP, TPS = SR / 16;
        T* Wst = dS;                                 // [SR][LDWS]
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int which = sub / SPP, r0 = (sub % SPP) * SR;     // 0..2: q, k, v rows of Wqkv; 3: rows of WprojT
            if (sub > 0) __syncthreads();            // the previous stage's fragments are read
            wstore();
            if (sub + 1 < NSUB) wload(h, sub + 1);
            else if (h + 1 < h_hi) wload(h + 1, 0);
            __syncthreads();
            for (int u = wv; u < 2 * TPS; u += 4) {
                const int ctl = u >> 1, cti = r0 / 16 + ctl, th = u & 1;
                f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
                const int wrow = (which < 3 ? which * C : 0) + h * HD + cti * 16;
                if (XR && which < 3) {                   // register operand (th == wv & 1 for every unit of a wave)
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        const frag_t w = load_frag<T>(Wst, CF::LDWS, ctl * 16, kc * TR::KCHUNK);
                        mma(c0, w, xf[0][XR ? kc : 0]);
                        mma(c1, w, xf[1][XR ? kc : 0]);
                    }
                } else if (XL && which < 3) {            // wave-uniform: LDS operand
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        const frag_t w = load_frag<T>(Wst, CF::LDWS, ctl * 16, kc * TR::KCHUNK);
                        mma(c0, w, load_frag<T>(Xs, CF::LDX, th * 32, kc * TR::KCHUNK));
                        mma(c1, w, load_frag<T>(Xs, CF::LDX, th * 32 + 16, kc * TR::KCHUNK));
                    }
                } else {                                 // global operand (rows written above by this workgroup)
                    const T* Bsrc = which < 3 ? Xg : Dg;
                    frag_t b0[NKC], b1[NKC];
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        b0[kc] = load_frag<T>(Bsrc, C, th * 32, kc * TR::KCHUNK);
                        b1[kc] = load_frag<T>(Bsrc, C, th * 32 + 16, kc * TR::KCHUNK);
                    }
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        const frag_t w = load_frag<T>(Wst, CF::LDWS, ctl * 16, kc * TR::KCHUNK);
                        mma(c0, w, b0[kc]);
                        mma(c1, w, b1[kc]);
                    }
                }
                const int cr = cti * 16 + (lane >> 4) * 4;
                if (which < 3) {
                    const float sc = which == 0 ? scale : 1.f;
                    for (int r = 0; r < 4; ++r) {
                        const float bb = a.bqkv[wrow + (lane >> 4) * 4 + r];
                        c0[r] = (c0[r] + bb) * sc;
                        c1[r] = (c1[r] + bb) * sc;
                    }
                }
                T* rowm = which == 0 ? Qr : which == 1 ? Kr : which == 2 ? Vr : Or;
                store4<T>(rowm + (th * 32 + (lane & 15)) * CF::LDQ + cr, c0);
                store4<T>(rowm + (th * 32 + 16 + (lane & 15)) * CF::LDQ + cr, c1);
            }
        }
        __syncthreads();

        } else {
        // ---- (a) recompute q,k,v and dO_h = d_sa W_proj[:, head] (row-major [tok][hd]) -------------------------------
        // unit u = (16-channel tile of q|k|v|dO, half of the 64 tokens).  The weight fragments come straight from L2 and
        // feed two MFMAs each: the next unit's fragments are loaded into a second register set during the current unit.
        constexpr int NUNITS = 4 * TPW * 2, NKC = C / TR::KCHUNK, UPW = NUNITS / 4;      // units per wave (NUNITS % 4 == 0)
        {
            frag_t wq[2][NKC];
            {
                const int ct = wv >> 1, which = ct / TPW, cti = ct % TPW;
                const int wrow = (which < 3 ? which * C : 0) + h * HD + cti * 16;
                const T* Wsrc = which < 3 ? Wqkv : WpT;
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) wq[0][kc] = load_frag<T>(Wsrc, C, wrow, kc * TR::KCHUNK);
            }
#pragma unroll
            for (int i = 0; i < UPW; ++i) {              // fully unrolled: the register set index i & 1 is static
                const int u = wv + 4 * i;
                if (i + 1 < UPW) {
                    const int ct = (u + 4) >> 1, which = ct / TPW, cti = ct % TPW;
                    const int wrow = (which < 3 ? which * C : 0) + h * HD + cti * 16;
                    const T* Wsrc = which < 3 ? Wqkv : WpT;
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) wq[(i + 1) & 1][kc] = load_frag<T>(Wsrc, C, wrow, kc * TR::KCHUNK);
                }
                const int ct = u >> 1, th = u & 1, which = ct / TPW, cti = ct % TPW;
                f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
                const int wrow = (which < 3 ? which * C : 0) + h * HD + cti * 16;
                if (XL && which < 3) {                   // wave-uniform: LDS operand
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        mma(c0, wq[i & 1][kc], load_frag<T>(Xs, CF::LDX, th * 32, kc * TR::KCHUNK));
                        mma(c1, wq[i & 1][kc], load_frag<T>(Xs, CF::LDX, th * 32 + 16, kc * TR::KCHUNK));
                    }
                } else {                                 // global operand (rows written above by this workgroup)
                    const T* Bsrc = which < 3 ? Xg : Dg;
                    frag_t b0[NKC], b1[NKC];
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        b0[kc] = load_frag<T>(Bsrc, C, th * 32, kc * TR::KCHUNK);
                        b1[kc] = load_frag<T>(Bsrc, C, th * 32 + 16, kc * TR::KCHUNK);
                    }
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        mma(c0, wq[i & 1][kc], b0[kc]);
                        mma(c1, wq[i & 1][kc], b1[kc]);
                    }
                }
                const int cr = cti * 16 + (lane >> 4) * 4;
                if (which < 3) {
                    const float sc = which == 0 ? scale : 1.f;
                    for (int r = 0; r < 4; ++r) {
                        const float bb = a.bqkv[wrow + (lane >> 4) * 4 + r];
                        c0[r] = (c0[r] + bb) * sc;
                        c1[r] = (c1[r] + bb) * sc;
                    }
                }
                T* rowm = which == 0 ? Qr : which == 1 ? Kr : which == 2 ? Vr : Or;
                store4<T>(rowm + (th * 32 + (lane & 15)) * CF::LDQ + cr, c0);
                store4<T>(rowm + (th * 32 + 16 + (lane & 15)) * CF::LDQ + cr, c1);
            }
        }
        __syncthreads();

        }
        // ---- (b) P^T (recomputed softmax) and dP^T = V dO^T for this wave's 16 queries -------------------
        f32x4 s[4], dp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { s[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kk = 0; kk < CF::HDP; kk += TR::KCHUNK) {
            const frag_t qf = load_frag<T>(Qr, CF::LDQ, wv * 16, kk);
            const frag_t of = load_frag<T>(Or, CF::LDQ, wv * 16, kk);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                mma(s[kt], load_frag<T>(Kr, CF::LDQ, kt * 16, kk), qf);
                mma(dp[kt], load_frag<T>(Vr, CF::LDQ, kt * 16, kk), of);
            }
        }
        const int qi = wv * 16 + (lane & 15), qy = qi >> 3, qx = qi & 7, qreg = reg[qi];
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            for (int r = 0; r < 4; ++r) {
                const int kj = kt * 16 + (lane >> 4) * 4 + r;
                float v = s[kt][r] + rpbs[(qy - (kj >> 3) + 7) * 15 + (qx - (kj & 7) + 7)];
                if (reg[kj] != qreg) v += -100.0f;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            for (int r = 0; r < 4; ++r) { const float e = Math<T>::exp(s[kt][r] - mx); s[kt][r] = e; sum += e; }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        float dot = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            for (int r = 0; r < 4; ++r) { s[kt][r] *= inv; dot += s[kt][r] * dp[kt][r]; }
        dot += __shfl_xor(dot, 16);
        dot += __shfl_xor(dot, 32);
        if (CF::ALIAS) __syncthreads();          // every wave is done reading v before P overwrites it
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 ds;
            for (int r = 0; r < 4; ++r) ds[r] = s[kt][r] * (dp[kt][r] - dot);
            store4<T>(dS + qi * CF::LDT + kt * 16 + (lane >> 4) * 4, ds);
            store4<T>(Ps + qi * CF::LDT + kt * 16 + (lane >> 4) * 4, s[kt]);
        }
        __syncthreads();

        // ---- (c) dQ = scale dS K, dK = dS^T Q, dV = P^T dO  (this wave's 16 tokens) ------------------------
        {
            const long row = row0 + wv * 16 + (lane & 15);
            T* out = dQKV + row * (3 * C) + h * HD + (lane >> 4) * 4;
#pragma unroll
            for (int ct = 0; ct < TPW; ++ct) {
                f32x4 dq = {0, 0, 0, 0}, dk = {0, 0, 0, 0}, dv = {0, 0, 0, 0};
#pragma unroll
                for (int kk = 0; kk < 64; kk += TR::KCHUNK) {
                    mma(dq, load_frag_tr<T>(Kr, CF::LDQ, ct * 16, kk), load_frag<T>(dS, CF::LDT, wv * 16, kk));
                    mma(dk, load_frag_tr<T>(Qr, CF::LDQ, ct * 16, kk), load_frag_tr<T>(dS, CF::LDT, wv * 16, kk));
                    mma(dv, load_frag_tr<T>(Or, CF::LDQ, ct * 16, kk), load_frag_tr<T>(Ps, CF::LDT, wv * 16, kk));
                }
                for (int r = 0; r < 4; ++r) dq[r] *= scale;
                store4<T>(out + ct * 16, dq);
                store4<T>(out + C + ct * 16, dk);
                store4<T>(out + 2 * C + ct * 16, dv);
            }
        }
        // relative-position-bias gradient: table entry (dy,dx) collects dS over its (8-|dy|)(8-|dx|) pairs
        if (tid < 225) {
            const int oy = tid / 15 - 7, ox = tid % 15 - 7;
            float acc = 0.f;
            for (int ky = (oy < 0 ? -oy : 0); ky < (oy > 0 ? 8 - oy : 8); ++ky)
                for (int kx = (ox < 0 ? -ox : 0); kx < (ox > 0 ? 8 - ox : 8); ++kx)
                    acc += to_f32(dS[((ky + oy) * 8 + kx + ox) * CF::LDT + ky * 8 + kx]);
            a.drpb[((long)blockIdx.x * 225 + tid) * CF::HEADS + h] = acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
struct LnBwdDev {
    const void* X; const void* dXNw; const void* dRes; const float* ln_w;
    void* dX; float* part;      // part [B*nW][2][C]
    WinGeom g; int C;
    const float* ln_b; void* XN;   // optional: also emit LN(x) (rows in the order of dXNw)
    int linear;                    // 1: dXNw rows are plain token order (no window gather)
};

template <class T, int MAXV>          // MAXV: 16-byte vectors of the row per lane (4 lanes per token): C <= 4 * MAXV * VEC
__global__ __launch_bounds__(256) void ln_bwd_win_kernel(LnBwdDev a) {
    constexpr int VEC = Vec16<T>::N;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    // pitch C + 1: the 4-lanes-per-token scalar accesses (row t, column 8q + e) and the per-column sums both stay off each
    // other's banks (C + 4 measured 0.62 conflict cycles per LDS cycle)
    const int C = a.C, LDF = C + 1, tid = threadIdx.x;
    float* Fs = reinterpret_cast<float*>(smem_v);          // [64][LDF] d_xn, then d_xn * xhat
    float* stat = Fs + 64 * LDF;                           // mean, rstd, s1, s2 per token
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* dXN = reinterpret_cast<const T*>(a.dXNw) + (long)blockIdx.x * 64 * C;
    const T* dRes = reinterpret_cast<const T*>(a.dRes);
    T* dX = reinterpret_cast<T*>(a.dX);
    const int nv = C / VEC;
    const int t = tid >> 2, q = tid & 3;
    const long pix = a.linear ? (long)blockIdx.x * 64 + t : win_pixel(a.g, blockIdx.x, t);
    // the lane's share of the token row (x, d_xn, d_res) is loaded ONCE and lives in registers through all phases
    // (the first form re-read x four times and ran three dependent global phases)
    Vec16<T> xv[MAXV], gv[MAXV], dr[MAXV];
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = q + 4 * k;
        if (i < nv) {
            xv[k] = load16<T>(X + pix * C + i * VEC);
            gv[k] = load16<T>(dXN + (long)t * C + i * VEC);
            dr[k] = load16<T>(dRes + pix * C + i * VEC);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
        if (q + 4 * k < nv)
            for (int e = 0; e < VEC; ++e) s += xv[k].get(e);
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
    const float mean = s / (float)C;
    float d2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k)
        if (q + 4 * k < nv)
            for (int e = 0; e < VEC; ++e) { const float d = xv[k].get(e) - mean; d2 += d * d; }
    d2 += __shfl_xor(d2, 1); d2 += __shfl_xor(d2, 2);
    const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = q + 4 * k;
        if (i < nv) {
            for (int e = 0; e < VEC; ++e) {
                const int c = i * VEC + e;
                const float gw = gv[k].get(e) * a.ln_w[c], xh = (xv[k].get(e) - mean) * rstd;
                s1 += gw; s2 += gw * xh;
                Fs[t * LDF + c] = gv[k].get(e);
            }
            if (a.XN) {
                Vec16<T> o;
                for (int e = 0; e < VEC; ++e) o.set(e, (xv[k].get(e) - mean) * rstd * a.ln_w[i * VEC + e] + a.ln_b[i * VEC + e]);
                store16<T>(reinterpret_cast<T*>(a.XN) + ((long)blockIdx.x * 64 + t) * C + i * VEC, o);
            }
        }
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2);
    s1 *= 1.0f / (float)C; s2 *= 1.0f / (float)C;
    // d_x needs nothing from other lanes any more: out it goes, before the column sums
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = q + 4 * k;
        if (i < nv) {
            Vec16<T> o;
            for (int e = 0; e < VEC; ++e) {
                const float dxn = gv[k].get(e), xh = (xv[k].get(e) - mean) * rstd;
                o.set(e, dr[k].get(e) + rstd * (dxn * a.ln_w[i * VEC + e] - s1 - xh * s2));
            }
            store16<T>(dX + pix * C + i * VEC, o);
        }
    }
    __syncthreads();
    float* part = a.part + (long)blockIdx.x * 2 * C;
    for (int c = tid; c < C; c += 256) {
        float acc = 0.f;
        for (int tt = 0; tt < 64; ++tt) acc += Fs[tt * LDF + c];
        part[C + c] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MAXV; ++k) {
        const int i = q + 4 * k;
        if (i < nv)
            for (int e = 0; e < VEC; ++e) Fs[t * LDF + i * VEC + e] = gv[k].get(e) * ((xv[k].get(e) - mean) * rstd);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float acc = 0.f;
        for (int tt = 0; tt < 64; ++tt) acc += Fs[tt * LDF + c];
        part[c] = acc;
    }
}

template <class T, int C, int HD, bool XL>
static int launch_win_bwd_xl(const WinBwdDev& d, hipStream_t s) {
    typedef WinBwdCfg<T, C, HD, XL> CF;
    static_assert(CF::FITS, "win_attn_bwd tiles do not fit LDS");
    allow_big_lds(win_attn_bwd_kernel<T, C, HD, XL>, CF::BYTES);
    const int nblk = d.g.B * (d.g.H / 8) * (d.g.W / 8);
    MPHSIR_LAUNCH(MPHSIR_K_WIN_ATTN_BWD, (win_attn_bwd_kernel<T, C, HD, XL>), dim3(nblk, d.hsplit), dim3(256), CF::BYTES, s, d);
    return MPHSIR_OK;
}

template <class T, int C, int HD>
static int launch_win_bwd_xr(const WinBwdDev& d, hipStream_t s) {
    typedef WinBwdCfg<T, C, HD, false> CF;
    allow_big_lds(win_attn_bwd_kernel<T, C, HD, false, true>, CF::BYTES);
    const int nblk = d.g.B * (d.g.H / 8) * (d.g.W / 8);
    MPHSIR_LAUNCH(MPHSIR_K_WIN_ATTN_BWD, (win_attn_bwd_kernel<T, C, HD, false, true>), dim3(nblk, d.hsplit), dim3(256), CF::BYTES, s, d);
    return MPHSIR_OK;
}

template <class T, int C, int HD>
static int launch_win_bwd(const WinBwdDev& d, hipStream_t s) {
    // measured (tools/bench_win.py, batch 32): C=64 76.8 -> 61.6 us, C=128 / 64-wide heads 132.6 -> 122.3; C=128 / 32-wide heads
    // 45.4 -> 47.3 (164 registers under the three-wave bound): that shape keeps the LDS tile
    if constexpr (sizeof(T) == 2 && (C == 64 || (C == 128 && HD == 64)) && (HD == 32 || HD == 64)) return launch_win_bwd_xr<T, C, HD>(d, s);
    constexpr bool PICK = WinBwdPick<T, C, HD>::XL;
    constexpr int ov = -1;                      // (the X-tile placement is WinBwdPick's: measured both ways)
    if constexpr (WinBwdCfg<T, C, HD, true>::FITS) {
        if (ov == 1 || (ov < 0 && PICK)) return launch_win_bwd_xl<T, C, HD, true>(d, s);
    }
    return launch_win_bwd_xl<T, C, HD, false>(d, s);
}

#define MPHSIR_WINB_SHAPES(X) X(32, 32) X(64, 32) X(64, 64) X(128, 32) X(128, 64) X(256, 32) X(96, 48) X(192, 48) X(192, 96) X(384, 48)

template <class T>
static int dispatch_win_bwd(const WinBwdDev& d, int C, int HD, hipStream_t s) {
#define MPHSIR_WINB_CASE(c, hd) if (C == c && HD == hd) return launch_win_bwd<T, c, hd>(d, s);
    MPHSIR_WINB_SHAPES(MPHSIR_WINB_CASE)
#undef MPHSIR_WINB_CASE
    set_error("win_attn_bwd: (C=%d, head_dim=%d) not instantiated", C, HD);
    return MPHSIR_EINVAL;
}

static bool geom_ok(int B, int H, int W, int shift) {
    return B > 0 && H > 0 && W > 0 && H % 8 == 0 && W % 8 == 0 && (shift == 0 || shift == 4);
}

}  // namespace mphsir

extern "C" int mphsir_combine_bwd(const void* dY, const void* SA, const float* gate, const float* keep, void* dOut, void* dSA,
                                  float* dgate, int32_t B, int32_t H, int32_t W, int32_t C, int32_t shift, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(dY && SA && gate && dSA && dgate, "combine_bwd: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "combine_bwd: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(geom_ok(B, H, W, shift) && C > 0 && C % 8 == 0, "combine_bwd: bad geometry");
    MPHSIR_REQUIRE(aligned16(dY) && aligned16(SA) && aligned16(dSA) && (!dOut || aligned16(dOut)), "combine_bwd: 16-byte alignment required");
    CombBwdDev d{dY, SA, gate, keep, dOut, dSA, dgate, WinGeom{B, H, W, shift}, C};
    const int nblk = B * (H / 8) * (W / 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MPHSIR_F32)
        MPHSIR_LAUNCH(MPHSIR_K_COMBINE_BWD, (combine_bwd_kernel<float>), dim3(nblk), dim3(256), 0, s, d);
    else if (dtype == MPHSIR_BF16)
        MPHSIR_LAUNCH(MPHSIR_K_COMBINE_BWD, (combine_bwd_kernel<bf16_t>), dim3(nblk), dim3(256), 0, s, d);
    else
        MPHSIR_LAUNCH(MPHSIR_K_COMBINE_BWD, (combine_bwd_kernel<f16_t>), dim3(nblk), dim3(256), 0, s, d);
    return MPHSIR_OK;
}

extern "C" int mphsir_win_attn_bwd(const mphsir_win_attn_bwd_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "win_attn_bwd");
    MPHSIR_REQUIRE(a && a->X && a->dSA && a->dmu && a->ln_w && a->ln_b && a->Wqkv && a->bqkv && a->rpb && a->WprojT && a->dQKV &&
                       a->XNw && a->dSAt && a->drpb, "win_attn_bwd: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "win_attn_bwd: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(geom_ok(a->B, a->H, a->W, a->shift) && a->heads > 0 && a->C % a->heads == 0, "win_attn_bwd: bad geometry");
    // fewer than two workgroups per CU: deal the heads of a window to 2, 4, .. workgroups (measured, tools/bench_winb_hs.py: 128 windows
    // of C = 256 / 8 heads 92.9 -> 36.7 us with 4 groups, 42.7 with 8; 256 windows 98 -> 63 with 2; 512 windows are best left whole)
    int hsplit = 1;
    const long nwin = (long)a->B * (a->H / 8) * (a->W / 8);
    while (nwin * hsplit < 512 && a->heads % (2 * hsplit) == 0) hsplit *= 2;
    if (a->head_split > 0) {
        MPHSIR_REQUIRE(a->heads % a->head_split == 0, "win_attn_bwd: head_split=%d must divide heads=%d", a->head_split, a->heads);
        hsplit = a->head_split;
    }
    WinBwdDev d{a->X, a->dSA, a->dmu, a->ln_w, a->ln_b, a->Wqkv, a->bqkv, a->rpb, a->WprojT, a->dQKV, a->XNw, a->dSAt, a->drpb,
                WinGeom{a->B, a->H, a->W, a->shift}, hsplit};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, (dispatch_win_bwd<T_>(d, a->C, a->C / a->heads, s)));
}

extern "C" int mphsir_win_attn_bwd_fits(int32_t C, int32_t heads, int dtype) {
    // every instantiated (width, head_dim) fits in both dtypes since the kernel stopped holding [64][C] tiles of d_sa
    if (heads <= 0 || C % heads != 0 || !MPHSIR_DTYPE_OK(dtype)) return 0;
    const int hd = C / heads;
#define MPHSIR_WINB_CASE(c, h_) if (C == c && hd == h_) return 1;
    MPHSIR_WINB_SHAPES(MPHSIR_WINB_CASE)
#undef MPHSIR_WINB_CASE
    return 0;
}

extern "C" int mphsir_ln_bwd_win(const void* X, const void* dXNw, const void* dRes, const float* ln_w, void* dX, float* part,
                                 int32_t B, int32_t H, int32_t W, int32_t C, int32_t shift, const float* ln_b, void* XN, int32_t linear,
                                 int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && dXNw && dRes && ln_w && dX && part, "ln_bwd_win: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "ln_bwd_win: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(geom_ok(B, H, W, shift) && C > 0 && C % 32 == 0 && C <= 512, "ln_bwd_win: bad geometry");
    MPHSIR_REQUIRE(aligned16(X) && aligned16(dXNw) && aligned16(dRes) && aligned16(dX), "ln_bwd_win: 16-byte alignment required");
    MPHSIR_REQUIRE(!XN || ln_b, "ln_bwd_win: XN output needs ln_b");
    LnBwdDev d{X, dXNw, dRes, ln_w, dX, part, WinGeom{B, H, W, shift}, C, ln_b, XN, linear};
    const size_t shmem = (64 * (size_t)(C + 1) + 256 + 3) * sizeof(float);
    const int nblk = B * (H / 8) * (W / 8);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int vec = 16 / dtype_size(dtype), vpt = (C / vec + 3) / 4;
#define MPHSIR_LNB(MV)                                                                                                   \
    return MPHSIR_DISPATCH_T(dtype, ([&]() -> int {                                                                      \
        allow_big_lds(ln_bwd_win_kernel<T_, MV>, shmem);                                                                 \
        MPHSIR_LAUNCH(MPHSIR_K_LN_BWD_WIN, (ln_bwd_win_kernel<T_, MV>), dim3(nblk), dim3(256), shmem, s, d);             \
        return MPHSIR_OK;                                                                                                \
    }()))
    if (vpt <= 4) { MPHSIR_LNB(4); }
    if (vpt <= 8) { MPHSIR_LNB(8); }
    if (vpt <= 12) { MPHSIR_LNB(12); }
    if (vpt <= 16) { MPHSIR_LNB(16); }
#undef MPHSIR_LNB
    if (dtype == MPHSIR_F32 && vpt <= 32) {       // C = 512 in fp32 only (a 16-bit row of 512 channels is 16 vectors per lane): the 32-vector
        allow_big_lds(ln_bwd_win_kernel<float, 32>, shmem);      // form of the 16-bit types kept 96 vectors live and spilled 1356 registers
        MPHSIR_LAUNCH(MPHSIR_K_LN_BWD_WIN, (ln_bwd_win_kernel<float, 32>), dim3(nblk), dim3(256), shmem, s, d);
        return MPHSIR_OK;
    }
    set_error("ln_bwd_win: C=%d too wide", C);
    return MPHSIR_EINVAL;
}
