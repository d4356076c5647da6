// win_attn_fwd: LayerNorm -> cyclic shift -> 8x8 window MSA -> proj, plus the window mean the spectral-prompt gate reads.
//
// Replaces, per PGSSTB block (net/MP_HSIR.py:662-713): norm1 (:667), torch.roll (:672),
// window_partition (:677), Spatial_Attention.forward (:193-218: qkv Linear, q*scale, QK^T,
// relative-position bias gather, -100 shift mask, softmax, AV, proj), window_reverse + roll back
// (:690-696) and the window mean of PG_Spectral_Attention.forward (:135; the gate itself is csrc/pg_gate.hip, 16 windows
// per workgroup, and the final `out*shortcut` multiply, :153, is folded into gemm_tok's epilogue 2).
//
// One 256-thread workgroup = one window (64 tokens).  Roll/partition/reverse are address arithmetic
// on the channels-last cube; the shift mask is computed from coordinates (no mask tensor).  Per head:
//   (a) q,k [tok][hd] and v^T [hd][tok] by MFMA from the LN-ed tile in LDS (weights from L2);
//   (b) S^T = K Q^T so that one lane owns one query column: softmax = in-lane max/sum over 16
//       values + two xor-shuffles; P goes to LDS as the B operand of
//   (c) O^T = V^T P^T, stored as [tok][hd], which is (d) the K-slice of the output projection,
//       accumulated over heads in C/16 persistent fp32 tiles per wave.
// After (a) every step is wave-local (wave w owns query rows 16w..16w+15).
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct WinAttnDev {
    const void* X;
    const float* ln_w; const float* ln_b;
    const void* Wqkv; const float* bqkv;
    const float* rpb;
    const void* Wproj; const float* bproj;
    void* SA;
    float* mu;      // [B*nW][C]: window mean of SA = the input of the spectral-prompt gate (mphsir_pg_gate_fwd)
    void* Oattn;    // optional [B*nW*64][C]: attention output before proj, window-token order (training: dWproj)
    int B, H, W, shift;
    unsigned long long* dbg;    // diagnostics (mphsir_debug): shader-clock stamps of workgroup 0 at its phase boundaries
};
static unsigned long long* g_win_dbg = nullptr;
#define WIN_MARK(k) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[k] = __builtin_amdgcn_s_memtime(); } while (0)

template <class T, int C, int HD> struct WinAttnCfg {
    static constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    static constexpr int HEADS = C / HD;
    static constexpr int HDP = (HD + ElemTraits<T>::KCHUNK - 1) / ElemTraits<T>::KCHUNK * ElemTraits<T>::KCHUNK;
    static constexpr int LDX = C + (64 * (C + PAD) * sizeof(T) > 98304 ? 0 : PAD);   // drop the pad when LDS is tight
    static constexpr int LDQ = HDP + PAD;
    static constexpr int LDV = 64 + PAD;
    static constexpr int LDP = 64 + PAD;
    // the P tile's region doubles as the stage of one part (q, k or v rows of the head) of the qkv weights: P is only live
    // between the softmax and P V, the weights only while q, k, v are computed
    static constexpr int LDWS = C + PAD;
    // rows per weight stage: a whole part (HD rows) where LDS has room for it, else one 16-channel tile at a time
    static constexpr size_t REST = (64 * LDX + 2 * 64 * LDQ + HD * LDV) * sizeof(T) + (225 + 64) * 4;
    static constexpr int SR = (REST + (size_t)HD * LDWS * sizeof(T) <= 160 * 1024) ? HD : 16;
    static constexpr bool STAGE = REST + (size_t)SR * LDWS * sizeof(T) <= 160 * 1024;
    // XR (C = 128, 16-bit, staged weights): the LN-ed tile only passes through LDS (inside the q | k | v^T tiles, which are
    // not written before phase (a)) on its way into the waves' registers, and the output tile is staged there too: no X region
    static constexpr bool XR = STAGE && C == 128 && sizeof(T) == 2 && HDP == HD && 64 * LDX <= 2 * 64 * LDQ + HD * LDV;
    static constexpr size_t XS = XR ? 0 : 64 * LDX, QS = 64 * LDQ, VS = HD * LDV,
                            PS = (!STAGE || 64 * LDP > SR * LDWS) ? 64 * LDP : SR * LDWS;
    static constexpr size_t T_ELEMS = XS + 2 * QS + VS + PS;
    static constexpr size_t F_WORDS = 225 + 64;                      // bias column of one head, region ids
    static constexpr size_t BYTES = T_ELEMS * sizeof(T) + F_WORDS * 4;
    static_assert(BYTES <= 160 * 1024, "window-attention tile does not fit LDS");
    static_assert(C % 32 == 0 && HD % 16 == 0 && C % HD == 0, "unsupported width");
};

// three waves per SIMD where the registers allow it without spilling (32-wide heads: 156; 64-wide heads need 230 -- held to 168 the
// kernel spills 74 registers and a launch goes from 99 to 156 us)
template <class T, int C, int HD> constexpr int win_min_waves() { return WinAttnCfg<T, C, HD>::XR && HD <= 32 ? 3 : 1; }

template <class T, int C, int HD>
__global__ __launch_bounds__(256, (win_min_waves<T, C, HD>())) void win_attn_kernel(WinAttnDev a) {
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    typedef WinAttnCfg<T, C, HD> CF;
    constexpr int VEC = Vec16<T>::N;
    constexpr int NCT = C / 16;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* Xs = reinterpret_cast<T*>(smem_v);                        // XR: aliases the q | k | v^T tiles
    T* Qs = Xs + CF::XS;
    T* Ks = Qs + CF::QS;
    T* Vt = Ks + CF::QS;
    T* Ps = Vt + CF::VS;
    float* rpbs = reinterpret_cast<float*>(Ps + CF::PS);
    int* reg = reinterpret_cast<int*>(rpbs + 225);

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int nwx = a.W >> 3, nW = (a.H >> 3) * nwx;
    const int b = blockIdx.x / nW, wi = blockIdx.x % nW, wy = wi / nwx, wx = wi % nwx;
    const T* X = reinterpret_cast<const T*>(a.X);

    // token t=(ty,tx) of this window sits at shifted-frame (wy*8+ty, wx*8+tx) = image ((..+shift)%H, (..+shift)%W)
    auto pixel_of = [&](int t) -> long {
        const int ys = wy * 8 + (t >> 3), xs = wx * 8 + (t & 7);
        const int y = (ys + a.shift) % a.H, x = (xs + a.shift) % a.W;
        return ((long)b * a.H + y) * a.W + x;
    };

    WIN_MARK(0);
    // ---- LayerNorm(norm1) of the 64 tokens into LDS; 4 adjacent lanes per token -----------------
    {
        constexpr int NV = C / VEC, VPT = NV / 4;
        const int t = tid >> 2, q = tid & 3;
        const T* row = X + pixel_of(t) * C;
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
            store16<T>(Xs + t * CF::LDX + c0, o);
        }
        if (tid < 64) {   // region id of token tid in the shifted frame (calculate_mask, :639-660)
            const int ys = wy * 8 + (tid >> 3), xs = wx * 8 + (tid & 7);
            const int ry = (ys >= a.H - 8) + (ys >= a.H - 4), rx = (xs >= a.W - 8) + (xs >= a.W - 4);
            reg[tid] = a.shift ? 3 * ry + rx : 0;
        }
        if (CF::HDP != HD)   // zero the K-padding columns of q / k (= O) once
            for (int i = tid; i < 2 * 64 * (CF::HDP - HD); i += 256) {
                const int rr = i / (CF::HDP - HD), cc = HD + i % (CF::HDP - HD);
                Qs[rr * CF::LDQ + cc] = from_f32<T>(0.f);   // rr in [0,128): Qs and Ks are contiguous
            }
    }

    const T* Wqkv = reinterpret_cast<const T*>(a.Wqkv);
    const T* Wp = reinterpret_cast<const T*>(a.Wproj);
    const float scale = rsqrtf((float)HD);
    f32x4 out[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) out[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // weight stage: SR rows x C (one part q/k/v of the head, or one 16-channel tile of it where LDS is tight), 16-byte vectors
    // over all 256 threads, register prefetch one stage ahead
    constexpr int WVT = CF::SR * (C / VEC), NWV = (WVT + 255) / 256;
    Vec16<T> wpre[NWV];
    auto wload = [&](int h, int sub) __attribute__((always_inline)) {       // stage `sub` of head h: rows (sub % (HD/SR)) * SR .. of part sub / (HD/SR)
        const int row0 = (sub / (HD / CF::SR)) * C + h * HD + (sub % (HD / CF::SR)) * CF::SR;
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < WVT) wpre[i] = load16<T>(Wqkv + (long)(row0 + idx / (C / VEC)) * C + (idx % (C / VEC)) * VEC);
        }
    };
    auto wstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + 256 * i;
            if (idx < WVT) store16<T>(Ps + (idx / (C / VEC)) * CF::LDWS + (idx % (C / VEC)) * VEC, wpre[i]);
        }
    };
    if (CF::STAGE) wload(0, 0);
    // C = 128: a wave multiplies the same 32 tokens (half `wv & 1` of the window) in every unit of phase (a); their LN-ed rows
    // are read from LDS ONCE as MFMA fragments and stay in registers over all heads (a unit: 1204 -> 732 cycles by the
    // shader-clock stamps, the launch 2-4 % shorter).  Narrower nets lose a wave per SIMD to the 32 registers, wider ones two.
    constexpr bool XREG = CF::XR;
    constexpr int NKX = XREG ? C / TR::KCHUNK : 1;
    frag_t xf[2][NKX];

    for (int h = 0; h < CF::HEADS; ++h) {
        __syncthreads();   // Xs ready (h=0) / previous head's K, V^T, bias column and P tile (= the weight stage) no longer read
        if (h == 0) WIN_MARK(1);
        if constexpr (XREG) {
            if (h == 0) {
#pragma unroll
                for (int kc = 0; kc < NKX; ++kc) {
                    xf[0][kc] = load_frag<T>(Xs, CF::LDX, (wv & 1) * 32, kc * TR::KCHUNK);
                    xf[1][kc] = load_frag<T>(Xs, CF::LDX, (wv & 1) * 32 + 16, kc * TR::KCHUNK);
                }
            }
        }
        if (tid < 225) rpbs[tid] = a.rpb[tid * CF::HEADS + h];
        constexpr int TPW = HD / 16;                 // channel tiles per q/k/v
        if constexpr (CF::STAGE) {
        // ---- (a) q, k, v^T for head h ---------------------------------------------------------
        // The weight rows go through LDS one part (q, k, v) at a time: all 256 threads fetch the NEXT part's HD x C rows from
        // L2 (coalesced, all in flight) while the current part is multiplied, and store them once its readers are past a
        // barrier.  (First version: every 4-MFMA unit fetched its own fragments from L2 with a one-unit prefetch -- 5.0 k of the
        // 8.6 k cycles of a head at C=64, 13.8 k of 23 k at C=128/hd=64, shader-clock stamps of mphsir_debug.)
        // unit u = (16-channel tile of the part, half of the 64 tokens); a wave walks units u = wv, wv+4, ...
        constexpr int NKC = C / TR::KCHUNK;
        T* Wst = Ps;                                 // [SR][LDWS]
        constexpr int SR = CF::SR, NSUB = 3 * HD / SR, TPS = SR / 16;      // stages per head, channel tiles per stage
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int part = sub / (HD / SR), r0 = (sub % (HD / SR)) * SR;   // rows r0.. of q_h / k_h / v_h
            if (sub > 0) __syncthreads();            // the previous stage's fragments are read
            wstore();
            if (sub + 1 < NSUB) wload(h, sub + 1);
            else if (h + 1 < CF::HEADS) wload(h + 1, 0);
            __syncthreads();
            for (int u = wv; u < 2 * TPS; u += 4) {
                const int ctl = u >> 1, cti = r0 / 16 + ctl, th = u & 1;    // tile inside the stage / inside the head
                const int wrow = part * C + h * HD + cti * 16;
                f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
                if (part < 2) {
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        const frag_t w = load_frag<T>(Wst, CF::LDWS, ctl * 16, kc * TR::KCHUNK);
                        mma(c0, w, XREG ? xf[0][XREG ? kc : 0] : load_frag<T>(Xs, CF::LDX, th * 32, kc * TR::KCHUNK));
                        mma(c1, w, XREG ? xf[1][XREG ? kc : 0] : load_frag<T>(Xs, CF::LDX, th * 32 + 16, kc * TR::KCHUNK));
                    }
                    const int cr = cti * 16 + (lane >> 4) * 4;        // 4 consecutive channels of the head
                    const float sc = part == 0 ? scale : 1.f;
                    for (int r = 0; r < 4; ++r) {
                        const float bb = a.bqkv[wrow + (lane >> 4) * 4 + r];
                        c0[r] = (c0[r] + bb) * sc;
                        c1[r] = (c1[r] + bb) * sc;
                    }
                    T* dst = part == 0 ? Qs : Ks;
                    store4<T>(dst + (th * 32 + (lane & 15)) * CF::LDQ + cr, c0);
                    store4<T>(dst + (th * 32 + 16 + (lane & 15)) * CF::LDQ + cr, c1);
                } else {
#pragma unroll
                    for (int kc = 0; kc < NKC; ++kc) {
                        const frag_t w = load_frag<T>(Wst, CF::LDWS, ctl * 16, kc * TR::KCHUNK);
                        mma(c0, XREG ? xf[0][XREG ? kc : 0] : load_frag<T>(Xs, CF::LDX, th * 32, kc * TR::KCHUNK), w);
                        mma(c1, XREG ? xf[1][XREG ? kc : 0] : load_frag<T>(Xs, CF::LDX, th * 32 + 16, kc * TR::KCHUNK), w);
                    }
                    const float bb = a.bqkv[wrow + (lane & 15)];
                    for (int r = 0; r < 4; ++r) { c0[r] += bb; c1[r] += bb; }
                    T* vrow = Vt + (cti * 16 + (lane & 15)) * CF::LDV + (lane >> 4) * 4;   // 4 consecutive tokens
                    store4<T>(vrow + th * 32, c0);
                    store4<T>(vrow + th * 32 + 16, c1);
                }
            }
        }
        __syncthreads();
        } else {      // LDS too tight even for a 16-row stage (fp32 at C=384): weight fragments straight from L2
        // ---- (a) q, k, v^T for head h ---------------------------------------------------------
        // unit u = (16-channel tile of q|k|v, half of the 64 tokens); a wave walks units u = wv, wv+4, ...  The weight
        // fragments come straight from L2 and each feeds only two MFMAs, so their latency is the cost of this phase:
        // the next unit's fragments are loaded (second register set) while the current unit runs.
        constexpr int NUNITS = 3 * TPW * 2, NKC = C / TR::KCHUNK;
        auto loadw = [&](frag_t (&w)[NKC], int u) __attribute__((always_inline)) {
            if (u < NUNITS) {
                const int ct = u >> 1, wrow = (ct / TPW) * C + h * HD + (ct % TPW) * 16;
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) w[kc] = load_frag<T>(Wqkv, C, wrow, kc * TR::KCHUNK);
            }
        };
        auto unit = [&](const frag_t (&w)[NKC], int u) __attribute__((always_inline)) {
            if (u >= NUNITS) return;
            const int ct = u >> 1, th = u & 1, which = ct / TPW, cti = ct % TPW;
            const int wrow = which * C + h * HD + cti * 16;
            f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
            if (which < 2) {
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) {
                    mma(c0, w[kc], load_frag<T>(Xs, CF::LDX, th * 32, kc * TR::KCHUNK));
                    mma(c1, w[kc], load_frag<T>(Xs, CF::LDX, th * 32 + 16, kc * TR::KCHUNK));
                }
                const int cr = cti * 16 + (lane >> 4) * 4;        // 4 consecutive channels of the head
                const float sc = which == 0 ? scale : 1.f;
                for (int r = 0; r < 4; ++r) {
                    const float bb = a.bqkv[wrow + (lane >> 4) * 4 + r];
                    c0[r] = (c0[r] + bb) * sc;
                    c1[r] = (c1[r] + bb) * sc;
                }
                T* dst = which == 0 ? Qs : Ks;
                store4<T>(dst + (th * 32 + (lane & 15)) * CF::LDQ + cr, c0);
                store4<T>(dst + (th * 32 + 16 + (lane & 15)) * CF::LDQ + cr, c1);
            } else {
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) {
                    mma(c0, load_frag<T>(Xs, CF::LDX, th * 32, kc * TR::KCHUNK), w[kc]);
                    mma(c1, load_frag<T>(Xs, CF::LDX, th * 32 + 16, kc * TR::KCHUNK), w[kc]);
                }
                const float bb = a.bqkv[wrow + (lane & 15)];
                for (int r = 0; r < 4; ++r) { c0[r] += bb; c1[r] += bb; }
                T* vrow = Vt + (cti * 16 + (lane & 15)) * CF::LDV + (lane >> 4) * 4;   // 4 consecutive tokens
                store4<T>(vrow + th * 32, c0);
                store4<T>(vrow + th * 32 + 16, c1);
            }
        };
        {
            frag_t wa[NKC], wb[NKC];
            loadw(wa, wv);
            for (int u = wv; u < NUNITS; u += 8) {
                loadw(wb, u + 4);
                unit(wa, u);
                loadw(wa, u + 8);
                unit(wb, u + 4);
            }
        }
        __syncthreads();
        }
        if (h == 0) WIN_MARK(2);

        // ---- (b) S^T = K Q^T for this wave's 16 queries, + bias + mask, softmax over keys --------
        // the proj fragments of phase (d) (straight from L2, one MFMA each) are requested now: their latency hides
        // behind the softmax and P V of this head
        constexpr int NKD = CF::HDP / TR::KCHUNK;
        frag_t wpf[NKD][NCT];
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
                wpf[kd][ct] = load_frag<T>(Wp, CF::HEADS * CF::HDP, ct * 16, h * CF::HDP + kd * TR::KCHUNK);
        f32x4 s[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < CF::HDP; kk += TR::KCHUNK) {
            const frag_t qf = load_frag<T>(Qs, CF::LDQ, wv * 16, kk);
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) mma(s[kt], load_frag<T>(Ks, CF::LDQ, kt * 16, kk), qf);
        }
        const int qi = wv * 16 + (lane & 15), qy = qi >> 3, qx = qi & 7, qreg = reg[qi];
        // relative-position index of (query qi, key kj = 16 kt + 4 (lane >> 4) + r): (qy - ky + 7) * 15 + (qx - kx + 7) with
        // ky = 2 kt + (lane >> 5), kx = 4 ((lane >> 4) & 1) + r  ->  one per-lane base minus the compile-time 30 kt + r
        const float* rp = rpbs + (qy + 7) * 15 + qx + 7 - 15 * (lane >> 5) - 4 * ((lane >> 4) & 1);
        const int* kreg = reg + (lane >> 4) * 4;
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
            for (int r = 0; r < 4; ++r) {
                float v = s[kt][r] + rp[-(30 * kt + r)];
                if (kreg[kt * 16 + r] != qreg) v += -100.0f;
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
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 p = s[kt];
            for (int r = 0; r < 4; ++r) p[r] *= inv;
            store4<T>(Ps + qi * CF::LDP + kt * 16 + (lane >> 4) * 4, p);
        }
        wave_barrier();    // P rows of this wave's 16 queries are read back by this wave only (V^T is complete since the barrier above)
        if (h == 0) WIN_MARK(3);

        // ---- (c) O^T = V^T P^T -> O [tok][hd] (over the q tile rows of this wave) ---------------
#pragma unroll
        for (int ct = 0; ct < TPW; ++ct) {
            f32x4 o = {0, 0, 0, 0};
#pragma unroll
            for (int kk = 0; kk < 64; kk += TR::KCHUNK)
                mma(o, load_frag<T>(Vt, CF::LDV, ct * 16, kk), load_frag<T>(Ps, CF::LDP, wv * 16, kk));
            store4<T>(Qs + qi * CF::LDQ + ct * 16 + (lane >> 4) * 4, o);
        }
        wave_barrier();    // O overwrites this wave's own q rows (nobody else reads them) and is read back by this wave only
        if (h == 0) WIN_MARK(4);
        if (a.Oattn) {   // training: keep softmax(QK^T)V (before proj) for the proj weight gradient; a wave stores its own 16 rows
            constexpr int VPH = HD / VEC;
            T* Oa = reinterpret_cast<T*>(a.Oattn);
            for (int idx = lane; idx < 16 * VPH; idx += 64) {
                const int t = wv * 16 + idx / VPH, c0 = (idx % VPH) * VEC;
                store16<T>(Oa + ((long)blockIdx.x * 64 + t) * C + h * HD + c0, load16<T>(Qs + t * CF::LDQ + c0));
            }
        }

        // ---- (d) out[co][tok] += Wproj[co][h*hd + :] * O[tok][:] -------------------------------
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) {
            const frag_t of = load_frag<T>(Qs, CF::LDQ, wv * 16, kd * TR::KCHUNK);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) mma(out[ct], wpf[kd][ct], of);
        }
        if (h == 0) WIN_MARK(5);
    }
    __syncthreads();
    WIN_MARK(6);

    // ---- proj bias, stage the 64 x C output tile in LDS (reusing the X tile) --------------------
    {
        const int tok = wv * 16 + (lane & 15), cr = (lane >> 4) * 4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            f32x4 o = out[ct];
            for (int r = 0; r < 4; ++r) o[r] += a.bproj[ct * 16 + cr + r];
            store4<T>(Xs + tok * CF::LDX + ct * 16 + cr, o);
        }
    }
    __syncthreads();
    T* SA = reinterpret_cast<T*>(a.SA);
    {
        constexpr int NV = C / VEC;
        for (int idx = tid; idx < 64 * NV; idx += 256) {
            const int t = idx / NV, c0 = (idx % NV) * VEC;
            store16<T>(SA + pixel_of(t) * C + c0, load16<T>(Xs + t * CF::LDX + c0));
        }
    }

    // ---- window mean of the attention output: the input of the local spectral-prompt gate (csrc/pg_gate.hip) ---------
    for (int c = tid; c < C; c += 256) {
        float acc = 0.f;
        for (int t = 0; t < 64; ++t) acc += to_f32(Xs[t * CF::LDX + c]);
        a.mu[(long)blockIdx.x * C + c] = acc * (1.0f / 64.0f);
    }
    WIN_MARK(7);
}

template <class T, int C, int HD>
static int launch_win(const WinAttnDev& d, hipStream_t s) {
    const size_t shmem = WinAttnCfg<T, C, HD>::BYTES;
    allow_big_lds(win_attn_kernel<T, C, HD>, shmem);
    const int nblk = d.B * (d.H / 8) * (d.W / 8);
    MPHSIR_LAUNCH(MPHSIR_K_WIN_ATTN, (win_attn_kernel<T, C, HD>), dim3(nblk), dim3(256), shmem, s, d);
    return MPHSIR_OK;
}

template <class T>
static int dispatch_win(const WinAttnDev& d, int C, int HD, hipStream_t s) {
#define MPHSIR_WIN_CASE(c, hd) if (C == c && HD == hd) return launch_win<T, c, hd>(d, s);
    MPHSIR_WIN_CASE(32, 32) MPHSIR_WIN_CASE(64, 32) MPHSIR_WIN_CASE(64, 64) MPHSIR_WIN_CASE(128, 32)
    MPHSIR_WIN_CASE(128, 64) MPHSIR_WIN_CASE(256, 32)
    MPHSIR_WIN_CASE(96, 48) MPHSIR_WIN_CASE(192, 48) MPHSIR_WIN_CASE(192, 96) MPHSIR_WIN_CASE(384, 48)
#undef MPHSIR_WIN_CASE
    set_error("win_attn: (C=%d, head_dim=%d) not instantiated", C, HD);
    return MPHSIR_EINVAL;
}

}  // namespace mphsir

extern "C" int mphsir_win_attn_hdp(int head_dim, int dtype) {
    const int kc = dtype == MPHSIR_F32 ? 16 : 32;      // bf16 and f16 share the 32-wide MFMA K-chunk
    return (head_dim + kc - 1) / kc * kc;
}

namespace mphsir { void win_debug_buffer(unsigned long long* p) { g_win_dbg = p; } }    // mphsir_debug(MPHSIR_DEBUG_WIN_ATTN, ...)

extern "C" int mphsir_win_attn_fwd(const mphsir_win_attn_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "win_attn_fwd");
    MPHSIR_REQUIRE(a && a->X && a->SA && a->mu && a->Wqkv && a->bqkv && a->rpb && a->Wproj && a->bproj && a->ln_w && a->ln_b,
                   "win_attn: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "win_attn: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->H % 8 == 0 && a->W % 8 == 0, "win_attn: H,W must be multiples of 8");
    MPHSIR_REQUIRE(a->shift == 0 || a->shift == 4, "win_attn: shift must be 0 or 4");
    MPHSIR_REQUIRE(a->heads > 0 && a->C % a->heads == 0, "win_attn: C %% heads != 0");
    MPHSIR_REQUIRE(aligned16(a->X) && aligned16(a->SA) && aligned16(a->Wqkv) && aligned16(a->Wproj), "win_attn: 16-byte alignment required");
    WinAttnDev d{a->X, a->ln_w, a->ln_b, a->Wqkv, a->bqkv, a->rpb, a->Wproj, a->bproj, a->SA, a->mu, a->Oattn, a->B, a->H, a->W, a->shift, g_win_dbg};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, (dispatch_win<T_>(d, a->C, a->C / a->heads, s)));
}
