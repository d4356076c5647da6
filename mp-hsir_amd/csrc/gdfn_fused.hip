// gdfn_fused: the whole gated depthwise feed-forward of a (Cross)TransformerBlock in ONE launch (training: t is written too).
//
// Reference: FFN.forward / FeedForward.forward net/MP_HSIR.py:259-265 == :385-391 inside the pre-norm residual of
// CrossTransformer.forward :286 / TransformerBlock.forward :477:
//     t = project_in(LN(a));  [x1|x2] = dwconv3x3(t).chunk(2);  y = a + project_out(gelu(x1) * x2)
// The three-launch path (gemm_tok -> dwconv_gate -> gemm_tok) writes t (2*hid values per pixel, 5.3x the input) and
// u = gelu(x1)*x2 (hid values) to HBM and reads both back: 1.3 GB for one 512x512 cube at dim 128 against 0.2 GB of a and y.
// Here neither leaves the chip.  Same tiling as the tile form of the fused pass A (spectral_fused.hip): one workgroup (8
// waves) owns an 8x16-pixel tile (8x8 for D > 128), keeps the LayerNorm-ed MFMA fragments of the tile + one-pixel halo
// (180 pixels in 12 row blocks; 100 in 7) in registers, and walks the hidden dimension in slabs of 32 channel PAIRS (32 gelu-side channels + their 32
// partners):
//   (1) t_s = W_in[slab] LN(a)^T by MFMA (weights through LDS: stored during the previous slab's depthwise pass, requested
//       from L2 a slab before that), transposed accumulators
//       -> fp32 t tile [halo pixel][64] in LDS; rows of pixels outside the image are zeros (the depthwise conv zero-pads t);
//   (2) depthwise 3x3 from LDS: one thread = (side, 4 channels, 4 pixels of a row); the two sides of a channel group sit in
//       adjacent lanes, swap half of their results through one shuffle each and finish gelu(x1)*x2 for 2 pixels;
//       u_s -> LDS [128 pixel][32] in the storage type (the three-launch path rounds u the same way);
//   (3) y += W_out[:, slab] u_s^T: a wave owns a 16-pixel block x all (8x8 tile: half of the) D output channels as
//       persistent fp32 accumulator tiles over all slabs; issued together with the NEXT slab's project_in MFMAs (two
//       barriers per slab).
// After the last slab: y + a (residual, one rounding), staged through LDS, stored as whole 16-byte row chunks.
// t is kept in fp32 between the 1x1 conv and the depthwise conv (the three-launch path rounds it to the storage type).
#include <type_traits>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

constexpr int GF_TH = 8, GF_HH = GF_TH + 2;
constexpr int GF_THREADS = 512, GF_WAVES = 8;
constexpr int GF_SLAB = 32;                     // channel pairs per slab
// pixel tile 8 x TW (+ one-pixel halo).  TW = 16 (180 halo pixels in 12 MFMA row blocks, 1.5 t rows per output pixel) up to
// D = 128; the wider nets take TW = 8 (100 halo pixels, 7 row blocks, 1.75x): a wave then keeps ONE row block of LN(a)
// fragments and half of a 16-pixel block's output channels, which is what fits 256 registers at D = 192 / 256.
template <int TW> struct GfGeo {
    static constexpr int HW = TW + 2, ROWS = GF_HH * HW, MB = (ROWS + 15) / 16, PIX = GF_TH * TW;
    static constexpr int MBW = (MB + GF_WAVES - 1) / GF_WAVES;     // row blocks per wave: wave w owns blocks w, w+8, .. (< MB)
    static constexpr int PXB = PIX / 16, WPB = GF_WAVES / PXB;     // 16-pixel output blocks; waves sharing one (by output channels)
    static constexpr int PPT = PIX / 32;                           // pixels per thread of the depthwise pass
};
template <int D> struct GfTile { static constexpr int TW = D <= 128 ? 16 : 8; };

struct GdfnDev {
    const void* X; long ldx;
    const float* ln_w; const float* ln_b;
    const void* Win;                             // [2 HP][D]: rows 0..HP-1 gelu side, HP..2HP-1 partner side
    const float* w9; long ldw;                   // [9][ldw >= 2 HP]
    const void* Wout;                            // [D][HP]
    void* Y; long ldy;
    int B, H, W, HP, nsplit;
    void* Tout; long ldt;                        // training: t = project_in(LN(a)) [B*H*W][ldt >= 2 HP], kept for the backward
};

template <class T, int D> struct GfLds {
    typedef GfGeo<GfTile<D>::TW> G;
    static constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    // t tile pitch 2*SLAB + 4 floats: a ds_read_b128 of 16 lanes that differ in the channel group covers all 64 banks once
    static constexpr int LDW = D + PAD, LDT = 2 * GF_SLAB + 4, LDU = GF_SLAB + PAD, LDO = GF_SLAB + PAD, LDY = D + PAD;
    static constexpr size_t w_elems = (size_t)2 * GF_SLAB * LDW, o_elems = (size_t)D * LDO, u_elems = (size_t)G::PIX * LDU;
    static constexpr size_t t_floats = (size_t)G::MB * 16 * LDT, tap_floats = 9 * 2 * GF_SLAB;
    // the output tile is staged over the u and t tiles (dead after the last slab); the weight stages stay live across tiles
    static constexpr size_t ut_bytes_raw = u_elems * sizeof(T) + t_floats * sizeof(float), y_bytes = (size_t)G::PIX * LDY * sizeof(T);
    static constexpr size_t ut_bytes = ut_bytes_raw > y_bytes ? ut_bytes_raw : y_bytes;
    static constexpr size_t bytes = (w_elems + 2 * o_elems) * sizeof(T) + ut_bytes + 2 * tap_floats * sizeof(float);
    static_assert(bytes <= 160 * 1024, "fused GDFN tile does not fit LDS");
};

template <class T, int D, bool KEEP>
__global__ __launch_bounds__(GF_THREADS, 2) void gdfn_fused_kernel(GdfnDev a) {
    typedef ElemTraits<T> TR;
    typedef GfLds<T, D> L;
    typedef typename L::G G;
    constexpr int GF_TW = GfTile<D>::TW, GF_HW = G::HW, GF_ROWS = G::ROWS, GF_MB = G::MB, GF_PIX = G::PIX, GF_MBW = G::MBW;
    constexpr int PXB = G::PXB, WPB = G::WPB, PPT = G::PPT, NOBW = D / 16 / WPB;
    constexpr int VEC = Vec16<T>::N, KCH = TR::KCHUNK, EPL = TR::EPL;
    constexpr int SLAB = GF_SLAB, NKC = D / KCH, NB = 2 * SLAB / 16, NOB = D / 16;
    constexpr int LDW = L::LDW, LDT = L::LDT, LDU = L::LDU, LDO = L::LDO, LDY = L::LDY;
    static_assert(D % KCH == 0 && SLAB == KCH && NOB % WPB == 0 && PPT % 2 == 0, "shape");
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* Ws = reinterpret_cast<T*>(smem_v);                       // [2 SLAB][LDW]    project_in rows of the current slab
    T* Wos = Ws + L::w_elems;                                   // [2][D][LDO]      project_out columns of the current / previous slab
    T* Us = Wos + 2 * L::o_elems;                               // [PIX][LDU]       u = gelu(x1) * x2 of the interior pixels
    float* Ts = reinterpret_cast<float*>(Us + L::u_elems);     // [16 MB][LDT]     fp32 t of the halo tile, this slab
    T* Ys = Us;                                                 // [PIX][LDY]       output tile (after the last slab), over u | t
    float* tapsS = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(Us) + L::ut_bytes);     // [2][9][2 SLAB]

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int blk = (gridDim.x & 7) == 0 ? (int)xcd_contiguous_block() : (int)blockIdx.x;   // neighbours share halo rows
    const int b = blk / a.nsplit, sp = blk % a.nsplit;
    const int tilesx = a.W / GF_TW, tiles = (a.H / GF_TH) * tilesx, tpw = tiles / a.nsplit;
    const int nslab = a.HP / SLAB, HP = a.HP;
    const long img = (long)b * a.H * a.W;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* Wi = reinterpret_cast<const T*>(a.Win);
    const T* Wo = reinterpret_cast<const T*>(a.Wout);
    T* Y = reinterpret_cast<T*>(a.Y);

    // Software pipeline over the flattened steps g = (tile, slab): the weights and taps of step g+1 go to LDS during the
    // depthwise pass of step g (the project_in stage is dead once t is complete; project_out and the taps are double
    // buffered), those of step g+2 are requested from L2 into registers at the same time -- two barriers per slab and no
    // global latency between two phases of the steady state.
    constexpr int WVT = 2 * SLAB * (D / VEC), NWV = (WVT + GF_THREADS - 1) / GF_THREADS;
    constexpr int OVT = D * (SLAB / VEC), NOV = (OVT + GF_THREADS - 1) / GF_THREADS;
    constexpr int NTP = (9 * 2 * SLAB + GF_THREADS - 1) / GF_THREADS;
    Vec16<T> wpre[NWV], opre[NOV];
    float tpre[NTP];
    auto wload = [&](int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + GF_THREADS * i;
            if (idx < WVT) {
                const int n = idx / (D / VEC), cv = idx % (D / VEC);
                const long row = n < SLAB ? s * SLAB + n : HP + s * SLAB + n - SLAB;
                wpre[i] = load16<T>(Wi + row * D + cv * VEC);
            }
        }
#pragma unroll
        for (int i = 0; i < NOV; ++i) {
            const int idx = tid + GF_THREADS * i;
            if (idx < OVT) opre[i] = load16<T>(Wo + (long)(idx / (SLAB / VEC)) * HP + s * SLAB + (idx % (SLAB / VEC)) * VEC);
        }
#pragma unroll
        for (int i = 0; i < NTP; ++i) {
            const int idx = tid + GF_THREADS * i;
            if (idx < 9 * 2 * SLAB) {
                const int c = idx % (2 * SLAB);
                tpre[i] = a.w9[(idx / (2 * SLAB)) * a.ldw + (c < SLAB ? s * SLAB + c : HP + s * SLAB + c - SLAB)];
            }
        }
    };
    auto wstore = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWV; ++i) {
            const int idx = tid + GF_THREADS * i;
            if (idx < WVT) store16<T>(Ws + (idx / (D / VEC)) * LDW + (idx % (D / VEC)) * VEC, wpre[i]);
        }
#pragma unroll
        for (int i = 0; i < NOV; ++i) {
            const int idx = tid + GF_THREADS * i;
            if (idx < OVT) store16<T>(Wos + buf * L::o_elems + (idx / (SLAB / VEC)) * LDO + (idx % (SLAB / VEC)) * VEC, opre[i]);
        }
#pragma unroll
        for (int i = 0; i < NTP; ++i) {
            const int idx = tid + GF_THREADS * i;
            if (idx < 9 * 2 * SLAB) tapsS[buf * 9 * 2 * SLAB + idx] = tpre[i];
        }
    };
    const int gsteps = tpw * nslab;
    wload(0);
    wstore(0);
    if (gsteps > 1) wload(1 % nslab);
    __syncthreads();

    for (int tile = sp * tpw; tile < (sp + 1) * tpw; ++tile) {
        const int ty0 = (tile / tilesx) * GF_TH, tx0 = (tile % tilesx) * GF_TW;

        // ---- this wave's halo pixels (row blocks wv, wv+8, ..), LayerNorm-ed, as MFMA fragments: the B operand of every slab
        typename TR::frag_t xf[GF_MBW][NKC];
        bool valid[GF_MBW];
#pragma unroll
        for (int mb = 0; mb < GF_MBW; ++mb) {
            const int r = (wv + GF_WAVES * mb) * 16 + (lane & 15);
            const int y = ty0 - 1 + r / GF_HW, x = tx0 - 1 + r % GF_HW;
            valid[mb] = r < GF_ROWS && y >= 0 && y < a.H && x >= 0 && x < a.W;
            const T* src = X + (img + (long)y * a.W + x) * a.ldx + EPL * (lane >> 4);
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                if (valid[mb]) xf[mb][kc] = *reinterpret_cast<const typename TR::frag_t*>(src + kc * KCH);
                else for (int e = 0; e < EPL; ++e) xf[mb][kc][e] = from_f32<T>(0.f);
            }
            // a row is spread over the 4 lanes l, l^16, l^32, l^48
            float s = 0.f;
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc)
                for (int e = 0; e < EPL; ++e) s += to_f32(xf[mb][kc][e]);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mean = s / (float)D;
            float d2 = 0.f;
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc)
                for (int e = 0; e < EPL; ++e) { const float d = to_f32(xf[mb][kc][e]) - mean; d2 += d * d; }
            d2 += __shfl_xor(d2, 16);
            d2 += __shfl_xor(d2, 32);
            const float rstd = rsqrtf(d2 / (float)D + 1e-5f);
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc)
                for (int e = 0; e < EPL; ++e) {
                    const int k = kc * KCH + EPL * (lane >> 4) + e;
                    xf[mb][kc][e] = from_f32<T>((to_f32(xf[mb][kc][e]) - mean) * rstd * a.ln_w[k] + a.ln_b[k]);
                }
        }

        // y tiles of this wave: 16-pixel block pb x output channels 16 NOBW och .. (rows = channels, columns = pixels)
        const int pb = wv % PXB, och = wv / PXB;
        f32x4 yacc[NOBW];
#pragma unroll
        for (int ob = 0; ob < NOBW; ++ob) yacc[ob] = f32x4{0.f, 0.f, 0.f, 0.f};
        // y[channels of this wave, pixel block pb] += W_out[:, slab] u^T with the u tile in LDS and the project_out stage `buf`
        auto ymma = [&](int buf) __attribute__((always_inline)) {
            const typename TR::frag_t uf = load_frag<T>(Us, LDU, pb * 16, 0);
            const T* wo = Wos + buf * L::o_elems;
#pragma unroll
            for (int ob = 0; ob < NOBW; ++ob) mma(yacc[ob], load_frag<T>(wo, LDO, (och * NOBW + ob) * 16, 0), uf);
        };

#pragma unroll 1
        for (int s = 0; s < nslab; ++s) {
            const int g = (tile - sp * tpw) * nslab + s, buf = g & 1;
            // ---- (1) t_s = W_s LN(a)^T  (transposed accumulators: rows = channels x1_s | x2_s, columns = pixels)
            f32x4 acc[GF_MBW][NB];
#pragma unroll
            for (int mb = 0; mb < GF_MBW; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
            {
                // waves that own one row block fewer take the shorter loop body: ONE scalar branch, none around the MFMAs
                auto stage = [&](auto nmb_c) __attribute__((always_inline)) {
                    constexpr int NMB = decltype(nmb_c)::value;
                    typename TR::frag_t wf[2][NB];
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) wf[0][nb] = load_frag<T>(Ws, LDW, nb * 16, 0);
#pragma unroll
                    for (int kk = 0; kk < NKC; ++kk) {
                        if (kk + 1 < NKC) {
#pragma unroll
                            for (int nb = 0; nb < NB; ++nb) wf[(kk + 1) & 1][nb] = load_frag<T>(Ws, LDW, nb * 16, (kk + 1) * KCH);
                        }
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                            for (int mb = 0; mb < NMB; ++mb) mma(acc[mb][nb], wf[kk & 1][nb], xf[mb][kk]);
                    }
                };
                if (wv + GF_WAVES * (GF_MBW - 1) < GF_MB) stage(std::integral_constant<int, GF_MBW>{});
                else stage(std::integral_constant<int, GF_MBW - 1>{});
            }
#pragma unroll
            for (int mb = 0; mb < GF_MBW; ++mb) {
                if (wv + GF_WAVES * mb >= GF_MB) continue;
                float* trow = Ts + ((wv + GF_WAVES * mb) * 16 + (lane & 15)) * LDT + (lane >> 4) * 4;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    *reinterpret_cast<f32x4*>(trow + nb * 16) = valid[mb] ? acc[mb][nb] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            // ---- (3, of the previous slab) project_out of u_{s-1}, in the same barrier interval as this slab's project_in
            if (s > 0) ymma(buf ^ 1);
            __syncthreads();              // t tile complete; the project_in stage, the previous project_out stage and u_{s-1} are dead

            if constexpr (KEEP) {          // training: the interior pixels' t values of this slab go to HBM as 16-byte chunks
                T* Tg = reinterpret_cast<T*>(a.Tout);
                for (int idx = tid; idx < GF_PIX * (2 * SLAB / VEC); idx += GF_THREADS) {
                    const int px = idx / (2 * SLAB / VEC), cv = idx % (2 * SLAB / VEC), ch = cv * VEC;
                    const float* src = Ts + ((px / GF_TW + 1) * GF_HW + px % GF_TW + 1) * LDT + ch;
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
                    Vec16<T> o;
                    for (int e = 0; e < 4; ++e) { o.set(e, v0[e]); o.set(4 + e, v1[e]); }
                    const long col = ch < SLAB ? s * SLAB + ch : HP + s * SLAB + ch - SLAB;
                    store16<T>(Tg + (img + (long)(ty0 + px / GF_TW) * a.W + tx0 + px % GF_TW) * a.ldt + col, o);
                }
            }
            // ---- (2) depthwise 3x3 + gate.  thread = (side, 4 channels, PPT pixels of one tile row); sides in adjacent lanes
            {
                const int side = tid & 1, it = tid >> 1;
                const int c4 = it & 7, rem = it >> 3, iy = rem / (GF_TW / PPT), ix0 = (rem % (GF_TW / PPT)) * PPT;
                const float* tsrc = Ts + (iy * GF_HW + ix0) * LDT + side * SLAB + c4 * 4;
                const float* wsrc = tapsS + buf * 9 * 2 * SLAB + side * SLAB + c4 * 4;
                f32x4 w[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const f32x4*>(wsrc + t * 2 * SLAB);
                auto tvec = [&](int r, int col) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(tsrc + (r * GF_HW + col) * LDT); };
                f32x4 cl[3], cm[3], cr[3], res[PPT];
#pragma unroll
                for (int r = 0; r < 3; ++r) { cl[r] = tvec(r, 0); cm[r] = tvec(r, 1); }
#pragma unroll
                for (int i = 0; i < PPT; ++i) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) cr[r] = tvec(r, i + 2);
                    f32x4 o = cl[0] * w[0];                            // nine taps = one multiply + eight fused multiply-adds
                    o = __builtin_elementwise_fma(cm[0], w[1], o);
                    o = __builtin_elementwise_fma(cr[0], w[2], o);
#pragma unroll
                    for (int r = 1; r < 3; ++r) {
                        o = __builtin_elementwise_fma(cl[r], w[r * 3], o);
                        o = __builtin_elementwise_fma(cm[r], w[r * 3 + 1], o);
                        o = __builtin_elementwise_fma(cr[r], w[r * 3 + 2], o);
                    }
                    res[i] = o;
#pragma unroll
                    for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
                }
                // the gelu side (even lane) finishes the first half of the pixels, the partner side (odd lane) the second half
                constexpr int HPX = PPT / 2;
#pragma unroll
                for (int i = 0; i < HPX; ++i) {
                    const f32x4 mine = side ? res[HPX + i] : res[i], send = side ? res[i] : res[HPX + i];
                    f32x4 other;
                    for (int e = 0; e < 4; ++e) other[e] = __shfl_xor(send[e], 1);
                    const f32x4 g_ = side ? other : mine, p_ = side ? mine : other;    // g_: gelu-side value, p_: partner
                    f32x4 o;
                    for (int e = 0; e < 4; ++e) o[e] = Math<T>::gelu(g_[e]) * p_[e];
                    store4<T>(Us + (iy * GF_TW + ix0 + (side ? HPX : 0) + i) * LDU + c4 * 4, o);
                }
            }
            if (g + 1 < gsteps) {         // next step's weights / taps to LDS, the step after into registers
                wstore(buf ^ 1);
                if (g + 2 < gsteps) wload((g + 2) % nslab);
            }
            __syncthreads();              // u tile complete; next step's weights in place
        }
        ymma(((tile - sp * tpw) * nslab + nslab - 1) & 1);        // project_out of the last slab

        // ---- y + a (one rounding), staged as [pixel][channel] over the u | t tiles, stored as whole 16-byte row chunks
        __syncthreads();                  // every wave is past its last read of u
        {
            const int px = pb * 16 + (lane & 15), cr = (och * NOBW) * 16 + (lane >> 4) * 4;
            const T* arow = X + (img + (long)(ty0 + px / GF_TW) * a.W + tx0 + px % GF_TW) * a.ldx + cr;
#pragma unroll
            for (int ob = 0; ob < NOBW; ++ob) store4<T>(Ys + px * LDY + ob * 16 + cr, yacc[ob] + load4<T>(arow + ob * 16));
        }
        __syncthreads();
        {
            constexpr int NV = D / VEC;
            for (int idx = tid; idx < GF_PIX * NV; idx += GF_THREADS) {
                const int px = idx / NV, c = (idx % NV) * VEC;
                store16<T>(Y + (img + (long)(ty0 + px / GF_TW) * a.W + tx0 + px % GF_TW) * a.ldy + c, load16<T>(Ys + px * LDY + c));
            }
        }
        __syncthreads();                  // the next tile's first t rows land on the output tile
    }
}

template <class T, int D, bool KEEP>
static int launch_gdfn_k(const GdfnDev& d, hipStream_t s) {
    const size_t shmem = GfLds<T, D>::bytes;
    allow_big_lds(gdfn_fused_kernel<T, D, KEEP>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GDFN_FUSED, (gdfn_fused_kernel<T, D, KEEP>), dim3(d.B * d.nsplit), dim3(GF_THREADS), shmem, s, d);
    return MPHSIR_OK;
}
template <class T, int D>
static int launch_gdfn(const GdfnDev& d, hipStream_t s) {
    return d.Tout ? launch_gdfn_k<T, D, true>(d, s) : launch_gdfn_k<T, D, false>(d, s);
}

template <class T> struct GdfnShapes {
    static bool has(int D) { return D == 64 || D == 128 || D == 192 || D == 256; }
    static int run(const GdfnDev& d, int D, hipStream_t s) {
        switch (D) {
            case 64: return launch_gdfn<T, 64>(d, s);
            case 128: return launch_gdfn<T, 128>(d, s);
            case 192: return launch_gdfn<T, 192>(d, s);
            case 256: return launch_gdfn<T, 256>(d, s);
        }
        return MPHSIR_EINVAL;
    }
};
template <> struct GdfnShapes<float> {      // 16-bit storage only: the fp32 parity path keeps the three-launch form
    static bool has(int) { return false; }
    static int run(const GdfnDev&, int, hipStream_t) { return MPHSIR_EINVAL; }
};

}  // namespace mphsir

extern "C" int mphsir_gdfn_fused_tile_width(int32_t D) { return D <= 128 ? 16 : 8; }

extern "C" int mphsir_gdfn_fused_fits(int32_t D, int32_t HP, int32_t H, int32_t W, int dtype) {
    using namespace mphsir;
    if (!MPHSIR_DTYPE_OK(dtype) || D <= 0 || HP <= 0 || HP % GF_SLAB != 0 || H <= 0 || W <= 0 || H % GF_TH != 0 || W % 16 != 0) return 0;
    return MPHSIR_DISPATCH_T(dtype, (GdfnShapes<T_>::has(D) ? 1 : 0));
}

extern "C" int mphsir_gdfn_fused(const mphsir_gdfn_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "gdfn_fused");
    MPHSIR_REQUIRE(a && a->X && a->ln_w && a->ln_b && a->Win && a->w9 && a->Wout && a->Y, "gdfn_fused: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "gdfn_fused: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(a->B > 0 && mphsir_gdfn_fused_fits(a->D, a->HP, a->H, a->W, dtype),
                   "gdfn_fused: (D=%d, HP=%d, H=%d, W=%d, dtype=%d) not covered (ask mphsir_gdfn_fused_fits; 16-bit types, HP %% 32 == 0, H %% 8 == 0, W %% 16 == 0)",
                   a->D, a->HP, a->H, a->W, dtype);
    const int tiles = (a->H / GF_TH) * (a->W / mphsir_gdfn_fused_tile_width(a->D));
    MPHSIR_REQUIRE(a->nsplit > 0 && tiles % a->nsplit == 0, "gdfn_fused: nsplit=%d must divide the %d pixel tiles (8 x %d pixels)", a->nsplit, tiles,
                   mphsir_gdfn_fused_tile_width(a->D));
    MPHSIR_REQUIRE(aligned16(a->X) && aligned16(a->Win) && aligned16(a->Wout) && aligned16(a->Y) && (a->ldx * 2) % 16 == 0 &&
                       (a->ldy * 2) % 16 == 0 && a->ldx >= a->D && a->ldy >= a->D && a->ldw >= 2 * a->HP,
                   "gdfn_fused: 16-byte alignment / row pitch");
    MPHSIR_REQUIRE(a->X != a->Y, "gdfn_fused: Y must not alias X (neighbouring tiles read X's halo)");
    if (a->T)
        MPHSIR_REQUIRE(aligned16(a->T) && (a->ldt * 2) % 16 == 0 && a->ldt >= 2 * a->HP, "gdfn_fused: T must be 16-byte aligned with ldt >= 2 HP");
    GdfnDev d{a->X, (long)a->ldx, a->ln_w, a->ln_b, a->Win, a->w9, (long)a->ldw, a->Wout, a->Y, (long)a->ldy, a->B, a->H, a->W, a->HP, a->nsplit,
              a->T, (long)a->ldt};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, (GdfnShapes<T_>::run(d, a->D, s)));
}
