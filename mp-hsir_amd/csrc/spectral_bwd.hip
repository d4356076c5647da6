// spectral_dqkv_bwd: the middle of the backward of the global spectral (channel) attention in ONE launch.
//
// Reference: autograd of Spectral_Attention.forward net/MP_HSIR.py:96-113 (== Attention :301-322) through train.py:58-67.  With the
// folded form of the forward (out = M_b v, M_b = Wo blockdiag(softmax(normalised Gram))) the backward is
//     dv        = d_out M_b                                   (per-sample C x C matrix)
//     [dq | dk] = [q | k] W2_b^T                              (per-sample matrix from mphsir_spectral_fold_bwd: dG and the norm terms)
//     dt        = depthwise3x3^T([dq | dk | dv]),   d taps[c][tap] = sum_p t[c][p + tap] [dq|dk|dv][c][p]
// Until round 6 that was three launches -- two token GEMMs that wrote [dq | dk | dv] (3C values per token) to HBM and
// mphsir_dwconv3x3_bwd that read it back with its halo.  Here [dq | dk | dv] never leaves the chip: the structure is
// dwconv3x3_bwd_tile_kernel's (a persistent workgroup keeps the nine tap sums of its channels in registers while it walks 8x16-pixel
// tiles, one ordered LDS reduction at the end) with the halo tile PRODUCED by the matrix cores instead of loaded:
//   * a workgroup (4 waves) owns a slab of CS channels of ONE of q / k / v (CS = 64 / 48 / 32, a divisor of the head width, so a
//     q / k slab needs one head's q and k: K = 2 hd; a v slab needs all of d_out: K = C) over a contiguous range of tiles -- the
//     slabs of a range sit on one XCD (their input rows come out of that L2), and consecutive tiles are mostly one sample's, so the
//     slab's rows of the per-sample matrix go to LDS once per sample;
//   * per tile: the 180 halo pixels are 12 MFMA row blocks, three per wave; a wave reads the K-contiguous fragments of ITS pixels
//     straight from HBM / L2 (16 bytes per lane; the first four K chunks are requested a tile ahead, during the window pass), multiplies
//     them with the slab's weight fragments (LDS) into transposed accumulators (lane = 4 channels x 1 pixel) and stores those as
//     16-byte rows of the fp32 tile [pixel][channel] -- pixels outside the image produce zeros by themselves (their input fragment is 0);
//   * the window pass is the depthwise backward's: thread = 4 channels x a strip of 8 pixels, sliding 3x3 window over LDS, flipped
//     taps -> dt (8-byte stores), and  t[p] * window -> nine tap sums in registers.
// 16-bit types; H % 8 == 0, W % 16 == 0.
#include <stdlib.h>

#include <type_traits>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

constexpr int SB_TH = 8, SB_TW = 16, SB_HW = SB_TW + 2, SB_ROWS = (SB_TH + 2) * SB_HW;      // 180 halo pixels
constexpr int SB_MB = 12, SB_THREADS = 256, SB_WAVES = 4, SB_MBW = SB_MB / SB_WAVES;         // 12 row blocks, 3 per wave
constexpr int SB_G = 4;                                                                      // K chunks (of 32) in flight per wave and row block

struct SpecBwdDev {
    const void* QK; long ldqk; const void* DO; long lddo; const void* T; long ldt;
    const void* W2; const void* MbT;
    const float* w9; long ldw;
    void* dT; long lddt; float* part;
    int B, H, W, nblk, round_dall;
    const float* vscale;        // optional [B]: dv_b is multiplied by vscale[b] (DO handed over without its DropPath factor)
};

template <class T, int C, int HD> struct SbCfg {
    static constexpr int CS = HD % 64 == 0 ? 64 : (HD % 48 == 0 ? 48 : 32);      // channels per slab
    static constexpr int NB = CS / 16, NSLT = C / CS, NSLAB = 3 * NSLT, QPR = CS / 4;
    static constexpr int KQ = 2 * HD, KV = C, KMAX = KQ > KV ? KQ : KV;
    static constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    static constexpr int LDW = KMAX + PAD;           // weight rows [CS][K]: conflict-free ds_read_b128 fragments
    // pixels per window-pass thread: 8, as in the depthwise kernels -- but a 32-channel slab then keeps only 128 of the 256 threads busy
    // in that pass (8 channel quads x 16 strips): those slabs use strips of 4 pixels (32 strips: every thread busy, 4.5 instead of 3.75
    // LDS reads per output)
    static constexpr int SPX = CS == 32 ? 4 : 8, SPR = SB_TW / SPX, NST = SB_TH * SPR;
    // fp32 tile pitch.  8-pixel strips: = 4 (mod 8) floats -- the accumulator stores (16 pixels x 16 B) and the window reads (the two
    // strips of a 16-lane group are 8 pixels apart: 32 banks) both cover the 64 banks once.  4-pixel strips: = 8 (mod 16) floats -- the
    // two strips of a group are 4 pixels apart: 32 banks again (the accumulator stores of 16 pixels then collide pairwise: six per tile)
    static constexpr int LDT = SPX == 4 ? CS + 8 : CS + 4;
    static constexpr size_t t_floats0 = (size_t)SB_MB * 16 * LDT, red_floats = (size_t)NST * 9 * CS;
    static constexpr size_t t_floats = t_floats0 > red_floats ? t_floats0 : red_floats;      // the strip sums reuse the tile
    static constexpr size_t bytes = t_floats * sizeof(float) + (size_t)CS * LDW * sizeof(T);
    static_assert(HD % CS == 0 && C % CS == 0 && KQ % 32 == 0 && KV % 32 == 0 && QPR * NST <= SB_THREADS, "shape");
};

template <class T, int C, int HD>
__global__ __launch_bounds__(SB_THREADS, 2) void spectral_dqkv_bwd_kernel(SpecBwdDev a) {
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    typedef typename TR::vec4_t v4_t;
    typedef SbCfg<T, C, HD> CF;
    constexpr int CS = CF::CS, NB = CF::NB, NSLT = CF::NSLT, NSLAB = CF::NSLAB, QPR = CF::QPR, LDW = CF::LDW, LDT = CF::LDT;
    constexpr int SPX = CF::SPX, SPR = CF::SPR, NST = CF::NST;
    static_assert(sizeof(T) == 2, "16-bit types only");
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* Ts = reinterpret_cast<float*>(smem_v);                       // [192][LDT] fp32 [dq | dk | dv] of the halo tile, this slab; at the end: strip sums
    T* Ws = reinterpret_cast<T*>(Ts + CF::t_floats);                    // [CS][LDW]  the slab's rows of the sample's matrix, K gathered
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int L = (gridDim.x & 7) == 0 ? (int)xcd_contiguous_block() : (int)blockIdx.x;      // the slabs of a tile range share an XCD
    const int rg = L / NSLAB, slab = L % NSLAB, type = slab / NSLT, c0 = (slab % NSLT) * CS;  // type: 0 q, 1 k, 2 v; c0: channel inside it
    const int tilesx = a.W / SB_TW, tiles = (a.H / SB_TH) * tilesx;
    const long total = (long)a.B * tiles, per = (total + a.nblk - 1) / a.nblk;
    const long t_begin = (long)rg * per, t_end = t_begin + per < total ? t_begin + per : total;
    const int tcol = type * C + c0;                                     // column of T / dT / w9 / part
    const bool on = tid < QPR * NST;
    const int c4 = tid % QPR, st = tid / QPR, iy = st / SPR, ix0 = (st % SPR) * SPX;      // 4 channels x a strip of SPX pixels of tile row iy

    auto body = [&](auto isv_c) __attribute__((always_inline)) {
        constexpr bool ISV = decltype(isv_c)::value;
        constexpr int K = ISV ? CF::KV : CF::KQ, NKC = K / 32, NG = (NKC + SB_G - 1) / SB_G, GP = NKC < SB_G ? NKC : SB_G;
        const int hq0 = (c0 / HD) * HD;                                 // first channel of the slab's head inside q (inside k)
        const T* IN = reinterpret_cast<const T*>(ISV ? a.DO : a.QK);
        const long ldin = ISV ? a.lddo : a.ldqk;
        // element column of K index k (a multiple of 8) in the input rows AND in the rows of the sample's matrix: v: d_out / M_b^T
        // columns as they are; q, k: [q_h | k_h] of the head = columns h hd + k of the q part, C + h hd + (k - hd) of the k part
        auto kcol = [&](int k) __attribute__((always_inline)) { return ISV ? k : (k < HD ? hq0 + k : C + hq0 + (k - HD)); };
        const T* Wsrc = ISV ? reinterpret_cast<const T*>(a.MbT) : reinterpret_cast<const T*>(a.W2) + (long)(type * C) * 2 * C;
        constexpr long WLD = ISV ? C : 2 * C, WBS = ISV ? (long)C * C : (long)4 * C * C;      // row pitch / per-sample stride of the matrix

        // this lane's input row of row block mb of tile t: halo pixel r = 16 (wv + 4 mb) + (lane & 15); nullptr outside the image / the tile
        auto rowptr = [&](long t, int mb) __attribute__((always_inline)) -> const T* {
            const int b = (int)(t / tiles), tile = (int)(t % tiles);
            const int ty0 = (tile / tilesx) * SB_TH, tx0 = (tile % tilesx) * SB_TW;
            const int r = (wv + SB_WAVES * mb) * 16 + (lane & 15);
            const int y = ty0 - 1 + r / SB_HW, x = tx0 - 1 + r % SB_HW;
            if (r >= SB_ROWS || y < 0 || y >= a.H || x < 0 || x >= a.W) return nullptr;
            return IN + ((long)b * a.H * a.W + (long)y * a.W + x) * ldin;
        };
        auto ldfrag = [&](const T* row, int kc) __attribute__((always_inline)) -> frag_t {
            if (row) return *reinterpret_cast<const frag_t*>(row + kcol(32 * kc + 8 * (lane >> 4)));
            frag_t z;
#pragma unroll
            for (int e = 0; e < TR::EPL; ++e) z[e] = from_f32<T>(0.f);
            return z;
        };
        frag_t xf[SB_MBW][GP];          // the first GP K chunks of the tile's input fragments, requested a tile ahead
        v4_t xy[SPX];                   // t of this thread's SPX pixels x 4 channels
        auto gload = [&](long t) __attribute__((always_inline)) {
#pragma unroll
            for (int mb = 0; mb < SB_MBW; ++mb) {
                const T* row = rowptr(t, mb);
#pragma unroll
                for (int q = 0; q < GP; ++q) xf[mb][q] = ldfrag(row, q);
            }
            if (on) {
                const int b = (int)(t / tiles), tile = (int)(t % tiles);
                const int ty0 = (tile / tilesx) * SB_TH, tx0 = (tile % tilesx) * SB_TW;
                const T* ts = reinterpret_cast<const T*>(a.T) + ((long)b * a.H * a.W + (long)(ty0 + iy) * a.W + tx0 + ix0) * a.ldt + tcol + c4 * 4;
#pragma unroll
                for (int i = 0; i < SPX; ++i) xy[i] = *reinterpret_cast<const v4_t*>(ts + (long)i * a.ldt);
            }
        };
        // window position (r, c) holds dY[p + (r-1, c-1)] = dY[p - tap] for tap (1-r, 1-c): it meets the flipped tap 8 - (3r + c)
        f32x4 w[9], acc9[9];
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
            acc9[t9] = f32x4{0.f, 0.f, 0.f, 0.f};
            w[t9] = on ? *reinterpret_cast<const f32x4*>(a.w9 + (8 - t9) * a.ldw + tcol + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        int bprev = -1;
        if (t_begin < t_end) gload(t_begin);
        for (long t = t_begin; t < t_end; ++t) {
            const int b = (int)(t / tiles), tile = (int)(t % tiles);
            const int ty0 = (tile / tilesx) * SB_TH, tx0 = (tile % tilesx) * SB_TW;
            if (b != bprev) {           // (uniform) the slab's rows of this sample's matrix -> LDS; every wave is past the last tile's MFMAs
                const T* Wb = Wsrc + (long)b * WBS + (long)c0 * WLD;
                for (int idx = tid; idx < CS * (K / 8); idx += SB_THREADS) {
                    const int n = idx / (K / 8), k = (idx % (K / 8)) * 8;
                    store16<T>(Ws + n * LDW + k, load16<T>(Wb + (long)n * WLD + kcol(k)));
                }
                bprev = b;
                __syncthreads();
            }
            // ---- [dq | dk | dv]_slab = W_slab in^T on the matrix cores (transposed accumulators: rows = channels, columns = pixels)
            f32x4 acc[SB_MBW][NB];
#pragma unroll
            for (int mb = 0; mb < SB_MBW; ++mb)
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[mb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                frag_t xg[SB_MBW][SB_G];
#pragma unroll
                for (int mb = 0; mb < SB_MBW; ++mb) {
                    const T* row = g == 0 ? nullptr : rowptr(t, mb);
#pragma unroll
                    for (int q = 0; q < SB_G; ++q)
                        if (g * SB_G + q < NKC) xg[mb][q] = (g == 0) ? xf[mb][q < GP ? q : 0] : ldfrag(row, g * SB_G + q);
                }
#pragma unroll
                for (int q = 0; q < SB_G; ++q) {
                    const int kc = g * SB_G + q;
                    if (kc < NKC) {
#pragma unroll
                        for (int nb = 0; nb < NB; ++nb) {
                            const frag_t wf = load_frag<T>(Ws, LDW, nb * 16, kc * 32);
#pragma unroll
                            for (int mb = 0; mb < SB_MBW; ++mb) mma(acc[mb][nb], wf, xg[mb][q]);
                        }
                    }
                }
            }
#pragma unroll
            for (int mb = 0; mb < SB_MBW; ++mb) {
                float* trow = Ts + ((wv + SB_WAVES * mb) * 16 + (lane & 15)) * LDT + (lane >> 4) * 4;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    f32x4 o = acc[mb][nb];
                    if (ISV && a.vscale) o *= a.vscale[b];
                    if (a.round_dall) {      // tests: the values the three-launch path would have read back from its 16-bit [dq | dk | dv]
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = to_f32(from_f32<T>(o[e]));
                    }
                    *reinterpret_cast<f32x4*>(trow + nb * 16) = o;
                }
            }
            f32x4 xin[SPX];
#pragma unroll
            for (int i = 0; i < SPX; ++i) xin[i] = f32x4{to_f32(xy[i][0]), to_f32(xy[i][1]), to_f32(xy[i][2]), to_f32(xy[i][3])};
            __syncthreads();            // the tile is complete (and every wave is done with Ws for this tile)
            if (t + 1 < t_end) gload(t + 1);
            if (on) {
                const float* tsrc = Ts + (iy * SB_HW + ix0) * LDT + c4 * 4;
                auto tvec = [&](int r, int col) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(tsrc + (r * SB_HW + col) * LDT); };
                T* xdst = reinterpret_cast<T*>(a.dT) + ((long)b * a.H * a.W + (long)(ty0 + iy) * a.W + tx0 + ix0) * a.lddt + tcol + c4 * 4;
                f32x4 cl[3], cm[3], cr[3];
#pragma unroll
                for (int r = 0; r < 3; ++r) { cl[r] = tvec(r, 0); cm[r] = tvec(r, 1); }
#pragma unroll
                for (int i = 0; i < SPX; ++i) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) cr[r] = tvec(r, i + 2);
                    f32x4 o = cl[0] * w[0];      // the order of dwconv3x3_bwd_tile_kernel (bitwise the same dt for the same window)
                    o = __builtin_elementwise_fma(cm[0], w[1], o);
                    o = __builtin_elementwise_fma(cr[0], w[2], o);
#pragma unroll
                    for (int r = 1; r < 3; ++r) {
                        o = __builtin_elementwise_fma(cl[r], w[r * 3], o);
                        o = __builtin_elementwise_fma(cm[r], w[r * 3 + 1], o);
                        o = __builtin_elementwise_fma(cr[r], w[r * 3 + 2], o);
                    }
                    store4<T>(xdst + (long)i * a.lddt, o);
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        acc9[r * 3] = __builtin_elementwise_fma(cl[r], xin[i], acc9[r * 3]);
                        acc9[r * 3 + 1] = __builtin_elementwise_fma(cm[r], xin[i], acc9[r * 3 + 1]);
                        acc9[r * 3 + 2] = __builtin_elementwise_fma(cr[r], xin[i], acc9[r * 3 + 2]);
                    }
#pragma unroll
                    for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
                }
            }
            __syncthreads();            // the tile is free for the next one (and, after the last, for the strip sums)
        }
        // ---- the nine tap sums of the slab's channels: ordered sum over the NST strips -> this range's partial, tap-major in the
        // parameter's tap order (ranges without tiles write zeros: every partial is summed)
        float* red = Ts;                // [NST][9][CS]
        if (on)
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) *reinterpret_cast<f32x4*>(red + (st * 9 + tp) * CS + c4 * 4) = acc9[8 - tp];
        __syncthreads();
        for (int i = tid; i < 9 * CS; i += SB_THREADS) {
            const int tp = i / CS, c = i % CS;
            float s = 0.f;
            for (int k = 0; k < NST; ++k) s += red[(k * 9 + tp) * CS + c];
            a.part[((long)rg * 9 + tp) * (3 * C) + tcol + c] = s;
        }
    };
    if (type == 2) body(std::true_type{});
    else body(std::false_type{});
}

template <class T, int C, int HD>
static int launch_spec_bwd(const SpecBwdDev& d, hipStream_t s) {
    typedef SbCfg<T, C, HD> CF;
    allow_big_lds(spectral_dqkv_bwd_kernel<T, C, HD>, CF::bytes);
    MPHSIR_LAUNCH(MPHSIR_K_SPECTRAL_DQKV_BWD, (spectral_dqkv_bwd_kernel<T, C, HD>), dim3(d.nblk * CF::NSLAB), dim3(SB_THREADS), CF::bytes, s, d);
    return MPHSIR_OK;
}

static bool spec_bwd_shape(int C, int HD) {
    return (HD == 32 && (C == 32 || C == 64 || C == 128 || C == 256)) || (HD == 64 && C == 128) || (HD == 48 && (C == 96 || C == 192 || C == 384)) ||
           (HD == 96 && C == 192);
}
template <class T> struct SpecBwdShapes {
    static int run(const SpecBwdDev& d, int C, int HD, hipStream_t s) {
#define MPHSIR_SB_CASE(c, hd) if (C == c && HD == hd) return launch_spec_bwd<T, c, hd>(d, s);
        MPHSIR_SB_CASE(32, 32) MPHSIR_SB_CASE(64, 32) MPHSIR_SB_CASE(128, 32) MPHSIR_SB_CASE(256, 32) MPHSIR_SB_CASE(128, 64)
        MPHSIR_SB_CASE(96, 48) MPHSIR_SB_CASE(192, 48) MPHSIR_SB_CASE(384, 48) MPHSIR_SB_CASE(192, 96)
#undef MPHSIR_SB_CASE
        return MPHSIR_EINVAL;
    }
};
template <> struct SpecBwdShapes<float> {
    static int run(const SpecBwdDev&, int, int, hipStream_t) { return MPHSIR_EINVAL; }
};

}  // namespace mphsir

extern "C" int mphsir_spectral_dqkv_bwd_fits(int32_t C, int32_t heads, int32_t H, int32_t W, int dtype) {
    if (!(dtype == MPHSIR_BF16 || dtype == MPHSIR_F16) || heads <= 0 || C <= 0 || C % heads != 0 || H <= 0 || W <= 0) return 0;
    if (H % mphsir::SB_TH != 0 || W % mphsir::SB_TW != 0) return 0;
    return mphsir::spec_bwd_shape(C, C / heads) ? 1 : 0;
}

extern "C" int mphsir_spectral_dqkv_bwd_slabs(int32_t C, int32_t heads) {
    if (heads <= 0 || C <= 0 || C % heads != 0) return MPHSIR_EINVAL;
    const int HD = C / heads, CS = HD % 64 == 0 ? 64 : (HD % 48 == 0 ? 48 : 32);
    return 3 * C / CS;
}

extern "C" int mphsir_spectral_dqkv_bwd(const mphsir_spectral_bwd_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "spectral_dqkv_bwd");
    MPHSIR_REQUIRE(a->QK && a->DO && a->T && a->W2 && a->MbT && a->w9 && a->dT && a->part, "spectral_dqkv_bwd: null pointer");
    MPHSIR_REQUIRE(mphsir_spectral_dqkv_bwd_fits(a->C, a->heads, a->H, a->W, dtype),
                   "spectral_dqkv_bwd: (C=%d, heads=%d, H=%d, W=%d, dtype=%d) not covered (16-bit types, H %% 8 == 0, W %% 16 == 0: ask "
                   "mphsir_spectral_dqkv_bwd_fits)", a->C, a->heads, a->H, a->W, dtype);
    MPHSIR_REQUIRE(a->B > 0 && a->nblk > 0, "spectral_dqkv_bwd: B=%d, nblk=%d", a->B, a->nblk);
    MPHSIR_REQUIRE(aligned16(a->QK) && aligned16(a->DO) && aligned16(a->W2) && aligned16(a->MbT) && aligned16(a->w9) && (a->ldqk * 2) % 16 == 0 &&
                       (a->lddo * 2) % 16 == 0 && (a->ldt * 2) % 8 == 0 && (a->lddt * 2) % 8 == 0 && (a->ldw * 4) % 16 == 0 &&
                       (reinterpret_cast<uintptr_t>(a->T) & 7) == 0 && (reinterpret_cast<uintptr_t>(a->dT) & 7) == 0,
                   "spectral_dqkv_bwd: alignment (16 bytes for QK / DO / W2 / MbT / w9 rows, 8 bytes for T / dT rows)");
    MPHSIR_REQUIRE(a->ldqk >= 2 * a->C && a->lddo >= a->C && a->ldt >= 3 * a->C && a->lddt >= 3 * a->C && a->ldw >= 3 * a->C, "spectral_dqkv_bwd: row pitch");
    MPHSIR_REQUIRE(a->dT != a->T, "spectral_dqkv_bwd: dT must not alias T");
    SpecBwdDev d{a->QK, (long)a->ldqk, a->DO, (long)a->lddo, a->T, (long)a->ldt, a->W2, a->MbT, a->w9, (long)a->ldw, a->dT, (long)a->lddt, a->part,
                 a->B, a->H, a->W, a->nblk, a->round_dall, a->vscale};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, (SpecBwdShapes<T_>::run(d, a->C, a->C / a->heads, s)));
}
