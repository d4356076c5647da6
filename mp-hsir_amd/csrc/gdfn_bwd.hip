// gdfn_dw_bwd: the depthwise / gate middle of the GDFN backward in ONE launch.
//
// Reference: autograd of FFN.forward / FeedForward.forward (net/MP_HSIR.py:259-265 == :385-391) through train.py:58-67:
//     [x1 | x2] = dwconv3x3(t),  u = gelu(x1) * x2            (t = project_in(LN(a)), kept by the forward)
//     d x1 = du x2 gelu'(x1),  d x2 = du gelu(x1)             (du = dy W_out, a token GEMM)
//     dt = dwconv3x3^T([d x1 | d x2]),   d taps[c][tap] = sum_p t[c][p + tap] [d x1 | d x2][c][p]
// Until round 6 two launches: mphsir_dwconv_gate_bwd (recomputes the conv, writes u and [d x1 | d x2] -- 2 HP values per token) and
// mphsir_dwconv3x3_bwd (reads t again and [d x1 | d x2] back with its halo): 1.25 GB at the PromptFusion level-1 shape, both
// HBM-bound.  Here a persistent 256-thread workgroup owns a slab of 16 channel PAIRS (x1 channel c and its partner HP + c) over a
// contiguous range of 8x16-pixel tiles and keeps [d x1 | d x2] on the chip:
//   (1) the t tile with a TWO-pixel halo (12 x 20 pixels) goes to LDS as fp32 (requested a tile ahead);
//   (2) forward depthwise conv on the 10 x 18 pixels of the one-pixel halo (thread = side, 4 channels, strip of 6 pixels) -> x tile;
//   (3) the gate's backward on those 180 pixels in place (du read from HBM where the pixel is inside the image, else 0: the conv's
//       zero padding of [d x1 | d x2]); u of the 128 interior pixels leaves for the project_out weight gradient;
//   (4) transposed depthwise conv + tap gradients on the 128 interior pixels (thread = side, 4 channels, strip of 4 pixels; the
//       centre values of t come from the tile of (1)), nine tap sums per thread in registers over the whole tile range.
// The slabs of a tile range sit on one XCD (neighbouring slabs share 128-byte lines of t).  16-bit types, H % 8 == 0, W % 16 == 0,
// HP % 16 == 0.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

constexpr int GB_TH = 8, GB_TW = 16, GB_W1 = GB_TW + 2, GB_H1 = GB_TH + 2, GB_W2 = GB_TW + 4, GB_H2 = GB_TH + 4;
constexpr int GB_P1 = GB_H1 * GB_W1, GB_P2 = GB_H2 * GB_W2;      // 180 / 240 pixels
constexpr int GB_CP = 16, GB_LD = 2 * GB_CP + 4;                 // 16 pairs = 32 channels per slab; pitch = 4 (mod 8) floats
constexpr int GB_THREADS = 256;

struct GdfnBwdDev {
    const void* Tin; long ldt; const float* w9; long ldw; const void* dU; long lddu;
    void* U; long ldu; void* dT; long lddt; float* part;
    int B, H, W, HP, nblk, round_mid;
};

template <class T>
__global__ __launch_bounds__(GB_THREADS, 2) void gdfn_dw_bwd_kernel(GdfnBwdDev a) {
    constexpr int VEC = Vec16<T>::N, CP = GB_CP, LD = GB_LD;
    static_assert(sizeof(T) == 2, "16-bit types only");
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* T2 = reinterpret_cast<float*>(smem_v);          // [240][LD] t with its two-pixel halo: x1-side channels | x2-side channels
    float* Xs = T2 + GB_P2 * LD;                           // [180][LD] [x1 | x2], then [d x1 | d x2]; at the end: the strip sums
    const int tid = threadIdx.x;
    const int nslab = a.HP / CP;
    const int L = (gridDim.x & 7) == 0 ? (int)xcd_contiguous_block() : (int)blockIdx.x;
    const int rg = L / nslab, slab = L % nslab, c0 = slab * CP;
    const int tilesx = a.W / GB_TW, tiles = (a.H / GB_TH) * tilesx;
    const long total = (long)a.B * tiles, per = (total + a.nblk - 1) / a.nblk;
    const long t_begin = (long)rg * per, t_end = t_begin + per < total ? t_begin + per : total;
    const T* Tin = reinterpret_cast<const T*>(a.Tin);
    const T* dU = reinterpret_cast<const T*>(a.dU);

    // ---- (1) loader: 240 pixels x 4 vectors (x1 channels c0.., c0+8.. | x2 channels HP+c0.., HP+c0+8..)
    constexpr int NV = (GB_P2 * 4 + GB_THREADS - 1) / GB_THREADS;      // 4
    Vec16<T> tv[NV];
    auto gload = [&](long t) __attribute__((always_inline)) {
        const int b = (int)(t / tiles), tile = (int)(t % tiles);
        const int ty0 = (tile / tilesx) * GB_TH, tx0 = (tile % tilesx) * GB_TW;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + GB_THREADS * i, p = idx >> 2, v = idx & 3;
            const int y = ty0 - 2 + p / GB_W2, x = tx0 - 2 + p % GB_W2;
            if (p < GB_P2 && y >= 0 && y < a.H && x >= 0 && x < a.W)
                tv[i] = load16<T>(Tin + ((long)b * a.H * a.W + (long)y * a.W + x) * a.ldt + (v < 2 ? c0 + v * VEC : a.HP + c0 + (v - 2) * VEC));
            else tv[i] = Vec16<T>{};
        }
    };
    // ---- per-thread roles
    // pass 1: (side, 4 channels, strip of 6 of the 180 halo pixels): 2 x 4 x 30 = 240 items
    const bool on1 = tid < 240;
    const int s1 = tid & 1, q1 = (tid >> 1) & 3, st1 = tid >> 3, hy1 = st1 / 3, hx1 = (st1 % 3) * 6;
    // pass 2: (side, 4 channels, strip of 4 of the 128 interior pixels): 2 x 4 x 32 = 256 items
    const int s2 = tid & 1, q2 = (tid >> 1) & 3, st2 = tid >> 3, iy = st2 >> 2, ix0 = (st2 & 3) * 4;
    f32x4 wf[9], wb[9], acc9[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
        acc9[t9] = f32x4{0.f, 0.f, 0.f, 0.f};
        wf[t9] = *reinterpret_cast<const f32x4*>(a.w9 + t9 * a.ldw + s1 * a.HP + c0 + q1 * 4);             // forward taps of the pass-1 channels
        wb[t9] = *reinterpret_cast<const f32x4*>(a.w9 + (8 - t9) * a.ldw + s2 * a.HP + c0 + q2 * 4);       // flipped taps of the pass-2 channels
    }
    if (t_begin < t_end) gload(t_begin);
    for (long t = t_begin; t < t_end; ++t) {
        const int b = (int)(t / tiles), tile = (int)(t % tiles);
        const int ty0 = (tile / tilesx) * GB_TH, tx0 = (tile % tilesx) * GB_TW;
        const long img = (long)b * a.H * a.W;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + GB_THREADS * i, p = idx >> 2, v = idx & 3;
            if (p < GB_P2) {
                float* dst = T2 + p * LD + (v < 2 ? v * VEC : CP + (v - 2) * VEC);
                *reinterpret_cast<f32x4*>(dst) = f32x4{tv[i].get(0), tv[i].get(1), tv[i].get(2), tv[i].get(3)};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{tv[i].get(4), tv[i].get(5), tv[i].get(6), tv[i].get(7)};
            }
        }
        __syncthreads();
        if (t + 1 < t_end) gload(t + 1);
        // ---- (2) [x1 | x2] = dwconv3x3(t) on the 10 x 18 halo pixels (tap order of dwconv_gate_tile_kernel: bitwise its values)
        if (on1) {
            const float* tsrc = T2 + (hy1 * GB_W2 + hx1) * LD + s1 * CP + q1 * 4;
            auto tvec = [&](int r, int col) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(tsrc + (r * GB_W2 + col) * LD); };
            float* xdst = Xs + (hy1 * GB_W1 + hx1) * LD + s1 * CP + q1 * 4;
            f32x4 cl[3], cm[3], cr[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { cl[r] = tvec(r, 0); cm[r] = tvec(r, 1); }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
#pragma unroll
                for (int r = 0; r < 3; ++r) cr[r] = tvec(r, i + 2);
                f32x4 o = cl[0] * wf[0];
                o = __builtin_elementwise_fma(cm[0], wf[1], o);
                o = __builtin_elementwise_fma(cr[0], wf[2], o);
#pragma unroll
                for (int r = 1; r < 3; ++r) {
                    o = __builtin_elementwise_fma(cl[r], wf[r * 3], o);
                    o = __builtin_elementwise_fma(cm[r], wf[r * 3 + 1], o);
                    o = __builtin_elementwise_fma(cr[r], wf[r * 3 + 2], o);
                }
                *reinterpret_cast<f32x4*>(xdst + i * LD) = o;
#pragma unroll
                for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
            }
        }
        __syncthreads();
        // ---- (3) the gate's backward in place: item = (halo pixel, 4 pairs); u of the interior pixels -> HBM
        for (int it = tid; it < GB_P1 * 4; it += GB_THREADS) {
            const int p = it >> 2, q = it & 3, hy = p / GB_W1, hx = p % GB_W1;
            const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
            float* xp = Xs + p * LD + q * 4;
            const f32x4 g = *reinterpret_cast<const f32x4*>(xp), pr = *reinterpret_cast<const f32x4*>(xp + CP);
            f32x4 d1 = f32x4{0.f, 0.f, 0.f, 0.f}, d2 = d1;
            if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
                const long row = img + (long)y * a.W + x;
                const f32x4 du = load4<T>(dU + row * a.lddu + c0 + q * 4);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float ge, dge;
                    Math<T>::gelu_pair(g[e], ge, dge);
                    o[e] = ge * pr[e];
                    d1[e] = du[e] * pr[e] * dge;
                    d2[e] = du[e] * ge;
                }
                if (hy >= 1 && hy <= GB_TH && hx >= 1 && hx <= GB_TW) store4<T>(reinterpret_cast<T*>(a.U) + row * a.ldu + c0 + q * 4, o);
                if (a.round_mid) {           // tests: the values the two-launch path reads back from its 16-bit [d x1 | d x2]
#pragma unroll
                    for (int e = 0; e < 4; ++e) { d1[e] = to_f32(from_f32<T>(d1[e])); d2[e] = to_f32(from_f32<T>(d2[e])); }
                }
            }
            *reinterpret_cast<f32x4*>(xp) = d1;
            *reinterpret_cast<f32x4*>(xp + CP) = d2;
        }
        __syncthreads();
        // ---- (4) dt = dwconv3x3^T([d x1 | d x2]) and the tap sums on the interior pixels (order of dwconv3x3_bwd_tile_kernel)
        {
            const float* xsrc = Xs + (iy * GB_W1 + ix0) * LD + s2 * CP + q2 * 4;
            auto xvec = [&](int r, int col) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(xsrc + (r * GB_W1 + col) * LD); };
            const float* tc = T2 + ((iy + 2) * GB_W2 + ix0 + 2) * LD + s2 * CP + q2 * 4;
            T* ddst = reinterpret_cast<T*>(a.dT) + (img + (long)(ty0 + iy) * a.W + tx0 + ix0) * a.lddt + s2 * a.HP + c0 + q2 * 4;
            f32x4 cl[3], cm[3], cr[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { cl[r] = xvec(r, 0); cm[r] = xvec(r, 1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int r = 0; r < 3; ++r) cr[r] = xvec(r, i + 2);
                f32x4 o = cl[0] * wb[0];
                o = __builtin_elementwise_fma(cm[0], wb[1], o);
                o = __builtin_elementwise_fma(cr[0], wb[2], o);
#pragma unroll
                for (int r = 1; r < 3; ++r) {
                    o = __builtin_elementwise_fma(cl[r], wb[r * 3], o);
                    o = __builtin_elementwise_fma(cm[r], wb[r * 3 + 1], o);
                    o = __builtin_elementwise_fma(cr[r], wb[r * 3 + 2], o);
                }
                store4<T>(ddst + (long)i * a.lddt, o);
                const f32x4 tcen = *reinterpret_cast<const f32x4*>(tc + i * LD);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc9[r * 3] = __builtin_elementwise_fma(cl[r], tcen, acc9[r * 3]);
                    acc9[r * 3 + 1] = __builtin_elementwise_fma(cm[r], tcen, acc9[r * 3 + 1]);
                    acc9[r * 3 + 2] = __builtin_elementwise_fma(cr[r], tcen, acc9[r * 3 + 2]);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
            }
        }
        __syncthreads();
    }
    // ---- tap sums: ordered sum over the 32 strips -> this range's partial [9][2 HP] (tap-major, the parameter's tap order)
    float* red = Xs;                                       // [32][9][2 CP]   (32 * 9 * 32 floats = 36.9 KB <= T2 + Xs)
    float* redbase = T2;                                   // the sums need 9216 floats: T2 (8640) + Xs (6480) are contiguous
    (void)red;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) *reinterpret_cast<f32x4*>(redbase + ((st2 * 9 + tp) * 2 + s2) * CP + q2 * 4) = acc9[8 - tp];
    __syncthreads();
    for (int i = tid; i < 9 * 2 * CP; i += GB_THREADS) {
        const int tp = i / (2 * CP), sc = i % (2 * CP), side = sc / CP, c = sc % CP;
        float s = 0.f;
        for (int k = 0; k < 32; ++k) s += redbase[((k * 9 + tp) * 2 + side) * CP + c];
        a.part[((long)rg * 9 + tp) * (2 * a.HP) + side * a.HP + c0 + c] = s;
    }
}

}  // namespace mphsir

extern "C" int mphsir_gdfn_dw_bwd_fits(int32_t H, int32_t W, int32_t HP, int dtype) {
    return ((dtype == MPHSIR_BF16 || dtype == MPHSIR_F16) && H > 0 && W > 0 && H % mphsir::GB_TH == 0 && W % mphsir::GB_TW == 0 && HP > 0 &&
            HP % mphsir::GB_CP == 0) ? 1 : 0;
}

extern "C" int mphsir_gdfn_dw_bwd(const void* T, const float* w9, int64_t ldw, const void* dU, void* U, void* dT, float* partial, int32_t nblk,
                                  int32_t B, int32_t H, int32_t W, int32_t HP, int32_t round_mid, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(T && w9 && dU && U && dT && partial, "gdfn_dw_bwd: null pointer");
    MPHSIR_REQUIRE(mphsir_gdfn_dw_bwd_fits(H, W, HP, dtype), "gdfn_dw_bwd: (H=%d, W=%d, HP=%d, dtype=%d) not covered (16-bit types, H %% 8 == 0, W %% 16 == 0, "
                   "HP %% 16 == 0: ask mphsir_gdfn_dw_bwd_fits)", H, W, HP, dtype);
    MPHSIR_REQUIRE(B > 0 && nblk > 0 && ldw >= 2 * (int64_t)HP && (ldw * 4) % 16 == 0, "gdfn_dw_bwd: B=%d, nblk=%d, ldw=%ld", B, nblk, (long)ldw);
    MPHSIR_REQUIRE(aligned16(T) && aligned16(w9) && aligned16(dU) && aligned16(U) && aligned16(dT) && dT != T, "gdfn_dw_bwd: 16-byte alignment; dT must not alias T");
    GdfnBwdDev d{T, 2L * HP, w9, (long)ldw, dU, (long)HP, U, (long)HP, dT, 2L * HP, partial, B, H, W, HP, nblk, round_mid};
    const size_t shmem = (size_t)(GB_P2 + GB_P1) * GB_LD * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int grid = nblk * (HP / GB_CP);
    if (dtype == MPHSIR_BF16) {
        allow_big_lds(gdfn_dw_bwd_kernel<bf16_t>, shmem);
        MPHSIR_LAUNCH(MPHSIR_K_GDFN_GATE_BWD, (gdfn_dw_bwd_kernel<bf16_t>), dim3(grid), dim3(GB_THREADS), shmem, s, d);
    } else {
        allow_big_lds(gdfn_dw_bwd_kernel<f16_t>, shmem);
        MPHSIR_LAUNCH(MPHSIR_K_GDFN_GATE_BWD, (gdfn_dw_bwd_kernel<f16_t>), dim3(grid), dim3(GB_THREADS), shmem, s, d);
    }
    return MPHSIR_OK;
}
