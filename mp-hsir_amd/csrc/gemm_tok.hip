// gemm_tok: Y[M][N] = epi( pro(X)[M][K] @ W[N][K]^T ) on channels-last token matrices.
//
// Every 1x1 conv / Linear of the MP-HSIR path that is not inside a larger fused kernel goes through
// here (reference call sites listed in include/mphsir.h).  One 256-thread workgroup computes a
// 64-token x (64*NW)-channel tile: the token rows go through two LDS stages in K-chunks (optionally
// LayerNorm-ed on the way in; the next chunk is already in flight in registers during the MFMAs, one
// barrier per chunk), each wave owns 16-channel column tiles and streams its weight rows straight
// from L2 as MFMA fragments (weights are tiny and shared by every workgroup).  The accumulators hold
// the transposed tile (channels x tokens), so a lane owns 4 consecutive channels of a token and the
// epilogue (bias / residual / branch sum) is a direct 8- or 16-byte load-modify-store.
#include <stdlib.h>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

constexpr int GT_BM = 64, GT_BN = 64, GT_KC = 64;

struct GemmDev {
    const void* X; long ldx;
    const void* W; long wbs; long rpb;
    const float* bias; const float* ln_w; const float* ln_b;
    void* Y; long ldy;
    int M, N, K;
    const void* R; long ldr;
    const void* SA; long ldsa;
    const float* gate; const float* keep;
    int H, Wimg, shift;
};

// Epilogue shared by both kernel forms: acc[w][mt] holds the TRANSPOSED 16x16 tiles (rows = output channels ntile + 64 w + ...,
// columns = tokens m0 + 16 mt + ...) of the 256 threads' four waves (wave wv owns channel tiles ntile = n0 + 16 wv, + 64 per w).
// `smem` is an LDS region no wave still reads as anything else (the callers' barrier rules are stated at the call sites).
template <class T, int EPI, int NW, class Sync>
__device__ __forceinline__ void gemm_tok_epilogue(const GemmDev& a, unsigned char* smem, f32x4 (&acc)[NW][4], int m0, int n0, int tid, int lane, int wv,
                                                  Sync sync) {
    typedef ElemTraits<T> TR;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr int VEC = Vec16<T>::N;
    const int ntile = n0 + wv * 16;
    if constexpr (EPI == 0) {
        // ---- epilogue: accumulators -> LDS (as [token][channel], reusing the staging tiles) -> whole 16-byte chunks of output
        // rows.  Storing straight from the transposed accumulators wrote 32-byte pieces of 16 different rows per instruction;
        // the write-heavy shapes (N = 3C from K = C) ran at 2.5 TB/s against 3.2-3.7 TB/s for the read-heavy ones.  Plain
        // stores only (EPI 0): the residual epilogues keep the one-rounding fp32 path below and are read-heavy anyway.
        constexpr int LDCS = GT_BN * NW + PAD;
        T* Cs = reinterpret_cast<T*>(smem);                                   // [64][LDCS]
        sync();                                                      // every wave is done with the staging tiles
    #pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (ntile + w * 64 >= a.N) continue;
            const int nl = wv * 16 + w * 64 + (lane >> 4) * 4;                // column inside the workgroup's tile
            f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n0 + nl);
    #pragma unroll
            for (int mt = 0; mt < 4; ++mt) store4<T>(Cs + (mt * 16 + (lane & 15)) * LDCS + nl, acc[w][mt] + bias4);
        }
        sync();
        T* Y = reinterpret_cast<T*>(a.Y);
        const int ncols = (a.N - n0) < GT_BN * NW ? (a.N - n0) : GT_BN * NW, cpr = ncols / VEC;      // 16-byte chunks per row
        for (int idx = tid; idx < 64 * cpr; idx += 256) {
            const int tok = idx / cpr, c = (idx % cpr) * VEC;
            const long m = m0 + tok;
            const int n = n0 + c;
            Vec16<T> v = load16<T>(Cs + tok * LDCS + c);
            store16<T>(Y + m * a.ldy + n, v);
        }
    } else if constexpr (NW <= 2 && sizeof(T) == 2) {
        // ---- residual / branch-sum epilogues, 16-bit types: acc + bias -> fp32 LDS tile [token][channel] -> a thread finishes 8
        // consecutive channels of a token: 16-byte loads of R (and SA), 16-byte store, still ONE rounding.  (Straight from the
        // transposed accumulators a lane moved 8 bytes of 16 different rows per instruction: those launches ran at 1.3-2.7 TB/s.)
        constexpr int LDF = GT_BN * NW + 4;
        float* Cf = reinterpret_cast<float*>(smem);                           // [64][LDF]
        sync();                                                      // every wave is done with the staging tiles
    #pragma unroll
        for (int w = 0; w < NW; ++w) {
            if (ntile + w * 64 >= a.N) continue;
            const int nl = wv * 16 + w * 64 + (lane >> 4) * 4;
            f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n0 + nl);
    #pragma unroll
            for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(Cf + (mt * 16 + (lane & 15)) * LDF + nl) = acc[w][mt] + bias4;
        }
        sync();
        T* Y = reinterpret_cast<T*>(a.Y);
        const T* R = reinterpret_cast<const T*>(a.R);
        const T* SA = reinterpret_cast<const T*>(a.SA);
        const int hw = EPI == 2 ? a.H * a.Wimg : 1;
        const int ncols = (a.N - n0) < GT_BN * NW ? (a.N - n0) : GT_BN * NW, cpr = ncols / VEC;      // 16-byte chunks per row
        for (int idx = tid; idx < 64 * cpr; idx += 256) {
            const int tok = idx / cpr, c = (idx % cpr) * VEC;
            const long m = m0 + tok;
            const int n = n0 + c;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(Cf + tok * LDF + c), v1 = *reinterpret_cast<const f32x4*>(Cf + tok * LDF + c + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const Vec16<T> r = load16<T>(R + m * a.ldr + n);
            Vec16<T> o;
            if (EPI == 1) {
                for (int e = 0; e < 8; ++e) o.set(e, v[e] + r.get(e));
            } else {
                const int b = (int)(m / hw), p = (int)(m % hw), y = p / a.Wimg, x = p % a.Wimg;
                const int ys = (y - a.shift + a.H) % a.H, xs = (x - a.shift + a.Wimg) % a.Wimg;   // shifted-frame coords
                const float* gp = a.gate + ((long)b * (hw / 64) + (ys >> 3) * (a.Wimg >> 3) + (xs >> 3)) * a.N + n;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
                const float kf = a.keep ? a.keep[b] : 1.f;
                const Vec16<T> sa = load16<T>(SA + m * a.ldsa + n);
                for (int e = 0; e < 8; ++e) o.set(e, r.get(e) + kf * (sa.get(e) * g[e] + v[e]));
            }
            store16<T>(Y + m * a.ldy + n, o);
        }
    } else {
        T* Y = reinterpret_cast<T*>(a.Y);
        const T* R = reinterpret_cast<const T*>(a.R);
        const T* SA = reinterpret_cast<const T*>(a.SA);
        const int hw = EPI == 2 ? a.H * a.Wimg : 1;
    #pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int n = ntile + w * 64 + (lane >> 4) * 4;        // 4 consecutive output channels of this lane
            if (ntile + w * 64 >= a.N) continue;
            f32x4 bias4 = f32x4{0.f, 0.f, 0.f, 0.f};
            if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + n);
    #pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const long m = m0 + mt * 16 + (lane & 15);
                f32x4 v = acc[w][mt] + bias4;
                if (EPI == 1) v += load4<T>(R + m * a.ldr + n);
                if (EPI == 2) {
                    const int b = (int)(m / hw), p = (int)(m % hw), y = p / a.Wimg, x = p % a.Wimg;
                    const int ys = (y - a.shift + a.H) % a.H, xs = (x - a.shift + a.Wimg) % a.Wimg;   // shifted-frame coords
                    const f32x4 g = *reinterpret_cast<const f32x4*>(a.gate + ((long)b * (hw / 64) + (ys >> 3) * (a.Wimg >> 3) + (xs >> 3)) * a.N + n);
                    const float kf = a.keep ? a.keep[b] : 1.f;
                    v = load4<T>(R + m * a.ldr + n) + kf * (load4<T>(SA + m * a.ldsa + n) * g + v);
                }
                store4<T>(Y + m * a.ldy + n, v);
            }
        }
    }
}

// NW = 16-column tiles per wave: one workgroup covers 64 tokens x 64*NW output channels, so the token tile is
// staged (and LayerNorm-ed) once for up to 256 outputs instead of once per 64.
template <class T, int EPI, bool LN, int NW>
__global__ __launch_bounds__(256) void gemm_tok_kernel(GemmDev a) {
    typedef ElemTraits<T> TR;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr int LDA = GT_KC + PAD;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)                                     // 16-byte aligned base
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem_v);
    T* As = reinterpret_cast<T*>(smem);                                   // [2 stages][64][LDA]
    constexpr size_t TILE_B = 2 * 64 * LDA * sizeof(T);
    float* stat = reinterpret_cast<float*>(smem + TILE_B);               // mean[64], rstd[64]

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int m0 = blockIdx.x * GT_BM, n0 = blockIdx.y * GT_BN * NW;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* W = reinterpret_cast<const T*>(a.W) + (a.wbs ? (long)(m0 / a.rpb) * a.wbs : 0);
    const int K = a.K;

    if (LN) {   // per-row mean / rstd over K: 4 adjacent lanes per token row, 16-byte loads, two passes (second hits L1)
        constexpr int VECL = Vec16<T>::N;
        const int r = tid >> 2, q = tid & 3, nv = K / VECL;
        const T* row = X + (long)(m0 + r) * a.ldx;
        float s = 0.f;
        for (int i = q; i < nv; i += 4) {
            const Vec16<T> v = load16<T>(row + i * VECL);
            for (int e = 0; e < VECL; ++e) s += v.get(e);
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float mean = s / (float)K;
        float d2 = 0.f;
        for (int i = q; i < nv; i += 4) {
            const Vec16<T> v = load16<T>(row + i * VECL);
            for (int e = 0; e < VECL; ++e) { const float d = v.get(e) - mean; d2 += d * d; }
        }
        d2 += __shfl_xor(d2, 1);
        d2 += __shfl_xor(d2, 2);
        if (q == 0) { stat[r] = mean; stat[64 + r] = rsqrtf(d2 / (float)K + 1e-5f); }
        __syncthreads();
    }

    const int ntile = n0 + wv * 16;            // this wave's first 16 output channels (+64 per extra tile)
    // accumulators hold the TRANSPOSED tile (rows = output channels, columns = tokens): a lane then owns 4 consecutive
    // channels of one token and the epilogue is a plain 8/16-byte load-modify-store per tile, no LDS round trip
    f32x4 acc[NW][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
        for (int i = 0; i < 4; ++i) acc[w][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int VEC = Vec16<T>::N;
    constexpr int NX = 64 * (GT_KC / VEC) / 256;              // 16-byte vectors per thread and K-chunk
    // The token tile goes global -> registers -> LDS in 64-wide K-chunks.  Two register sets keep the NEXT TWO chunks in
    // flight while the current one is multiplied (the kernel is HBM/latency bound: bytes in flight per workgroup are
    // what matters), two LDS stages leave one barrier per chunk.
    Vec16<T> xa[NX], xb[NX];
    auto gload = [&](Vec16<T> (&xr)[NX], int k0) __attribute__((always_inline)) {
        if (k0 >= K) return;
        const int kc = (K - k0) < GT_KC ? (K - k0) : GT_KC, vpr = kc / VEC;
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int v = tid + 256 * it;
            if (v < 64 * vpr) xr[it] = load16<T>(X + (long)(m0 + v / vpr) * a.ldx + k0 + (v % vpr) * VEC);
        }
    };
    auto step = [&](Vec16<T> (&xr)[NX], int k0, T* Ab) __attribute__((always_inline)) {
        const int kc = (K - k0) < GT_KC ? (K - k0) : GT_KC, vpr = kc / VEC;    // 32 or 64
        // this chunk's weight fragments (straight from L2) are requested before the barrier: their latency overlaps the
        // LDS stage of the token tile instead of stalling the first MFMA of every K-step
        constexpr int NKK = GT_KC / TR::KCHUNK;
        typename TR::frag_t wfr[NKK][NW];
#pragma unroll
        for (int q = 0; q < NKK; ++q)
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (q * TR::KCHUNK < kc && ntile + w * 64 < a.N) wfr[q][w] = load_frag<T>(W, K, ntile + w * 64, k0 + q * TR::KCHUNK);
#pragma unroll
        for (int it = 0; it < NX; ++it) {                      // registers (-> LayerNorm) -> LDS stage
            const int v = tid + 256 * it;
            if (v < 64 * vpr) {
                const int r = v / vpr, c = (v % vpr) * VEC;
                Vec16<T> x = xr[it];
                if (LN) {
                    const float mean = stat[r], rstd = stat[64 + r];
                    for (int i = 0; i < VEC; ++i)
                        x.set(i, (x.get(i) - mean) * rstd * a.ln_w[k0 + c + i] + a.ln_b[k0 + c + i]);
                }
                store16<T>(Ab + r * LDA + c, x);
            }
        }
        __syncthreads();           // one barrier per chunk: the other stage was last read before the previous barrier
        gload(xr, k0 + 2 * GT_KC);
#pragma unroll
        for (int q = 0; q < NKK; ++q) {
            if (q * TR::KCHUNK >= kc) break;
            typename TR::frag_t af[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = load_frag<T>(Ab, LDA, mt * 16, q * TR::KCHUNK);
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                if (ntile + w * 64 < a.N) {            // wave-uniform
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) mma(acc[w][mt], wfr[q][w], af[mt]);
                }
            }
        }
    };
    gload(xa, 0);
    gload(xb, GT_KC);
    for (int k0 = 0; k0 < K; k0 += 2 * GT_KC) {
        step(xa, k0, As);
        if (k0 + GT_KC < K) step(xb, k0 + GT_KC, As + 64 * LDA);
    }

    gemm_tok_epilogue<T, EPI, NW>(a, smem, acc, m0, n0, tid, lane, wv, [] { __syncthreads(); });
}

// ---- ring form (16-bit types, no LayerNorm prologue): persistent workgroups, a loader wave and four MFMA waves ------------------
// The kernel above is a workgroup per 64-token tile: load -> LDS -> MFMA -> LDS -> store, phases separated by barriers, ~3 workgroups
// per CU whose phases do not overlap (per CU the HBM, LDS, weight-fragment and matrix-pipe times ADD).  Here a
// workgroup is persistent -- it walks token tiles blockIdx.x, + gridDim.x, ... -- and FIVE waves: wave 4 does nothing but move
// token rows: K-chunks of 64 tokens x 64 channels (8 KB = eight global_load_lds_dwordx4) into a ring of GR_RING slots, GR_RING - 1
// chunks ahead, ACROSS tile boundaries (the first chunks of the next tile are in flight while the MFMA waves finish and store the
// current one).  Its vmcnt queue holds nothing but those transfers, so a counted wait retires exactly one chunk per step, whatever
// loads and stores the compiler schedules in the other waves (vmcnt is per wave).  One LDS-only barrier per chunk hands the chunk
// over and frees the slot of the chunk before.  The image is the one ds_read_b128 fragment reads want: 128-byte rows, 16-byte chunk
// c of row r at position c ^ (r & 7) -- the swizzle sits in the per-lane SOURCE address (an LDS-DMA destination is lane-linear).
// Channels beyond K come from a page of zeros.
constexpr int GR_RING = 4;            // ring slots (8 KB each)
constexpr int GR_LEAD = GR_RING - 1;  // chunks in flight ahead of the one being multiplied
__device__ __attribute__((aligned(16))) const unsigned char g_tok_zero_page[16] = {0};

template <class T, int EPI, int NW>
__global__ __launch_bounds__(320) void gemm_tok_ring_kernel(GemmDev a) {
    typedef ElemTraits<T> TR;
    static_assert(sizeof(T) == 2, "ring form: 16-bit element types");
    constexpr int SLOT = 64 * 128;                                      // 64 tokens x 64 channels
    constexpr int EPI_SYNCS = (EPI == 0 || NW <= 2) ? 2 : 0;      // barriers inside gemm_tok_epilogue
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    unsigned char* ring = reinterpret_cast<unsigned char*>(smem_v);      // [GR_RING][SLOT]
    unsigned char* stage = ring + GR_RING * SLOT;                        // the epilogue's staging tile
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int K = a.K, NCH = (K + 63) / 64, mtiles = a.M / GT_BM;
    const int n0 = blockIdx.y * GT_BN * NW;
    const int mine = mtiles > (int)blockIdx.x ? (mtiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;     // tiles of this workgroup
    const int total = mine * NCH;                                        // chunks of this workgroup
    if (wv == 4) {
        // ---- loader wave
        const char* X = reinterpret_cast<const char*>(a.X);
        const char* zero = reinterpret_cast<const char*>(g_tok_zero_page);
        const int r8 = lane >> 3, pos = lane & 7;
        auto issue = [&](int g) __attribute__((always_inline)) {
            const int tl = g / NCH, c = g - tl * NCH;
            const long m0 = ((long)blockIdx.x + (long)tl * gridDim.x) * GT_BM;
            unsigned char* slot = ring + (g % GR_RING) * SLOT;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = 8 * j + r8, kk = c * 64 + 8 * (pos ^ (row & 7));
                const char* src = kk < K ? X + ((m0 + row) * a.ldx + kk) * 2 : zero;
                MPHSIR_LDS_DMA16P(src, slot + 1024 * j);
            }
        };
        for (int g = 0; g < GR_LEAD && g < total; ++g) issue(g);
        for (int g = 0; g < total; ++g) {
            if (g + GR_LEAD <= total) wait_vmcnt<8 * (GR_LEAD - 1)>();     // chunk g has landed; the younger ones stay in flight
            else wait_vmcnt<0>();
            lds_barrier();                                                 // hand-over of chunk g; chunk g - 1 is free
            if (g + GR_LEAD < total) issue(g + GR_LEAD);
            if ((g + 1) % NCH == 0)
                for (int e = 0; e < EPI_SYNCS; ++e) lds_barrier();         // the barriers of the MFMA waves' epilogue
        }
        return;
    }
    // ---- MFMA waves
    const T* Wb = reinterpret_cast<const T*>(a.W);
    const int ntile = n0 + wv * 16;
    constexpr int NKK = GT_KC / TR::KCHUNK;
    // weight rows of this wave, clamped into the matrix (tiles beyond N multiply into accumulators that are never stored)
    int wrow[NW];
#pragma unroll
    for (int w = 0; w < NW; ++w) wrow[w] = (ntile + w * 64 < a.N) ? ntile + w * 64 : a.N - 16;
    int g = 0;
    for (int tl = 0; tl < mine; ++tl) {
        const int m0 = ((int)blockIdx.x + tl * (int)gridDim.x) * GT_BM;
        const T* W = Wb + (a.wbs ? (long)(m0 / a.rpb) * a.wbs : 0);
        f32x4 acc[NW][4];
#pragma unroll
        for (int w = 0; w < NW; ++w)
            for (int i = 0; i < 4; ++i) acc[w][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < NCH; ++c, ++g) {
            const int k0 = c * 64, kc = (K - k0) < GT_KC ? (K - k0) : GT_KC;
            // this chunk's weight fragments (straight from L2) are requested before the barrier
            typename TR::frag_t wfr[NKK][NW];
#pragma unroll
            for (int q = 0; q < NKK; ++q)
#pragma unroll
                for (int w = 0; w < NW; ++w)
                    wfr[q][w] = load_frag<T>(W, K, wrow[w], (q * TR::KCHUNK < kc) ? k0 + q * TR::KCHUNK : k0);
            lds_barrier();
            const unsigned char* slot = ring + (g % GR_RING) * SLOT;
#pragma unroll
            for (int q = 0; q < NKK; ++q) {
                if (q * TR::KCHUNK >= kc) break;
                typename TR::frag_t af[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int row = mt * 16 + (lane & 15);
                    af[mt] = *reinterpret_cast<const typename TR::frag_t*>(slot + row * 128 + 16 * ((q * 4 + (lane >> 4)) ^ (row & 7)));
                }
#pragma unroll
                for (int w = 0; w < NW; ++w)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) mma(acc[w][mt], wfr[q][w], af[mt]);
            }
        }
        // the staging tile is this workgroup's own region beside the ring: the tile before was read out of it at least one chunk
        // barrier ago
        gemm_tok_epilogue<T, EPI, NW>(a, stage, acc, m0, n0, tid, lane, wv, [] { lds_barrier(); });
    }
}

template <class T, int EPI, int NW> constexpr size_t gemm_tok_ring_lds() {
    constexpr size_t st = EPI == 0 ? 64 * (size_t)(GT_BN * NW + LDS_PAD_BYTES / sizeof(T)) * sizeof(T) : (NW <= 2 ? 64 * (size_t)(GT_BN * NW + 4) * sizeof(float) : 0);
    return (size_t)GR_RING * 64 * 128 + st;
}

// workgroups per CU the ring form is launched with (LDS: 32 KB ring + up to 35 KB staging tile)
constexpr int GR_WG_PER_CU = 2;

template <class T, int EPI, int NW>
static int launch_gemm_ring(const GemmDev& d, hipStream_t s) {
    const int mtiles = d.M / GT_BM, ny = (d.N + GT_BN * NW - 1) / (GT_BN * NW);
    int gx = (256 * GR_WG_PER_CU) / ny;
    if (gx > mtiles) gx = mtiles;
    if (gx < 1) gx = 1;
    const size_t shmem = gemm_tok_ring_lds<T, EPI, NW>();
    allow_big_lds(gemm_tok_ring_kernel<T, EPI, NW>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GEMM_TOK, (gemm_tok_ring_kernel<T, EPI, NW>), dim3(gx, ny), dim3(320), shmem, s, d);
    return MPHSIR_OK;
}

template <class T, int EPI, bool LN, int NW>
static int launch_gemm_nw(const GemmDev& d, hipStream_t s) {
    dim3 grid(d.M / GT_BM, (d.N + GT_BN * NW - 1) / (GT_BN * NW));
    const size_t stage = 2 * 64 * (size_t)(GT_KC + LDS_PAD_BYTES / sizeof(T)) * sizeof(T) + 128 * sizeof(float);
    const size_t epil = EPI == 0 ? 64 * (size_t)(GT_BN * NW + LDS_PAD_BYTES / sizeof(T)) * sizeof(T)
                                            : ((NW <= 2 && sizeof(T) == 2) ? 64 * (size_t)(GT_BN * NW + 4) * sizeof(float) : 0);
    const size_t shmem = stage > epil ? stage : epil;
    allow_big_lds(gemm_tok_kernel<T, EPI, LN, NW>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GEMM_TOK, (gemm_tok_kernel<T, EPI, LN, NW>), grid, dim3(256), shmem, s, d);
    return MPHSIR_OK;
}

// Where the ring form is taken: 16-bit types, no LayerNorm prologue; by default (form 0) only for the shapes it measured faster on
// MI355X (tools/bench/bench_gemm_shapes.py, MPHSIR_TOK_FORM=1 / 2): the INPUT-heavy ones, K >= 3 N with at most 128 outputs and at least
// 512 token tiles (K = 384 -> 128: 38.9 -> 33.9 us at M = 131072, 12.7 -> 10.9 at 32768; K = 704 -> 128: 61.9 -> 53.5 / 21.3 ->
// 16.3).  Output-heavy shapes lose (128 -> 352: 41 -> 59 us; 64 -> 192: 17 -> 26): a persistent workgroup stores its tile while its
// MFMA waves wait, three independent workgroups per CU overlap their stores with each other's loads.
template <class T, int EPI, bool LN>
static bool gemm_ring_applies(const GemmDev& d, int form) {
    if (LN || sizeof(T) != 2 || form == 1) return false;
    return form == 2 || (d.K >= 3 * d.N && d.N <= 128 && d.M / GT_BM >= 512);
}

template <class T, int EPI, bool LN>
static int launch_gemm(const GemmDev& d, int form, hipStream_t s) {
    if constexpr (!LN && sizeof(T) == 2) {
        if (gemm_ring_applies<T, EPI, LN>(d, form)) {
            if (d.N > 128) return launch_gemm_ring<T, EPI, 4>(d, s);
            if (d.N > 64) return launch_gemm_ring<T, EPI, 2>(d, s);
            return launch_gemm_ring<T, EPI, 1>(d, s);
        }
    }
    // enough workgroups to fill the chip first, then as many output channels per staged token tile as possible
    const long mt = d.M / GT_BM;
    // the widest tile (token tile staged / LayerNorm-ed once per 64*NW outputs) that still leaves >= ~1024 workgroups
    constexpr long wg_min = 1024;
    auto wgs = [&](int nw) { return mt * ((d.N + 64 * nw - 1) / (64 * nw)); };
    if (d.N > 128 && (mt >= 1024 || wgs(4) >= wg_min)) return launch_gemm_nw<T, EPI, LN, 4>(d, s);
    if (d.N > 64 && (mt >= 512 || wgs(2) >= wg_min)) return launch_gemm_nw<T, EPI, LN, 2>(d, s);
    return launch_gemm_nw<T, EPI, LN, 1>(d, s);
}

template <class T>
static int dispatch_gemm(const GemmDev& d, int epi, bool ln, int form, hipStream_t s) {
    switch (epi * 2 + (ln ? 1 : 0)) {
        case 0: return launch_gemm<T, 0, false>(d, form, s);
        case 1: return launch_gemm<T, 0, true>(d, form, s);
        case 2: return launch_gemm<T, 1, false>(d, form, s);
        case 3: return launch_gemm<T, 1, true>(d, form, s);
        case 4: return launch_gemm<T, 2, false>(d, form, s);
        case 5: return launch_gemm<T, 2, true>(d, form, s);
    }
    return MPHSIR_EINVAL;
}

}  // namespace mphsir

extern "C" int mphsir_gemm_tok(const mphsir_gemm_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "gemm_tok");
    MPHSIR_REQUIRE(a && a->X && a->W && a->Y, "gemm_tok: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "gemm_tok: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2;
    MPHSIR_REQUIRE(a->M > 0 && a->M % 64 == 0, "gemm_tok: M=%lld must be a positive multiple of 64", (long long)a->M);
    MPHSIR_REQUIRE(a->N > 0 && a->N % 16 == 0, "gemm_tok: N=%lld must be a multiple of 16", (long long)a->N);
    MPHSIR_REQUIRE(a->K > 0 && a->K % 32 == 0, "gemm_tok: K=%lld must be a multiple of 32", (long long)a->K);
    MPHSIR_REQUIRE(a->epi >= 0 && a->epi <= 2, "gemm_tok: epi %d unknown", a->epi);
    MPHSIR_REQUIRE(aligned16(a->X) && aligned16(a->W) && aligned16(a->Y) && (a->ldx * esz) % 16 == 0 &&
                       (a->ldy * esz) % 16 == 0, "gemm_tok: X/W/Y must be 16-byte aligned with 16-byte row pitch");
    MPHSIR_REQUIRE((a->ln_w == nullptr) == (a->ln_b == nullptr), "gemm_tok: ln_w and ln_b go together");
    if (a->epi >= 1)
        MPHSIR_REQUIRE(a->R && aligned16(a->R) && (a->ldr * esz) % 16 == 0, "gemm_tok: epi %d needs an aligned R", a->epi);
    if (a->epi == 2) {
        MPHSIR_REQUIRE(a->SA && a->gate && aligned16(a->SA) && (a->ldsa * esz) % 16 == 0, "gemm_tok: epi 2 needs SA and gate");
        MPHSIR_REQUIRE(a->H > 0 && a->Wimg > 0 && a->H % 8 == 0 && a->Wimg % 8 == 0 && a->M % ((int64_t)a->H * a->Wimg) == 0,
                       "gemm_tok: epi 2 needs H,W multiples of 8 with M = B*H*W");
        MPHSIR_REQUIRE(a->shift == 0 || a->shift == 4, "gemm_tok: shift must be 0 or 4");
    }
    if (a->w_batch_stride)
        MPHSIR_REQUIRE(a->rows_per_batch > 0 && a->rows_per_batch % 64 == 0, "gemm_tok: rows_per_batch must be a multiple of 64");
    GemmDev d{a->X, (long)a->ldx, a->W, (long)a->w_batch_stride, (long)a->rows_per_batch, a->bias, a->ln_w, a->ln_b,
              a->Y, (long)a->ldy, (int)a->M, (int)a->N, (int)a->K, a->R, (long)a->ldr, a->SA, (long)a->ldsa,
              a->gate, a->keep, a->H, a->Wimg, a->shift};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    MPHSIR_REQUIRE(a->form >= 0 && a->form <= 2, "gemm_tok: form %d unknown", a->form);
    return MPHSIR_DISPATCH_T(dtype, (dispatch_gemm<T_>(d, a->epi, a->ln_w != nullptr, a->form, s)));
}
