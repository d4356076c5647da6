// conv3x3_tok: dense 3x3 convolution (stride 1, zero padding, no bias) on channels-last cubes as an
// implicit GEMM, plus the im2col gather its weight gradient needs.
//
// Replaces the reference's dense convs on the path: OverlapPatchEmbed.proj (net/MP_HSIR.py:458),
// Downsample/Upsample bodies (:436,:446), TVSP.conv_last (:566) and the output conv (:807).
//   Y[p][co] = sum_tap sum_ci X[p + tap][ci] * W[co][tap][ci]        (W packed [Cout][9*Cin], tap-major)
// Same tile structure as gemm_tok (64 pixels x 64 output channels per workgroup, K streamed through LDS
// in 32-wide chunks) -- the only difference is that the A tile of chunk (tap, ci0) is gathered from the
// shifted pixel rows (zeros outside the image).  The input gradient is the same kernel with spatially
// flipped, transposed weights; the weight gradient is dY^T im2col(X), a token-reduction GEMM
// (mphsir_gemm_tn) over the gathered matrix written by im2col3x3.  Deterministic (no split-K atomics).
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct ConvDev {
    const void* X; long ldx;   // [B*H*W][ldx], Cin channels used
    const void* W;             // [N][9*Cin]
    void* Y; long ldy;         // [B*H*W][ldy], N channels written
    int B, H, Wd, Cin, N;
};

template <class T>
__global__ __launch_bounds__(256) void conv3x3_tok_kernel(ConvDev a) {
    typedef ElemTraits<T> TR;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T), KC = 32, LDA = KC + PAD, LDC = 64 + 4;
    constexpr int VEC = Vec16<T>::N;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* As = reinterpret_cast<T*>(smem_v);          // [64][LDA]
    float* Cs = reinterpret_cast<float*>(smem_v);  // [64][LDC] (aliases As after the K loop)
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int HW = a.H * a.Wd, K = 9 * a.Cin;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* W = reinterpret_cast<const T*>(a.W);
    const int ntile = n0 + wv * 16;
    const bool active = ntile < a.N;
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int VPR = KC / VEC;
    for (int k0 = 0; k0 < K; k0 += KC) {
        const int tap = k0 / a.Cin, ci0 = k0 % a.Cin, dy = tap / 3 - 1, dx = tap % 3 - 1;
        for (int v = tid; v < 64 * VPR; v += 256) {
            const int r = v / VPR, c = (v % VPR) * VEC;
            const int m = m0 + r, b = m / HW, p = m % HW, y = p / a.Wd + dy, x = p % a.Wd + dx;
            Vec16<T> xv;
            if (y >= 0 && y < a.H && x >= 0 && x < a.Wd) xv = load16<T>(X + ((long)b * HW + (long)y * a.Wd + x) * a.ldx + ci0 + c);
            else for (int e = 0; e < VEC; ++e) xv.set(e, 0.f);
            store16<T>(As + r * LDA + c, xv);
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int kk = 0; kk < KC; kk += TR::KCHUNK) {
                const typename TR::frag_t wf = load_frag<T>(W, K, ntile, k0 + kk);
                for (int mt = 0; mt < 4; ++mt) mma(acc[mt], load_frag<T>(As, LDA, mt * 16, kk), wf);
            }
        }
        __syncthreads();
    }
    if (active)
        for (int mt = 0; mt < 4; ++mt)
            for (int r = 0; r < 4; ++r) Cs[(mt * 16 + (lane >> 4) * 4 + r) * LDC + wv * 16 + (lane & 15)] = acc[mt][r];
    __syncthreads();
    T* Y = reinterpret_cast<T*>(a.Y);
    constexpr int G = 64 / VEC;
    for (int idx = tid; idx < 64 * G; idx += 256) {
        const int r = idx / G, c = (idx % G) * VEC, n = n0 + c;
        if (n >= a.N) continue;
        Vec16<T> o;
        for (int e = 0; e < VEC; ++e) o.set(e, Cs[r * LDC + c + e]);
        store16<T>(Y + (long)(m0 + r) * a.ldy + n, o);
    }
}

// ---- pipelined form (16-bit types): the structure of gemm_tok.hip with a gathered token tile --------------------------------
// One workgroup = 64 pixels x 64*NW output channels; K = (tap, 32 or 64 input channels) chunks.  The chunk's pixel rows (shifted by
// the tap, zeros outside the image) go global -> registers -> LDS with the NEXT TWO chunks already in flight in registers and
// two LDS stages (one barrier per chunk); each wave owns NW 16-channel tiles whose weight fragments come straight from L2
// (requested before the barrier), accumulators hold the transposed tile (channels x pixels) so the output leaves through an
// LDS staging tile as whole 16-byte row chunks.  The first form above (one synchronous gather + two barriers per chunk, 64
// outputs per staged tile) ran at 230-380 TFLOP/s.
template <class T, int NW, int KC>
__global__ __launch_bounds__(256) void conv3x3_pipe_kernel(ConvDev a) {
    typedef ElemTraits<T> TR;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T), LDA = KC + PAD, VEC = Vec16<T>::N, VPR = KC / VEC, NX = 64 * VPR / 256, NKK = KC / TR::KCHUNK;
    static_assert(64 * VPR % 256 == 0 && KC % TR::KCHUNK == 0, "whole vectors per thread, whole MFMA K-steps per chunk");
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* As = reinterpret_cast<T*>(smem_v);          // [2 stages][64][LDA]; afterwards the output tile [64][64*NW + PAD]
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int bx = (gridDim.x & 7) == 0 ? (int)xcd_contiguous_block() : (int)blockIdx.x;     // neighbouring pixel tiles share halo rows: one L2
    const int m0 = bx * 64, n0 = blockIdx.y * 64 * NW;
    const int HW = a.H * a.Wd, K = 9 * a.Cin;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* W = reinterpret_cast<const T*>(a.W);
    const int ntile = n0 + wv * 16;
    // this thread's vectors of every chunk: pixel rows r + (256 / VPR) it of the tile, channels c .. c+7 of the chunk
    const int r = tid / VPR, c = (tid % VPR) * VEC;
    int py[NX], px[NX];
    const T* xrow[NX];
#pragma unroll
    for (int it = 0; it < NX; ++it) {
        const int m = m0 + r + (256 / VPR) * it, pb = m / HW, pp = m % HW;
        py[it] = pp / a.Wd;
        px[it] = pp % a.Wd;
        xrow[it] = X + ((long)pb * HW + (long)py[it] * a.Wd + px[it]) * a.ldx + c;
    }
    f32x4 acc[NW][4];
#pragma unroll
    for (int w = 0; w < NW; ++w)
        for (int i = 0; i < 4; ++i) acc[w][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    Vec16<T> xa[NX], xb[NX];
    auto gload = [&](Vec16<T> (&xr)[NX], int k0) __attribute__((always_inline)) {
        if (k0 >= K) return;
        const int tap = k0 / a.Cin, ci0 = k0 - tap * a.Cin, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int it = 0; it < NX; ++it) {
            const int y = py[it] + dy, x = px[it] + dx;
            if (y >= 0 && y < a.H && x >= 0 && x < a.Wd) xr[it] = load16<T>(xrow[it] + ((long)dy * a.Wd + dx) * a.ldx + ci0);
            else xr[it] = Vec16<T>{};
        }
    };
    auto step = [&](Vec16<T> (&xr)[NX], int k0, T* Ab) __attribute__((always_inline)) {
        typename TR::frag_t wfr[NKK][NW];
#pragma unroll
        for (int q = 0; q < NKK; ++q)
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (ntile + w * 64 < a.N) wfr[q][w] = load_frag<T>(W, K, ntile + w * 64, k0 + q * TR::KCHUNK);
#pragma unroll
        for (int it = 0; it < NX; ++it) store16<T>(Ab + (r + (256 / VPR) * it) * LDA + c, xr[it]);
        __syncthreads();           // one barrier per chunk: the other stage was last read before the previous barrier
        gload(xr, k0 + 2 * KC);
#pragma unroll
        for (int q = 0; q < NKK; ++q) {
            typename TR::frag_t af[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) af[mt] = load_frag<T>(Ab, LDA, mt * 16, q * TR::KCHUNK);
#pragma unroll
            for (int w = 0; w < NW; ++w)
                if (ntile + w * 64 < a.N) {    // wave-uniform
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) mma(acc[w][mt], wfr[q][w], af[mt]);
                }
        }
    };
    gload(xa, 0);
    gload(xb, KC);
    for (int k0 = 0; k0 < K; k0 += 2 * KC) {
        step(xa, k0, As);
        if (k0 + KC < K) step(xb, k0 + KC, As + 64 * LDA);
    }
    // ---- accumulators (channels x pixels: a lane owns 4 consecutive channels of a pixel) -> LDS [pixel][channel] -> 16-byte row chunks
    constexpr int LDCS = 64 * NW + PAD;
    T* Cs = reinterpret_cast<T*>(smem_v);
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        if (ntile + w * 64 >= a.N) continue;
        const int nl = wv * 16 + w * 64 + (lane >> 4) * 4;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) store4<T>(Cs + (mt * 16 + (lane & 15)) * LDCS + nl, acc[w][mt]);
    }
    __syncthreads();
    T* Y = reinterpret_cast<T*>(a.Y);
    const int ncols = (a.N - n0) < 64 * NW ? (a.N - n0) : 64 * NW, cpr = ncols / VEC;
    for (int idx = tid; idx < 64 * cpr; idx += 256) {
        const int tok = idx / cpr, cc = (idx % cpr) * VEC;
        store16<T>(Y + (long)(m0 + tok) * a.ldy + n0 + cc, load16<T>(Cs + tok * LDCS + cc));
    }
}

template <class T, int NW, int KC>
static int launch_conv_pipe(const ConvDev& d, hipStream_t s) {
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    const size_t stage = 2 * 64 * (size_t)(KC + PAD) * sizeof(T), epil = 64 * (size_t)(64 * NW + PAD) * sizeof(T);
    const size_t shmem = stage > epil ? stage : epil;
    dim3 grid((unsigned)((long)d.B * d.H * d.Wd / 64), (d.N + 64 * NW - 1) / (64 * NW));
    allow_big_lds(conv3x3_pipe_kernel<T, NW, KC>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_CONV3X3, (conv3x3_pipe_kernel<T, NW, KC>), grid, dim3(256), shmem, s, d);
    return MPHSIR_OK;
}

template <class T, int KC>
static int launch_conv16_kc(const ConvDev& d, hipStream_t s) {
    // as many output channels per gathered pixel tile as possible while >= ~1024 workgroups remain (gemm_tok's rule)
    const long mt = (long)d.B * d.H * d.Wd / 64;
    auto wgs = [&](int nw) { return mt * ((d.N + 64 * nw - 1) / (64 * nw)); };
    if (d.N > 128 && (mt >= 1024 || wgs(4) >= 1024)) return launch_conv_pipe<T, 4, KC>(d, s);
    if (d.N > 64 && (mt >= 512 || wgs(2) >= 1024)) return launch_conv_pipe<T, 2, KC>(d, s);
    return launch_conv_pipe<T, 1, KC>(d, s);
}

template <class T>
static int launch_conv16(const ConvDev& d, hipStream_t s) {
    // a chunk never straddles two taps: 64-wide chunks (two MFMA K-steps per barrier) where the input width allows them
    return d.Cin % 64 == 0 ? launch_conv16_kc<T, 64>(d, s) : launch_conv16_kc<T, 32>(d, s);
}

struct ColDev {
    const void* X; long ldx; void* Col; int B, H, Wd, Cin;
};

// Col[p][tap*Cin + ci] = X[p + tap][ci] (zeros outside the image)
template <class T>
__global__ __launch_bounds__(256) void im2col3x3_kernel(ColDev a) {
    constexpr int VEC = Vec16<T>::N;
    const int cv = a.Cin / VEC, HW = a.H * a.Wd;
    const long total = (long)a.B * HW * 9 * cv;
    const T* X = reinterpret_cast<const T*>(a.X);
    T* Col = reinterpret_cast<T*>(a.Col);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % cv) * VEC;
        const long q = idx / cv;
        const int tap = (int)(q % 9);
        const long m = q / 9;
        const int b = (int)(m / HW), p = (int)(m % HW), y = p / a.Wd + tap / 3 - 1, x = p % a.Wd + tap % 3 - 1;
        Vec16<T> xv;
        if (y >= 0 && y < a.H && x >= 0 && x < a.Wd) xv = load16<T>(X + ((long)b * HW + (long)y * a.Wd + x) * a.ldx + c);
        else for (int e = 0; e < VEC; ++e) xv.set(e, 0.f);
        store16<T>(Col + m * 9 * a.Cin + tap * a.Cin + c, xv);
    }
}

}  // namespace mphsir

extern "C" int mphsir_conv3x3_tok(const void* X, int64_t ldx, const void* W, void* Y, int64_t ldy, int32_t B, int32_t H, int32_t Wd,
                                  int32_t Cin, int32_t N, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && W && Y, "conv3x3_tok: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "conv3x3_tok: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2;
    MPHSIR_REQUIRE(B > 0 && H > 0 && Wd > 0 && ((long)B * H * Wd) % 64 == 0, "conv3x3_tok: B*H*W must be a multiple of 64");
    MPHSIR_REQUIRE(Cin > 0 && Cin % 32 == 0 && N > 0 && N % 16 == 0, "conv3x3_tok: Cin %% 32 and N %% 16 must be 0 (pad the channels)");
    MPHSIR_REQUIRE(aligned16(X) && aligned16(W) && aligned16(Y) && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0, "conv3x3_tok: 16-byte alignment required");
    ConvDev d{X, (long)ldx, W, Y, (long)ldy, B, H, Wd, Cin, N};
    dim3 grid((unsigned)((long)B * H * Wd / 64), (N + 63) / 64);
    const size_t shmem = 64 * (64 + 4) * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MPHSIR_BF16 && N % 8 == 0) return launch_conv16<bf16_t>(d, s);
    if (dtype == MPHSIR_F16 && N % 8 == 0) return launch_conv16<f16_t>(d, s);
    if (dtype == MPHSIR_F32)
        MPHSIR_LAUNCH(MPHSIR_K_CONV3X3, (conv3x3_tok_kernel<float>), grid, dim3(256), shmem, s, d);
    else if (dtype == MPHSIR_BF16)
        MPHSIR_LAUNCH(MPHSIR_K_CONV3X3, (conv3x3_tok_kernel<bf16_t>), grid, dim3(256), shmem, s, d);
    else
        MPHSIR_LAUNCH(MPHSIR_K_CONV3X3, (conv3x3_tok_kernel<f16_t>), grid, dim3(256), shmem, s, d);
    return MPHSIR_OK;
}

extern "C" int mphsir_im2col3x3(const void* X, int64_t ldx, void* Col, int32_t B, int32_t H, int32_t Wd, int32_t Cin, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && Col, "im2col3x3: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "im2col3x3: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2, vec = 16 / esz;
    MPHSIR_REQUIRE(B > 0 && H > 0 && Wd > 0 && Cin > 0 && Cin % vec == 0, "im2col3x3: bad shape");
    MPHSIR_REQUIRE(aligned16(X) && aligned16(Col) && (ldx * esz) % 16 == 0, "im2col3x3: 16-byte alignment required");
    ColDev d{X, (long)ldx, Col, B, H, Wd, Cin};
    long blocks = ((long)B * H * Wd * 9 * (Cin / vec) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MPHSIR_F32)
        MPHSIR_LAUNCH(MPHSIR_K_IM2COL, (im2col3x3_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    else if (dtype == MPHSIR_BF16)
        MPHSIR_LAUNCH(MPHSIR_K_IM2COL, (im2col3x3_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    else
        MPHSIR_LAUNCH(MPHSIR_K_IM2COL, (im2col3x3_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    return MPHSIR_OK;
}
