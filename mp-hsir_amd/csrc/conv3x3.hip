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
