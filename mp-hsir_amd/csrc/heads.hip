// The two ends of MP_HSIR_Net.forward (net/MP_HSIR.py:822-844) and the task-prompt algebra in front of it, as single launches:
//
//   nchw_to_cl        x_cl[b,p,c] = (T) inp[b,c,p] for c < C, 0 for C <= c < Cp      (`inp_img` -> the patch embedding's channels-last,
//                     32-padded input, :824; the same launch is the backward of the output head: d conv_out from d restored)
//   cl_to_nchw_add    out[b,c,p] = (float) y_cl[b,p,c] + inp[b,c,p]                   (`self.output(...) + inp_img`, :842-843)
//   task_weights      w[b,t] = #{k: ids[b,k] == t} / n                               (Text_Prompt.forward, training path :519-523:
//                     the mean of the one-hot rows of a sample's task ids)
//   mix_rows          O[i,d] = scale * sum_j A(i,j) Bm[j,d]                           (the (B,T)x(T,D) weighted means of :527 and TVSP :574
//                     -- (w.unsqueeze(-1) * table).mean(1) -- and their gradient d table = w^T dO / T; fp32, J <= a few dozen)
//
// Each replaces 3-6 framework launches of 4-20 us; none is bound by anything but its launch.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

constexpr int HEAD_PX = 64;      // pixels per workgroup tile

// X [B][C][HW] fp32 -> Y [B][HW][Cp] (T), zero for c >= C.  One workgroup = 64 pixels of one sample, all channels through LDS:
// reads coalesced along the pixels of a channel plane, writes along the channels of a pixel row.
template <class T>
__global__ __launch_bounds__(256) void nchw_to_cl_kernel(const float* __restrict__ X, T* __restrict__ Y, int C, long HW, int Cp) {
    HIP_DYNAMIC_SHARED(float, tile)                 // [Cp][HEAD_PX + 1]
    const long tiles = (HW + HEAD_PX - 1) / HEAD_PX;
    const long b = blockIdx.x / tiles, p0 = (blockIdx.x % tiles) * HEAD_PX;
    const int np = (int)((HW - p0) < HEAD_PX ? (HW - p0) : HEAD_PX);
    const float* src = X + b * C * HW + p0;
    for (int i = threadIdx.x; i < Cp * HEAD_PX; i += 256) {
        const int c = i / HEAD_PX, p = i % HEAD_PX;
        tile[c * (HEAD_PX + 1) + p] = (c < C && p < np) ? src[(long)c * HW + p] : 0.f;
    }
    __syncthreads();
    T* dst = Y + (b * HW + p0) * Cp;
    for (int i = threadIdx.x; i < np * Cp; i += 256) {
        const int p = i / Cp, c = i % Cp;
        dst[i] = from_f32<T>(tile[c * (HEAD_PX + 1) + p]);
    }
}

// Y [B][HW][ldy] (T) (+ R [B][C][HW] fp32) -> O [B][C][HW] fp32
template <class T>
__global__ __launch_bounds__(256) void cl_to_nchw_add_kernel(const T* __restrict__ Y, long ldy, const float* __restrict__ R, float* __restrict__ O,
                                                             int C, long HW) {
    HIP_DYNAMIC_SHARED(float, tile)                 // [C][HEAD_PX + 1]
    const long tiles = (HW + HEAD_PX - 1) / HEAD_PX;
    const long b = blockIdx.x / tiles, p0 = (blockIdx.x % tiles) * HEAD_PX;
    const int np = (int)((HW - p0) < HEAD_PX ? (HW - p0) : HEAD_PX);
    const T* src = Y + (b * HW + p0) * ldy;
    for (int i = threadIdx.x; i < np * C; i += 256) {
        const int p = i / C, c = i % C;
        tile[c * (HEAD_PX + 1) + p] = to_f32<T>(src[(long)p * ldy + c]);
    }
    __syncthreads();
    const long plane = b * C * HW + p0;
    for (int i = threadIdx.x; i < C * HEAD_PX; i += 256) {
        const int c = i / HEAD_PX, p = i % HEAD_PX;
        if (p < np) {
            const long o = plane + (long)c * HW + p;
            const float v = tile[c * (HEAD_PX + 1) + p];
            O[o] = R ? v + R[o] : v;
        }
    }
}

__global__ __launch_bounds__(256) void task_weights_kernel(const long long* __restrict__ ids, float* __restrict__ w, int B, int n, int T) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * T) return;
    const int b = i / T, t = i % T;
    int cnt = 0;
    for (int k = 0; k < n; ++k) cnt += ids[(long)b * n + k] == (long long)t;
    w[i] = (float)cnt / (float)n;
}

// O[i][d] = scale * sum_j A(i,j) Bm[j][d];  A(i,j) = A[i*J + j], or A[j*I + i] when transA
__global__ __launch_bounds__(256) void mix_rows_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ O,
                                                       int I, int J, int D, float scale, int transA) {
    const long n = (long)I * D;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
        const int i = (int)(e / D), d = (int)(e % D);
        float acc = 0.f;
        for (int j = 0; j < J; ++j) acc += (transA ? A[(long)j * I + i] : A[(long)i * J + j]) * Bm[(long)j * D + d];
        O[e] = acc * scale;
    }
}

}  // namespace mphsir

extern "C" int mphsir_nchw_to_cl(const float* X, void* Y, int32_t B, int32_t C, int64_t HW, int32_t Cp, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && Y && B > 0 && C > 0 && HW > 0 && Cp >= C, "nchw_to_cl: bad arguments");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "nchw_to_cl: dtype %d unsupported", dtype);
    const size_t shmem = (size_t)Cp * (HEAD_PX + 1) * sizeof(float);
    MPHSIR_REQUIRE(shmem <= 64 * 1024, "nchw_to_cl: Cp = %d exceeds the tile (at most 252 channels)", Cp);
    const long tiles = (HW + HEAD_PX - 1) / HEAD_PX;
    MPHSIR_REQUIRE((long)B * tiles < (1L << 31), "nchw_to_cl: too many tiles");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, ([&]() -> int {
        MPHSIR_LAUNCH(MPHSIR_K_LAYOUT, (nchw_to_cl_kernel<T_>), dim3((unsigned)(B * tiles)), dim3(256), shmem, s, X, reinterpret_cast<T_*>(Y), (int)C,
                      (long)HW, (int)Cp);
        return MPHSIR_OK; }()));
}

extern "C" int mphsir_cl_to_nchw_add(const void* Y, int64_t ldy, const float* R, float* O, int32_t B, int32_t C, int64_t HW, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(Y && O && B > 0 && C > 0 && HW > 0 && ldy >= C, "cl_to_nchw_add: bad arguments");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "cl_to_nchw_add: dtype %d unsupported", dtype);
    const size_t shmem = (size_t)C * (HEAD_PX + 1) * sizeof(float);
    MPHSIR_REQUIRE(shmem <= 64 * 1024, "cl_to_nchw_add: C = %d exceeds the tile (at most 252 channels)", C);
    const long tiles = (HW + HEAD_PX - 1) / HEAD_PX;
    MPHSIR_REQUIRE((long)B * tiles < (1L << 31), "cl_to_nchw_add: too many tiles");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, ([&]() -> int {
        MPHSIR_LAUNCH(MPHSIR_K_LAYOUT, (cl_to_nchw_add_kernel<T_>), dim3((unsigned)(B * tiles)), dim3(256), shmem, s, reinterpret_cast<const T_*>(Y),
                      (long)ldy, R, O, (int)C, (long)HW);
        return MPHSIR_OK; }()));
}

extern "C" int mphsir_task_weights(const int64_t* ids, float* w, int32_t B, int32_t n, int32_t T, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(ids && w && B > 0 && n > 0 && T > 0, "task_weights: bad arguments");
    MPHSIR_LAUNCH(MPHSIR_K_LAYOUT, task_weights_kernel, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                  reinterpret_cast<const long long*>(ids), w, (int)B, (int)n, (int)T);
    return MPHSIR_OK;
}

extern "C" int mphsir_mix_rows(const float* A, const float* Bm, float* O, int32_t I, int32_t J, int32_t D, float scale, int32_t transA, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(A && Bm && O && I > 0 && J > 0 && D > 0, "mix_rows: bad arguments");
    const long blocks = ((long)I * D + 255) / 256;
    MPHSIR_LAUNCH(MPHSIR_K_LAYOUT, mix_rows_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                  A, Bm, O, (int)I, (int)J, (int)D, scale, (int)transA);
    return MPHSIR_OK;
}
