// The two resamplers of TVSP.forward (net/MP_HSIR.py:572-583), channels-last, forward and backward:
//
//   tvsp_text_map   text[b,i,j,d] = L[b,d] * clip[floor(i*B/ps)][floor(j*512/ps)]              (:575-577)
//                   The reference multiplies the learnable prompt (B,D,1,1) with the CLIP prompt viewed as (B,1,B?,512) and
//                   F.interpolate(nearest)s the product to (ps,ps): the prompt's BATCH axis lands on the image rows
//                   (SURVEY Q1).  Written out, that is the formula above -- one fp32 multiply per output element.
//   resize_bilinear F.interpolate(mode="bilinear", align_corners=False) (:580) from the prompt size to the feature size;
//                   only taken when the input is not 64x64 (the 512x512 test cubes: an 8x upsample of a (B,ps,ps,D) map).
//                   src = max((dst + 0.5) * in/out - 0.5, 0), i0 = floor(src), i1 = min(i0 + 1, in - 1) -- PyTorch's
//                   area_pixel_compute_source_index.  The backward is a gather (each input pixel sums the output pixels
//                   whose two taps touch it, with the same weights): no atomics, bitwise reproducible.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

__global__ __launch_bounds__(256) void tvsp_text_map_kernel(const float* __restrict__ L, const float* __restrict__ clip,
                                                            float* __restrict__ text, int B, int ps, int D) {
    const long n = (long)B * ps * ps * D;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int d = (int)(i % D);
        const long t = i / D;
        const int j = (int)(t % ps), ii = (int)((t / ps) % ps), b = (int)(t / ((long)ps * ps));
        text[i] = L[b * D + d] * clip[(long)((ii * B) / ps) * 512 + (j * 512) / ps];
    }
}

// dL partials: part[b][i][d] = sum_j dtext[b,i,j,d] * clip_map[i][j]   (the caller sums over i in a fixed order)
__global__ __launch_bounds__(256) void tvsp_text_map_bwd_kernel(const float* __restrict__ dtext, const float* __restrict__ clip,
                                                                float* __restrict__ part, int B, int ps, int D) {
    const int b = blockIdx.x / ps, i = blockIdx.x % ps;
    const float* crow = clip + (long)((i * B) / ps) * 512;
    const float* src = dtext + ((long)b * ps + i) * ps * D;
    for (int d = threadIdx.x; d < D; d += 256) {
        float acc = 0.f;
        for (int j = 0; j < ps; ++j) acc += src[(long)j * D + d] * crow[(j * 512) / ps];
        part[((long)b * ps + i) * D + d] = acc;
    }
}

struct ResizeDev {
    const void* X; void* Y;
    int B, h, w, H, W, C;       // X (B,h,w,C) -> Y (B,H,W,C)
};

__device__ __forceinline__ void bilinear_tap(int dst, int in, int out, int& i0, int& i1, float& l1) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
}

template <class T>
__global__ __launch_bounds__(256) void resize_bilinear_kernel(ResizeDev a) {
    constexpr int VEC = Vec16<T>::N;
    const int cv = a.C / VEC;
    const long n = (long)a.B * a.H * a.W * cv;
    const T* X = reinterpret_cast<const T*>(a.X);
    T* Y = reinterpret_cast<T*>(a.Y);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * VEC;
        const long p = i / cv;
        const int x = (int)(p % a.W), y = (int)((p / a.W) % a.H), b = (int)(p / ((long)a.W * a.H));
        int y0, y1, x0, x1;
        float ly, lx;
        bilinear_tap(y, a.h, a.H, y0, y1, ly);
        bilinear_tap(x, a.w, a.W, x0, x1, lx);
        const T* base = X + (long)b * a.h * a.w * a.C + c0;
        const Vec16<T> v00 = load16<T>(base + ((long)y0 * a.w + x0) * a.C), v01 = load16<T>(base + ((long)y0 * a.w + x1) * a.C);
        const Vec16<T> v10 = load16<T>(base + ((long)y1 * a.w + x0) * a.C), v11 = load16<T>(base + ((long)y1 * a.w + x1) * a.C);
        Vec16<T> o;
        for (int e = 0; e < VEC; ++e)
            o.set(e, (1.f - ly) * ((1.f - lx) * v00.get(e) + lx * v01.get(e)) + ly * ((1.f - lx) * v10.get(e) + lx * v11.get(e)));
        store16<T>(Y + p * a.C + c0, o);
    }
}

// dX[b,iy,ix,:] = sum over the output pixels (y,x) whose taps include (iy,ix) of weight * dY[b,y,x,:]
template <class T>
__global__ __launch_bounds__(256) void resize_bilinear_bwd_kernel(ResizeDev a) {      // a.X = dY (B,H,W,C), a.Y = dX (B,h,w,C)
    constexpr int VEC = Vec16<T>::N;
    const int cv = a.C / VEC;
    const long n = (long)a.B * a.h * a.w * cv;
    const T* dY = reinterpret_cast<const T*>(a.X);
    T* dX = reinterpret_cast<T*>(a.Y);
    const float sy = (float)a.H / (float)a.h, sx = (float)a.W / (float)a.w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cv) * VEC;
        const long p = i / cv;
        const int ix = (int)(p % a.w), iy = (int)((p / a.w) % a.h), b = (int)(p / ((long)a.w * a.h));
        float acc[VEC];
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
        // output pixel y reads input rows floor(s) and floor(s) + 1 with s = (y + 0.5) h / H - 0.5: it touches iy for s in
        // (iy - 1, iy + 1), i.e. y in ((iy - 0.5) H / h - 0.5, (iy + 1.5) H / h - 0.5) -- any ratio (one pixel of slack each
        // side for the rounding of the bounds; the weights decide)
        int ylo = (int)floorf(((float)iy - 0.5f) * sy - 0.5f) - 1, xlo = (int)floorf(((float)ix - 0.5f) * sx - 0.5f) - 1;
        int yhi = (int)ceilf(((float)iy + 1.5f) * sy - 0.5f) + 1, xhi = (int)ceilf(((float)ix + 1.5f) * sx - 0.5f) + 1;
        ylo = ylo < 0 ? 0 : ylo;
        xlo = xlo < 0 ? 0 : xlo;
        yhi = yhi > a.H - 1 ? a.H - 1 : yhi;
        xhi = xhi > a.W - 1 ? a.W - 1 : xhi;
        for (int y = ylo; y <= yhi; ++y) {
            int y0, y1;
            float ly;
            bilinear_tap(y, a.h, a.H, y0, y1, ly);
            const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int x = xlo; x <= xhi; ++x) {
                int x0, x1;
                float lx;
                bilinear_tap(x, a.w, a.W, x0, x1, lx);
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                if (wx == 0.f) continue;
                const Vec16<T> g = load16<T>(dY + (((long)b * a.H + y) * a.W + x) * a.C + c0);
                for (int e = 0; e < VEC; ++e) acc[e] += wy * wx * g.get(e);
            }
        }
        Vec16<T> o;
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        store16<T>(dX + p * a.C + c0, o);
    }
}

static unsigned grid_for(long n) {
    long blocks = (n + 255) / 256;
    return (unsigned)(blocks > 256 * 16 ? 256 * 16 : blocks);
}

}  // namespace mphsir

extern "C" int mphsir_tvsp_text_map(const float* L, const float* clip, float* text, int32_t B, int32_t ps, int32_t D, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(L && clip && text && B > 0 && ps > 0 && D > 0, "tvsp_text_map: bad arguments");
    MPHSIR_LAUNCH(MPHSIR_K_RESAMPLE, tvsp_text_map_kernel, dim3(grid_for((long)B * ps * ps * D)), dim3(256), 0,
                  reinterpret_cast<hipStream_t>(stream), L, clip, text, (int)B, (int)ps, (int)D);
    return MPHSIR_OK;
}

extern "C" int mphsir_tvsp_text_map_bwd(const float* dtext, const float* clip, float* part, int32_t B, int32_t ps, int32_t D, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(dtext && clip && part && B > 0 && ps > 0 && D > 0, "tvsp_text_map_bwd: bad arguments");
    MPHSIR_LAUNCH(MPHSIR_K_RESAMPLE, tvsp_text_map_bwd_kernel, dim3((unsigned)(B * ps)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                  dtext, clip, part, (int)B, (int)ps, (int)D);
    return MPHSIR_OK;
}

extern "C" int mphsir_resize_bilinear(const void* X, void* Y, int32_t B, int32_t h, int32_t w, int32_t H, int32_t W, int32_t C, int32_t backward,
                                      int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && Y && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, "resize_bilinear: bad arguments");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "resize_bilinear: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(C % (16 / dtype_size(dtype)) == 0 && aligned16(X) && aligned16(Y), "resize_bilinear: C must fill 16-byte vectors, 16-byte alignment");
    ResizeDev d{X, Y, B, h, w, H, W, C};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long n = (long)B * (backward ? (long)h * w : (long)H * W) * (C / (16 / dtype_size(dtype)));
    const dim3 grid(grid_for(n));
    if (backward)
        return MPHSIR_DISPATCH_T(dtype, ([&]() -> int { MPHSIR_LAUNCH(MPHSIR_K_RESAMPLE, (resize_bilinear_bwd_kernel<T_>), grid, dim3(256), 0, s, d); return MPHSIR_OK; }()));
    return MPHSIR_DISPATCH_T(dtype, ([&]() -> int { MPHSIR_LAUNCH(MPHSIR_K_RESAMPLE, (resize_bilinear_kernel<T_>), grid, dim3(256), 0, s, d); return MPHSIR_OK; }()));
}
