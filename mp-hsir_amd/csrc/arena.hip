// arena utilities: the two places where hundreds of tiny framework kernels per step are replaced by one launch each.
//
//  * reduce_parts: every split-M / per-workgroup partial buffer a backward function produced (gemm_tn tiles and
//    column sums, LayerNorm / rel-pos-bias / depthwise-tap / fold partials) is summed in a fixed order by ONE launch
//    over up to MPHSIR_REDUCE_MAX_SEGS segments (the reference: one at::sum per parameter gradient).  Fixed summation
//    order, no atomics: bitwise reproducible.
//  * pack_gather: all kernel-layout weights (cast to the compute dtype, padded, transposed, split, gathered) are one
//    gather from the flat fp32 parameter arena through a static int32 index map built once from the packers
//    (host: engine.PackPlan); index < 0 = zero padding.  Runs right after flat_adamw inside the captured step.
#include <stdlib.h>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct RedSegDev {
    const float* src; float* dst;
    long n, stride, sbs, dbs, t0, nitems; // t0: first thread of this segment (a multiple of 64: waves never straddle segments)
    long src_ld, dst_ld;                  // 2-D form: row pitches (rows x n block of a wider matrix)
    int nsplit, vec, lg, rows, dcs;       // 2^lg adjacent lanes share one work item (4 consecutive outputs) and split the split axis
};
struct RedDev {
    RedSegDev s[MPHSIR_REDUCE_MAX_SEGS];
    int nseg; long threads;
};

__global__ __launch_bounds__(256) void reduce_parts_kernel(RedDev a) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= a.threads) return;            // whole waves only (threads is a multiple of 64)
    // constant indices only: a dynamically indexed by-value kernel argument would be copied to scratch by every lane
    RedSegDev s = a.s[0];
#pragma unroll
    for (int k = 1; k < MPHSIR_REDUCE_MAX_SEGS; ++k)
        if (k < a.nseg && t >= a.s[k].t0) s = a.s[k];
    const int G = 1 << s.lg;
    const long local = t - s.t0, item = local >> s.lg, ipr = (s.n + 3) / 4, ipb = ipr * s.rows;
    const int g = (int)(local & (G - 1));
    const bool live = item < s.nitems;
    const long b = live ? item / ipb : 0, rem = live ? item % ipb : 0, r = rem / ipr, i4 = (rem % ipr) * 4;
    const float* src = s.src + b * s.sbs + r * s.src_ld + i4;
    const int ne = (s.n - i4) < 4 ? (int)(s.n - i4) : 4;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    if (live) {
        // 8 independent loads (16-byte ones where the source allows: bit 0 of `vec`) in flight per lane -- the kernel is a latency
        // chain of strided L2 reads -- summed in split order.  (Until round 5 only the all-vector segments did this; the others -- tap
        // gradients written transposed, the 450-wide relative-position-bias rows with 2048 splits -- walked their splits one by one
        // and set the time of the launch: 60 us for 69 MB.)
        int sp = g;
        if (s.vec & 1) {
            for (; sp + 7 * G < s.nsplit; sp += 8 * G) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (long)(sp + u * G) * s.stride);
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; sp < s.nsplit; sp += G) acc += *reinterpret_cast<const f32x4*>(src + (long)sp * s.stride);
        } else {
            for (; sp + 7 * G < s.nsplit; sp += 8 * G) {
                f32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    for (int e = 0; e < 4; ++e) v[u][e] = e < ne ? src[(long)(sp + u * G) * s.stride + e] : 0.f;
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
            for (; sp < s.nsplit; sp += G)
                for (int e = 0; e < ne; ++e) acc[e] += src[(long)sp * s.stride + e];
        }
    }
    // fixed xor tree over the lanes of the group: the summation order depends on (nsplit, G) only -> deterministic
    for (int m = 1; m < G; m <<= 1)
        for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], m);
    if (live && g == 0) {
        float* dst = s.dst + b * s.dbs + r * s.dst_ld + i4 * s.dcs;
        if (s.vec & 2) *reinterpret_cast<f32x4*>(dst) = acc;
        else
            for (int e = 0; e < ne; ++e) dst[(long)e * s.dcs] = acc[e];
    }
}

// 4 elements per thread (one 16-byte store for fp32, 8 bytes for the 16-bit types) and one 256-thread workgroup per 1024 elements.  What was
// measured on the 12.8 M-element bf16 pack of the natural-scene net (rocprofv3, round 6): 8 elements per thread grid-strided over 4096
// workgroups 113 us; one piece per workgroup 99; 4 / 8 / 16 CONSECUTIVE pieces per workgroup (every source cache line used whole: the
// transposed copies otherwise re-fetch them, 311 MB read for 25.6 MB written) 139 / 143 / 131 -- the gather is bound by the number of
// distinct lines a wave asks for per instruction, not by bytes, and wants threads, not locality; 4 elements per thread 92.
template <class T>
__global__ __launch_bounds__(256) void pack_gather_kernel(const float* arena, const int* idx, T* dst, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n / 4) return;
    const int* ip = idx + i * 4;
    const int i0 = ip[0], i1 = ip[1], i2 = ip[2], i3 = ip[3];
    f32x4 v;
    v[0] = i0 >= 0 ? arena[i0] : 0.f;
    v[1] = i1 >= 0 ? arena[i1] : 0.f;
    v[2] = i2 >= 0 ? arena[i2] : 0.f;
    v[3] = i3 >= 0 ? arena[i3] : 0.f;
    store4<T>(dst + i * 4, v);
}

// multi_copy: the gradient hand-over.  autograd leaves ~370 freshly allocated fp32 gradient tensors; each goes to its slot of the flat
// gradient arena.  (torch._foreach_copy_ took 11 launches of its multi-tensor kernel, 0.2 ms per step.)  One launch: the table --
// rows {src, dst, n floats, first block} in DEVICE memory -- is searched by block index; a block moves up to 4096 floats.
constexpr int MC_BLOCK = 4096;
__global__ __launch_bounds__(256) void multi_copy_kernel(const long* __restrict__ table, int nseg) {
    const long b = blockIdx.x;
    int lo = 0, hi = nseg - 1;                               // the last row whose first block is <= b (wave-uniform: scalar loads)
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[4 * mid + 3] <= b) lo = mid; else hi = mid - 1;
    }
    const float* src = reinterpret_cast<const float*>(table[4 * lo]);
    float* dst = reinterpret_cast<float*>(table[4 * lo + 1]);
    const long n = table[4 * lo + 2], e0 = (b - table[4 * lo + 3]) * MC_BLOCK;
    const long e1 = e0 + MC_BLOCK < n ? e0 + MC_BLOCK : n;
    if ((((unsigned long)src | (unsigned long)dst) & 15) == 0) {
        for (long e = e0 + 4 * threadIdx.x; e + 4 <= e1; e += 1024) *reinterpret_cast<f32x4*>(dst + e) = *reinterpret_cast<const f32x4*>(src + e);
        for (long e = e0 + ((e1 - e0) & ~3L) + threadIdx.x; e < e1; e += 256) dst[e] = src[e];
    } else {
        for (long e = e0 + threadIdx.x; e < e1; e += 256) dst[e] = src[e];
    }
}

// l1_clamp_loss: the training loss of the reference (train.py:58-61: clamp(restored, 0, 1), nn.L1Loss) and its gradient in one
// pass: part[block] = sum |clamp(y) - c| / n over the block's elements, g = sign(clamp(y) - c) * [0 <= y <= 1] / n (clamp's
// autograd passes the gradient where 0 <= y <= 1, bounds included; sign(0) = 0).  The caller sums the partials in order.
// The clamp PROPAGATES NaN like torch.clamp (two compares, both false for a NaN; fminf / fmaxf would return the bound and a
// diverged run would log a plausible finite loss): a NaN output makes the loss NaN, its gradient entry 0 as in the reference.
__device__ __forceinline__ float clamp01_nan(float y) { return y < 0.f ? 0.f : (y > 1.f ? 1.f : y); }
__global__ __launch_bounds__(256) void l1_clamp_loss_kernel(const float* __restrict__ y, const float* __restrict__ c, float* __restrict__ g,
                                                            float* __restrict__ part, long n, float inv_n) {
    __shared__ float red[4];
    float acc = 0.f;
    const long stride = (long)gridDim.x * 1024;
    for (long e = (long)blockIdx.x * 1024 + 4 * threadIdx.x; e < n; e += stride) {
        if (e + 4 <= n) {
            const f32x4 yv = *reinterpret_cast<const f32x4*>(y + e), cv = *reinterpret_cast<const f32x4*>(c + e);
            f32x4 gv;
            for (int i = 0; i < 4; ++i) {
                const float d = clamp01_nan(yv[i]) - cv[i];
                acc += fabsf(d);
                gv[i] = (yv[i] >= 0.f && yv[i] <= 1.f) ? (d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f)) : 0.f;
            }
            if (g) *reinterpret_cast<f32x4*>(g + e) = gv;
        } else {
            for (long k = e; k < n; ++k) {
                const float d = clamp01_nan(y[k]) - c[k];
                acc += fabsf(d);
                if (g) g[k] = (y[k] >= 0.f && y[k] <= 1.f) ? (d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f)) : 0.f;
            }
        }
    }
    acc = wave_sum(acc);
    if (lane_id() == 0) red[wave_id()] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_n;
}

}  // namespace mphsir

extern "C" int mphsir_multi_copy(const int64_t* table_dev, int32_t nseg, int64_t total_blocks, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(table_dev && nseg > 0 && total_blocks > 0 && total_blocks < (1LL << 31), "multi_copy: bad arguments");
    static_assert(sizeof(long) == sizeof(int64_t), "table rows are 64-bit");
    MPHSIR_LAUNCH(MPHSIR_K_MULTI_COPY, multi_copy_kernel, dim3((unsigned)total_blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                  reinterpret_cast<const long*>(table_dev), (int)nseg);
    return MPHSIR_OK;
}

extern "C" int mphsir_l1_clamp_loss(const float* y, const float* clean, float* grad, float* part, int64_t n, int32_t nblocks, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(y && clean && part && n > 0 && nblocks > 0 && nblocks <= 65535, "l1_clamp_loss: bad arguments");
    MPHSIR_REQUIRE(aligned16(y) && aligned16(clean) && (grad == nullptr || aligned16(grad)), "l1_clamp_loss: 16-byte alignment required");
    MPHSIR_LAUNCH(MPHSIR_K_L1_LOSS, l1_clamp_loss_kernel, dim3((unsigned)nblocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                  y, clean, grad, part, (long)n, 1.0f / (float)n);
    return MPHSIR_OK;
}

extern "C" int mphsir_reduce_parts(const mphsir_reduce_seg* segs, int32_t nseg, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(segs && nseg > 0 && nseg <= MPHSIR_REDUCE_MAX_SEGS, "reduce_parts: 1..%d segments per call", MPHSIR_REDUCE_MAX_SEGS);
    RedDev d;
    d.nseg = nseg;
    long threads = 0;
    for (int k = 0; k < nseg; ++k) {
        const mphsir_reduce_seg& g = segs[k];
        MPHSIR_REQUIRE(g.src && g.dst && g.n > 0 && g.nsplit > 0 && g.nbatch > 0, "reduce_parts: bad segment %d", k);
        const int rows = g.rows > 1 ? g.rows : 1, dcs = g.dst_col_stride > 0 ? g.dst_col_stride : 1;
        const long src_ld = rows > 1 ? (long)g.src_ld : 0, dst_ld = rows > 1 ? (long)g.dst_ld : 0;
        const bool vload = aligned16(g.src) && g.n % 4 == 0 && g.stride % 4 == 0 && g.src_batch_stride % 4 == 0 && src_ld % 4 == 0;
        const bool vstore = vload && aligned16(g.dst) && dcs == 1 && g.dst_batch_stride % 4 == 0 && dst_ld % 4 == 0;
        const long nitems = (long)g.nbatch * rows * ((g.n + 3) / 4);
        int lg = 0;                        // lanes per item: keep <= 32 splits per lane, and small segments still fill waves
        while (lg < 6 && (g.nsplit >> lg) > 32) ++lg;      // (<= 8 / <= 4 per lane measured in round 5: 21.10 / 21.29 against 21.09 ms per step)
        while (lg < 6 && (nitems << lg) < 8192 && (g.nsplit >> lg) > 4) ++lg;      // (32 k / 64 k / 128 k threads measured: no difference)
        d.s[k] = RedSegDev{g.src, g.dst, (long)g.n, (long)g.stride, (long)g.src_batch_stride, (long)g.dst_batch_stride, threads, nitems,
                           src_ld, dst_ld, g.nsplit, (vload ? 1 : 0) | (vstore ? 2 : 0), lg, rows, dcs};
        threads += ((nitems << lg) + 63) / 64 * 64;
    }
    d.threads = threads;
    MPHSIR_LAUNCH(MPHSIR_K_REDUCE_PARTS, reduce_parts_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                  reinterpret_cast<hipStream_t>(stream), d);
    return MPHSIR_OK;
}

extern "C" int mphsir_pack_gather(const float* arena, const int32_t* index, void* dst, int64_t n, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(arena && index && dst, "pack_gather: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "pack_gather: dtype %d unsupported", dtype);
    const int vec = dtype == MPHSIR_F32 ? 4 : 8;
    MPHSIR_REQUIRE(n > 0 && n % vec == 0 && aligned16(index) && aligned16(dst), "pack_gather: n must be a multiple of %d, 16-byte alignment", vec);
    const dim3 grid((unsigned)((n / 4 + 255) / 256));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MPHSIR_F32) {
        MPHSIR_LAUNCH(MPHSIR_K_PACK_GATHER, pack_gather_kernel<float>, grid, dim3(256), 0, s, arena, index, reinterpret_cast<float*>(dst), (long)n);
    } else if (dtype == MPHSIR_BF16) {
        MPHSIR_LAUNCH(MPHSIR_K_PACK_GATHER, pack_gather_kernel<bf16_t>, grid, dim3(256), 0, s, arena, index, reinterpret_cast<bf16_t*>(dst), (long)n);
    } else {
        MPHSIR_LAUNCH(MPHSIR_K_PACK_GATHER, pack_gather_kernel<f16_t>, grid, dim3(256), 0, s, arena, index, reinterpret_cast<f16_t*>(dst), (long)n);
    }
    return MPHSIR_OK;
}
