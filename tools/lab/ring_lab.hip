// Lab: what read rate does ONE workgroup per CU reach when it streams a contiguous token range by LDS-DMA into a ring
// (no arithmetic), as a function of waves per workgroup, bytes in flight and row segment length?  And plain register loads.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/ring_lab tools/lab/ring_lab.hip && /tmp/ring_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt(0x0f70 | (N & 15) | ((N >> 4) << 14)); }
__device__ __forceinline__ void lds_barrier() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
#define DMA16P(gptr, lds)                                                                                              \
    do {                                                                                                                \
        unsigned keep_;                                                                                                 \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                     : "=&s"(keep_) : "v"(gptr), "s"((unsigned)(unsigned long long)(lds)) : "memory");                  \
    } while (0)

// Each workgroup streams rows [wg*rows_per, +rows_per) of a [M][ld] bf16 matrix, `seg` bytes per row (seg = 256: 4 rows per DMA
// instruction).  WAVES waves; every wave issues IPW instructions (1 KB each) per step; DEPTH steps in flight; barrier per step if BAR.
template <int WAVES, int IPW, int DEPTH, bool BAR>
__global__ __launch_bounds__(WAVES * 64) void dma_stream(const char* X, long ld_bytes, int seg, long rows_per, float* out, int share = 0, int reps = 1) {
    extern __shared__ char smem[];
    constexpr int STEP_BYTES = WAVES * IPW * 1024, RING = DEPTH + 1;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int rows_per_instr = 1024 / seg, rows_per_step = WAVES * IPW * rows_per_instr;
    // share = k > 0: workgroups b, b+8, b+16, ... (k of them, one XCD) stream the SAME rows; the region is swept reps times
    const long r0 = share ? (long)((blockIdx.x & 7) + 8 * (blockIdx.x / (8 * share))) * rows_per : (long)blockIdx.x * rows_per;
    const int steps1 = (int)(rows_per / rows_per_step), nsteps = steps1 * reps;
    const char* base = X + (r0 + (long)(wv * IPW) * rows_per_instr + lane / (seg / 16)) * ld_bytes + (lane % (seg / 16)) * 16;
    auto issue = [&](int s) {
        char* slot = smem + (s % RING) * STEP_BYTES + wv * IPW * 1024;
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const char* g = base + ((long)(s % steps1) * rows_per_step + (long)i * rows_per_instr) * ld_bytes;
            DMA16P(g, slot + i * 1024);
        }
    };
    for (int s = 0; s < DEPTH && s < nsteps; ++s) issue(s);
    float acc = 0.f;
    for (int s = 0; s < nsteps; ++s) {
        if (s + DEPTH <= nsteps) wait_vmcnt<(DEPTH - 1) * IPW>(); else wait_vmcnt<0>();
        if (BAR) lds_barrier();
        acc += *reinterpret_cast<const float*>(smem + (s % RING) * STEP_BYTES + threadIdx.x * 4);   // touch the slot
        if (BAR) lds_barrier();
        if (s + DEPTH < nsteps) issue(s + DEPTH);
    }
    if (acc == 123.456f) out[0] = acc;
}

// register loads: every thread keeps U 16-byte loads in flight
template <int THREADS, int U>
__global__ __launch_bounds__(THREADS) void reg_stream(const char* X, long bytes_per_wg, float* out) {
    const char* p = X + (long)blockIdx.x * bytes_per_wg + threadIdx.x * 16;
    const long n = bytes_per_wg / (THREADS * 16 * U);
    float acc = 0.f;
    for (long i = 0; i < n; ++i) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(p + ((long)i * U + u) * THREADS * 16);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const long BYTES = 1L << 30;            // 1 GiB: beyond the Infinity Cache
    char* X; float* out;
    CK(hipMalloc(&X, BYTES)); CK(hipMalloc(&out, 4));
    CK(hipMemset(X, 1, BYTES));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double bytes) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e9;
        for (int r = 0; r < 5; ++r) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        printf("%-60s %8.1f us  %6.2f TB/s\n", name, best * 1e3, bytes / best / 1e9);
    };
    const int WGS = 256;
#define DMA_CASE(WAVES, IPW, DEPTH, BAR, seg, ld)                                                              \
    {                                                                                                          \
        const long rows = BYTES / (ld), rows_per = rows / WGS / (WAVES * IPW * (1024 / (seg))) * (WAVES * IPW * (1024 / (seg))); \
        const size_t sh = (size_t)(DEPTH + 1) * WAVES * IPW * 1024;                                            \
        CK(hipFuncSetAttribute((const void*)dma_stream<WAVES, IPW, DEPTH, BAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh)); \
        char nm[128];                                                                                          \
        snprintf(nm, 128, "dma %d waves x %d KB/step, %d steps ahead (%d KB), bar=%d, seg %d of %d", WAVES, WAVES * IPW, DEPTH, DEPTH * WAVES * IPW, (int)BAR, seg, (int)(ld)); \
        timeit(nm, [&] { hipLaunchKernelGGL((dma_stream<WAVES, IPW, DEPTH, BAR>), dim3(WGS), dim3(WAVES * 64), sh, 0, X, (long)(ld), seg, rows_per, out); }, \
               (double)rows_per * WGS * (seg));                                                                \
    }
    DMA_CASE(8, 2, 3, true, 256, 256)
    DMA_CASE(8, 2, 6, true, 256, 256)
    DMA_CASE(8, 2, 8, true, 256, 256)
    DMA_CASE(8, 2, 6, false, 256, 256)
    DMA_CASE(8, 1, 6, true, 256, 256)
    DMA_CASE(8, 1, 12, true, 256, 256)
    DMA_CASE(8, 1, 18, true, 256, 256)
    DMA_CASE(4, 2, 8, true, 256, 256)
    DMA_CASE(4, 2, 16, true, 256, 256)
    DMA_CASE(4, 4, 8, true, 256, 256)
    DMA_CASE(16, 1, 8, true, 256, 256)
    DMA_CASE(8, 2, 6, true, 256, 768)
    DMA_CASE(8, 2, 6, true, 256, 1408)
    DMA_CASE(8, 2, 6, true, 128, 128)
    DMA_CASE(8, 2, 6, true, 128, 384)
    DMA_CASE(8, 2, 6, true, 1024, 1024)
    // sharing: k workgroups of one XCD stream the same rows (the B operand of k output tiles)
    for (int k : {1, 2, 3, 4, 8}) {
        const long ld = 256, rows = BYTES / ld, rows_per = rows / WGS / 64 * 64;
        const size_t sh = (size_t)7 * 16 * 1024;
        CK(hipFuncSetAttribute((const void*)dma_stream<8, 2, 6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        char nm[128];
        snprintf(nm, 128, "dma 8 waves, 96 KB ahead: %d workgroups of an XCD share a stream (CU-side bytes)", k);
        timeit(nm, [&] { hipLaunchKernelGGL((dma_stream<8, 2, 6, true>), dim3(WGS), dim3(512), sh, 0, X, ld, 256, rows_per, out, k, 1); }, (double)rows_per * WGS * 256);
    }
    // everything from L2: every workgroup sweeps its own 64 KB region 64 times / one shared 2 MB region
    {
        const size_t sh = (size_t)7 * 16 * 1024;
        timeit("dma 8 waves, 96 KB ahead: own 64 KB region x 256 sweeps (L2-resident)", [&] { hipLaunchKernelGGL((dma_stream<8, 2, 6, true>), dim3(WGS), dim3(512), sh, 0, X, 256L, 256, 256L, out, 0, 256); }, 256.0 * WGS * 256 * 256);
        timeit("dma 8 waves, 96 KB ahead: own 1 MB region x 16 sweeps", [&] { hipLaunchKernelGGL((dma_stream<8, 2, 6, true>), dim3(WGS), dim3(512), sh, 0, X, 256L, 256, 4096L, out, 0, 16); }, 4096.0 * WGS * 256 * 16);
    }
    for (int wgs : {256, 2048}) {
        char nm[128];
        snprintf(nm, 128, "reg 256 threads x 8 loads in flight, %d workgroups", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL((reg_stream<256, 8>), dim3(wgs), dim3(256), 0, 0, X, BYTES / wgs, out); }, (double)BYTES);
        snprintf(nm, 128, "reg 512 threads x 4 loads in flight, %d workgroups", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL((reg_stream<512, 4>), dim3(wgs), dim3(512), 0, 0, X, BYTES / wgs, out); }, (double)BYTES);
        snprintf(nm, 128, "reg 1024 threads x 4 loads in flight, %d workgroups", wgs);
        timeit(nm, [&] { hipLaunchKernelGGL((reg_stream<1024, 4>), dim3(wgs), dim3(1024), 0, 0, X, BYTES / wgs, out); }, (double)BYTES);
    }
    return 0;
}
