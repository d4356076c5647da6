// ================================================================================================
// hipemu -- a CPU stand-in for <hip/hip_runtime.h>.            *** TEST INFRASTRUCTURE ONLY ***
//
// Lets the *unmodified* kernel sources under mp-hsir_amd/csrc/ (and the C-ABI host code that
// launches them) be compiled for x86 with clang++ and run under pytest on a machine without a GPU,
// so index arithmetic, LDS layouts, MFMA fragment maps and wave-shuffle reductions are checked on
// every CPU test run.  It is not a product path: the product library (libmphsir.so) is built by hipcc
// for gfx950 only, and the Python package refuses to run without it.
//
// Model: one workgroup at a time; every work-item is a fiber (hand-rolled x86-64 context switch);
// __syncthreads() and the wave-collective operations (shuffles, MFMA) are rendezvous points.
// A wave is 64 consecutive work-items.  Wave collectives must be reached by all 64 lanes (as on the
// hardware: MFMA needs EXEC all ones).  Global memory is host memory.
// ================================================================================================
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <sys/mman.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(hipemu::dyn_lds());

struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipPeekAtLastError() { return hipSuccess; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
template <class K> inline hipError_t hipFuncSetAttribute(K, hipFuncAttribute, int) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
struct hipDeviceProp_t { char gcnArchName[64]; int multiProcessorCount; };
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    strcpy(p->gcnArchName, "gfx950:hipemu"); p->multiProcessorCount = 256; return hipSuccess;
}

namespace hipemu {

constexpr int kWave = 64;
constexpr size_t kStack = 512 * 1024;
constexpr int kMaxThreads = 1024;

struct Fiber {
    void* sp;
    dim3 tid;
    int flat;
    bool done;
};

struct WaveState {
    int count = 0, gen = 0;
    alignas(16) unsigned char xa[kWave][64];
    alignas(16) unsigned char xb[kWave][64];
};

struct State {
    dim3 bid, bdim, gdim;
    int nthreads = 0;
    int bar_count = 0, bar_gen = 0;
    Fiber fibers[kMaxThreads];
    WaveState waves[kMaxThreads / kWave];
    Fiber* cur = nullptr;
    void* sched_sp = nullptr;
    char* stacks = nullptr;
    std::vector<unsigned char> lds;
    std::function<void()> body;
};

inline State& S() { static State s; return s; }

extern "C" void hipemu_switch(void** save_sp, void* load_sp);
#ifdef HIPEMU_IMPLEMENTATION
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");
#endif

inline void yield() { State& s = S(); hipemu_switch(&s.cur->sp, s.sched_sp); }

inline void fiber_entry() {
    State& s = S();
    s.body();
    s.cur->done = true;
    hipemu_switch(&s.cur->sp, s.sched_sp);
    abort();
}

inline void* dyn_lds() { return S().lds.data(); }

inline void block_sync() {
    State& s = S();
    int gen = s.bar_gen;
    if (++s.bar_count == s.nthreads) { s.bar_count = 0; s.bar_gen++; }
    else while (s.bar_gen == gen) yield();
}

inline WaveState& my_wave() { State& s = S(); return s.waves[s.cur->flat / kWave]; }
inline int lane_id() { return S().cur->flat % kWave; }

inline void wave_sync() {
    State& s = S();
    WaveState& w = my_wave();
    int first = (s.cur->flat / kWave) * kWave;
    int n = s.nthreads - first < kWave ? s.nthreads - first : kWave;
    int gen = w.gen;
    if (++w.count == n) { w.count = 0; w.gen++; }
    else while (w.gen == gen) yield();
}

template <class T> inline T wave_xchg(T v, int src_lane) {
    static_assert(sizeof(T) <= 64, "");
    WaveState& w = my_wave();
    memcpy(w.xa[lane_id()], &v, sizeof(T));
    wave_sync();
    T r;
    memcpy(&r, w.xa[src_lane & (kWave - 1)], sizeof(T));
    wave_sync();
    return r;
}

inline void run_block(dim3 bid, dim3 gdim, dim3 bdim, size_t shmem) {
    State& s = S();
    if (!s.stacks) {
        s.stacks = (char*)mmap(nullptr, kStack * kMaxThreads, PROT_READ | PROT_WRITE,
                               MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (s.stacks == MAP_FAILED) { perror("hipemu mmap"); abort(); }
    }
    s.bid = bid; s.gdim = gdim; s.bdim = bdim;
    s.nthreads = (int)(bdim.x * bdim.y * bdim.z);
    if (s.nthreads > kMaxThreads || s.nthreads % kWave) { fprintf(stderr, "hipemu: bad block size %d\n", s.nthreads); abort(); }
    s.bar_count = 0; s.bar_gen = 0;
    for (auto& w : s.waves) { w.count = 0; w.gen = 0; }
    if (s.lds.size() < shmem + 64) s.lds.resize(shmem + 64);
    memset(s.lds.data(), 0xCD, s.lds.size());   // poison: reads of unwritten LDS show up as garbage
    for (int t = 0; t < s.nthreads; ++t) {
        Fiber& f = s.fibers[t];
        f.flat = t; f.done = false;
        f.tid = dim3(t % bdim.x, (t / bdim.x) % bdim.y, t / (bdim.x * bdim.y));
        void** top = (void**)(s.stacks + kStack * (t + 1));
        *--top = nullptr;                       // keeps rsp % 16 == 8 at fiber_entry's first instruction
        *--top = (void*)&fiber_entry;           // 'ret' target
        for (int r = 0; r < 6; ++r) *--top = nullptr;
        f.sp = top;
    }
    int remaining = s.nthreads;
    while (remaining) {
        for (int t = 0; t < s.nthreads; ++t) {
            Fiber& f = s.fibers[t];
            if (f.done) continue;
            s.cur = &f;
            hipemu_switch(&s.sched_sp, f.sp);
            if (f.done) --remaining;
        }
    }
    s.cur = nullptr;
}

template <class K, class... A> inline void launch(K kern, dim3 grid, dim3 block, size_t shmem, A... args) {
    State& s = S();
    s.body = [=]() { kern(args...); };
    for (unsigned z = 0; z < grid.z; ++z)
        for (unsigned y = 0; y < grid.y; ++y)
            for (unsigned x = 0; x < grid.x; ++x) run_block(dim3(x, y, z), grid, block, shmem);
}

}  // namespace hipemu

#define threadIdx (hipemu::S().cur->tid)
#define blockIdx (hipemu::S().bid)
#define blockDim (hipemu::S().bdim)
#define gridDim (hipemu::S().gdim)
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...) \
    hipemu::launch(kern, dim3(grid), dim3(block), (size_t)(shmem), ##__VA_ARGS__)

inline void __syncthreads() { hipemu::block_sync(); }
template <class T> inline T __shfl_xor(T v, int mask, int = 64) { return hipemu::wave_xchg(v, hipemu::lane_id() ^ mask); }
template <class T> inline T __shfl(T v, int src, int = 64) { return hipemu::wave_xchg(v, src); }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) {
    int l = hipemu::lane_id(); return hipemu::wave_xchg(v, l + (int)d < 64 ? l + (int)d : l);
}
inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
inline int atomicExch(int* p, int v) { int o = *p; *p = v; return o; }
inline void __threadfence() {}
#ifndef __HIP_MEMORY_SCOPE_AGENT          // __hip_atomic_load / _store / _fetch_add themselves are clang builtins on the host too
#define __HIP_MEMORY_SCOPE_AGENT 4
#endif
inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
inline float __expf(float x) { return expf(x); }
inline float __frcp_rn(float x) { return 1.0f / x; }
inline float __fdividef(float a, float b) { return a / b; }
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_fmed3f(x, lo, hi) fminf(fmaxf((x), (lo)), (hi))

// ---- matrix-core emulation: fragment maps per cdna_hip_programming.md §3 -----------------------
namespace hipemu {
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;

// D[row][col] += sum_k A[row][k] * B[k][col]; lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15],
// D[(l>>4)*4+r][l&15].
template <class V8> inline f32x4_t mfma_16x16x32(V8 a, V8 b, f32x4_t c) {
    WaveState& w = my_wave();
    int l = lane_id();
    memcpy(w.xa[l], &a, 16);
    memcpy(w.xb[l], &b, 16);
    wave_sync();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 32; ++k) {
            V8 fa, fb;
            memcpy(&fa, w.xa[row + 16 * (k >> 3)], 16);
            memcpy(&fb, w.xb[col + 16 * (k >> 3)], 16);
            acc = fmaf((float)fa[k & 7], (float)fb[k & 7], acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// f32 16x16x4: lane l holds A[l&15][l>>4], B[l>>4][l&15]; exact fmaf chain in k order.
inline f32x4_t mfma_16x16x4_f32(float a, float b, f32x4_t c) {
    WaveState& w = my_wave();
    int l = lane_id();
    memcpy(w.xa[l], &a, 4);
    memcpy(w.xb[l], &b, 4);
    wave_sync();
    int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float fa, fb;
            memcpy(&fa, w.xa[row + 16 * k], 4);
            memcpy(&fb, w.xb[col + 16 * k], 4);
            acc = fmaf(fa, fb, acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
// ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per group of 16 lanes, lane 4q+p supplies the address of row q,
// columns 4p..4p+3 of a 4x16 block of 16-bit elements; lane i receives column i, row q in element q.
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
inline bf16x4_t ds_read_tr16_b64(const void* p) {
    WaveState& w = my_wave();
    int l = lane_id();
    memcpy(w.xa[l], &p, sizeof(p));
    wave_sync();
    int g = l & ~15, i = l & 15;
    bf16x4_t r;
    for (int q = 0; q < 4; ++q) {
        const char* src;
        memcpy(&src, w.xa[g + 4 * q + (i >> 2)], sizeof(src));
        __bf16 v;
        memcpy(&v, src + 2 * (i & 3), 2);
        r[q] = v;
    }
    wave_sync();
    return r;
}
}  // namespace hipemu
// DPP row shifts (v_mov_b32_dpp row_shl:n = 0x100+n: lane i <- lane i+n; row_shr:n = 0x110+n: lane i <- lane i-n, inside the
// 16-lane row; a lane whose source falls outside its row keeps `old` (bound_ctrl = 0) or gets 0) and ds_bpermute_b32.
namespace hipemu {
inline int update_dpp(int old, int src, int ctrl, int, int, bool bound_ctrl) {
    const int l = lane_id(), row = l & ~15, i = l & 15;
    int sl = -1;
    if (ctrl >= 0x101 && ctrl <= 0x10f) sl = i + (ctrl - 0x100) < 16 ? row + i + (ctrl - 0x100) : -1;
    else if (ctrl >= 0x111 && ctrl <= 0x11f) sl = i - (ctrl - 0x110) >= 0 ? row + i - (ctrl - 0x110) : -1;
    else { fprintf(stderr, "hipemu: dpp_ctrl 0x%x not emulated\n", ctrl); abort(); }
    const int v = wave_xchg(src, sl < 0 ? l : sl);
    return sl < 0 ? (bound_ctrl ? 0 : old) : v;
}
inline int ds_bpermute(int addr, int src) { return wave_xchg(src, (addr >> 2) & (kWave - 1)); }
}  // namespace hipemu
// LDS-DMA (global_load_lds): lane l copies `size` bytes from its own source address to the wave's LDS base + l * size
// (synchronous here); the raw barrier, the counted waits and address-space qualifiers mean nothing on the CPU.
namespace hipemu {
inline void global_load_lds(const void* src, void* dst, int size) { memcpy((char*)dst + (size_t)lane_id() * size, src, size); }
}  // namespace hipemu
#define address_space(n)
#define __builtin_amdgcn_global_load_lds(src, dst, size, off, aux) hipemu::global_load_lds((const void*)(src), (void*)(dst), size)
#define MPHSIR_LDS_DMA16(gbase, byte_off, lds_wave_base) hipemu::global_load_lds((const char*)(gbase) + (byte_off), (void*)(lds_wave_base), 16)
#define MPHSIR_LDS_DMA16P(gptr, lds_wave_base) hipemu::global_load_lds((const void*)(gptr), (void*)(lds_wave_base), 16)
#define __builtin_amdgcn_s_barrier() hipemu::block_sync()
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu::update_dpp(old, src, ctrl, rm, bm, bc)
#define __builtin_amdgcn_ds_bpermute(addr, src) hipemu::ds_bpermute(addr, src)
#define __builtin_amdgcn_s_memtime() 0ull
#define __builtin_amdgcn_readfirstlane(x) (x)      /* wave-uniform by contract: any lane's value */
#define __builtin_amdgcn_wave_barrier() hipemu::wave_sync()
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p) hipemu::ds_read_tr16_b64((const void*)(p))
namespace hipemu {
typedef __attribute__((__vector_size__(4 * sizeof(__fp16)))) __fp16 fp16x4_b;
inline fp16x4_b ds_read_tr16_b64_f16(const void* p) {         // same lane exchange, 2-byte elements moved bit-wise
    const bf16x4_t r = ds_read_tr16_b64(p);
    fp16x4_b o;
    memcpy(&o, &r, sizeof(o));
    return o;
}
}  // namespace hipemu
#define __builtin_amdgcn_ds_read_tr16_b64_v4f16(p) hipemu::ds_read_tr16_b64_f16((const void*)(p))
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu::mfma_16x16x32<hipemu::bf16x8_t>(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hipemu::mfma_16x16x32<hipemu::f16x8_t>(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, x, y, z) hipemu::mfma_16x16x4_f32(a, b, c)
