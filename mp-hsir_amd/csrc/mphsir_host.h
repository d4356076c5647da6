// Host-side plumbing shared by the C-ABI entry points: argument checks, error text, the optional
// per-kernel launch timer, and the launch macro.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mphsir.h"
#include "mphsir_dev.h"

namespace mphsir {

void set_error(const char* fmt, ...);
void clear_error();

// launch timer (mphsir_prof_*): brackets launches of one kernel id with events on the launch stream
void prof_before(int kid, hipStream_t s);
void prof_after(int kid, hipStream_t s);
// timing diagnostic: MPHSIR_DIAG_SKIP=name[,name...] (mphsir_kernel_name values) turns the launches of those kernel families into
// no-ops -- their outputs stay uninitialised, results are GARBAGE -- so that the change of a step's wall time shows what the family
// costs on the critical path (its kernel time minus whatever ran beside it).  Announced on stderr; never set in tests or the bench.
bool diag_skip(int kid);

#define MPHSIR_REQUIRE(cond, ...)          \
    do {                                   \
        if (!(cond)) {                     \
            mphsir::set_error(__VA_ARGS__); \
            return MPHSIR_EINVAL;          \
        }                                  \
    } while (0)

// Every mphsir_*_args struct starts with struct_size = sizeof(the struct as the CALLER compiled it): a caller built against an
// older include/mphsir.h (the structs grow at the end) is refused instead of having fields read past the end of its struct.
#define MPHSIR_CHECK_ARGS(a, fn)                                                                                          \
    do {                                                                                                                  \
        MPHSIR_REQUIRE((a) != nullptr, "%s: null pointer (args)", fn);                                                     \
        MPHSIR_REQUIRE((a)->struct_size == (uint32_t)sizeof(*(a)),                                                         \
                       "%s: args struct_size %u, this library expects %u (built against another include/mphsir.h?)", fn,   \
                       (unsigned)(a)->struct_size, (unsigned)sizeof(*(a)));                                                \
    } while (0)

// Launches `kern` (a __global__ function, already instantiated) and reports launch errors.
#define MPHSIR_LAUNCH(kid, kern, grid, block, shmem, stream, ...)                                   \
    do {                                                                                            \
        mphsir::prof_before(kid, stream);                                                           \
        if (!mphsir::diag_skip(kid)) hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__); \
        mphsir::prof_after(kid, stream);                                                            \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) {                                                                     \
            mphsir::set_error("%s: launch failed: %s", mphsir_kernel_name(kid), hipGetErrorString(e_)); \
            return MPHSIR_ELAUNCH;                                                                  \
        }                                                                                           \
    } while (0)

// element-type dispatch of the templated kernels: EXPR is evaluated with T_ bound to float / bf16_t / f16_t
#define MPHSIR_DTYPE_OK(dt) ((dt) == MPHSIR_F32 || (dt) == MPHSIR_BF16 || (dt) == MPHSIR_F16)
#define MPHSIR_DISPATCH_T(dt, EXPR)                                                  \
    ((dt) == MPHSIR_F32    ? [&]() -> int { using T_ = float; return EXPR; }()       \
     : (dt) == MPHSIR_BF16 ? [&]() -> int { using T_ = mphsir::bf16_t; return EXPR; }() \
                           : [&]() -> int { using T_ = mphsir::f16_t; return EXPR; }())
inline int dtype_size(int dt) { return dt == MPHSIR_F32 ? 4 : 2; }

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// dynamic LDS above 64 KiB must be opted into per kernel function; done once per (function, size) so that replays /
// stream captures of a warmed-up step issue no runtime-attribute calls
bool lds_attr_needed(const void* fn, size_t bytes);
template <class K> inline void allow_big_lds(K kern, size_t bytes) {
    if (bytes > 48 * 1024 && lds_attr_needed(reinterpret_cast<const void*>(kern), bytes))
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

}  // namespace mphsir
