// libmphsir: version / error text / launch timer.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>
#include <vector>

#include "mphsir_host.h"

namespace mphsir {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
void clear_error() { g_err[0] = 0; }

bool lds_attr_needed(const void* fn, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> done;
    std::lock_guard<std::mutex> lock(mu);
    size_t& cur = done[fn];
    if (cur >= bytes) return false;
    cur = bytes;
    return true;
}

struct ProfLog {
    int kid = -1;
    std::vector<hipEvent_t> pool;   // start/stop pairs, reused
    size_t used = 0;
};
static ProfLog g_prof;

bool diag_skip(int kid) {
    static const unsigned long long mask = [] {
        unsigned long long m = 0;
        const char* e = getenv("MPHSIR_DIAG_SKIP");
        if (e && *e) {
            for (int k = 0; k < 64; ++k) {
                const char* n = mphsir_kernel_name(k);
                if (!n || !*n || n[0] == '?') continue;
                const size_t len = strlen(n);
                for (const char* p = strstr(e, n); p; p = strstr(p + 1, n))
                    if ((p == e || p[-1] == ',') && (p[len] == 0 || p[len] == ',')) m |= 1ull << k;
            }
            fprintf(stderr, "mphsir: MPHSIR_DIAG_SKIP=%s -- these kernel families are NOT launched (mask %llx): timing diagnostic, results are garbage\n", e, m);
        }
        return m;
    }();
    return kid >= 0 && kid < 64 && ((mask >> kid) & 1);
}

void prof_before(int kid, hipStream_t s) {
    if (kid != g_prof.kid) return;
    if (g_prof.used + 2 > g_prof.pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        g_prof.pool.push_back(a);
        g_prof.pool.push_back(b);
    }
    (void)hipEventRecord(g_prof.pool[g_prof.used], s);
}
void prof_after(int kid, hipStream_t s) {
    if (kid != g_prof.kid || g_prof.used + 2 > g_prof.pool.size()) return;
    (void)hipEventRecord(g_prof.pool[g_prof.used + 1], s);
    g_prof.used += 2;
}

}  // namespace mphsir

extern "C" {

const char* mphsir_version(void) { return "1.0.0-gfx950"; }
const char* mphsir_last_error(void) { return mphsir::g_err; }

int mphsir_device_arch(char* buf, int n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (!buf || n <= 0) return MPHSIR_EINVAL;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        mphsir::set_error("device_arch: no HIP device");
        return MPHSIR_ELAUNCH;
    }
    strncpy(buf, prop.gcnArchName, (size_t)n - 1);
    buf[n - 1] = 0;
    return MPHSIR_OK;
}

const char* mphsir_kernel_name(int kid) {
    static const char* names[] = {"gemm_tok", "win_attn", "dwconv_gram", "spectral_fold", "gated_mlp", "dwconv_gate", "flat_adamw", "dwconv3x3", "dwconv3x3_wgrad", "gated_mlp_bwd", "combine_bwd", "win_attn_bwd", "ln_bwd_win", "gemm_tn", "gdfn_gate_bwd", "spectral_fold_bwd", "pg_gate_bwd", "conv3x3_tok", "im2col3x3", "reduce_parts", "pack_gather", "layernorm_tok", "pg_gate", "resample", "qkv_dwconv_gram", "gdfn_fused", "dwconv3x3_bwd", "multi_copy", "l1_clamp_loss", "gated_mlp_wgrad", "spectral_dqkv_bwd", "layout"};
    return (kid >= 0 && kid < (int)(sizeof(names) / sizeof(names[0]))) ? names[kid] : "?";
}

int64_t mphsir_gemm_tn_workspace_bytes(int32_t N1, int32_t N2, int32_t nsplit, int32_t batch, int32_t with_colsum) {
    if (N1 <= 0 || N2 <= 0 || nsplit <= 0 || batch <= 0) return MPHSIR_EINVAL;
    return (int64_t)batch * nsplit * ((int64_t)N1 * N2 + (with_colsum ? N1 : 0)) * 4;
}
int64_t mphsir_dwconv_gram_workspace_bytes(int32_t B, int32_t nsplit, int32_t C, int32_t heads) {
    if (B <= 0 || nsplit <= 0 || heads <= 0 || C % heads != 0) return MPHSIR_EINVAL;
    const int64_t hd = C / heads;
    return (int64_t)B * nsplit * ((int64_t)heads * hd * hd + 2 * (int64_t)C) * 4;
}
int64_t mphsir_pg_gate_bwd_workspace_bytes(int32_t nW, int32_t C, int32_t r, int dtype, int32_t* KL, int32_t* KR) {
    if (nW <= 0 || C <= 0 || r <= 0) return MPHSIR_EINVAL;
    const int kl = (C + 5 * r + 256 + 7) / 8 * 8, kr = (5 * r + 1 + C + 7) / 8 * 8;
    if (KL) *KL = kl;
    if (KR) *KR = kr;
    return (int64_t)nW * (kl + kr) * (dtype == MPHSIR_F32 ? 4 : 2);
}
int64_t mphsir_win_attn_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C, int32_t heads, int dtype) {
    if (B <= 0 || H % 8 || W % 8 || heads <= 0) return MPHSIR_EINVAL;
    const int64_t M = (int64_t)B * H * W, esz = dtype == MPHSIR_F32 ? 4 : 2;
    return M * 5 * C * esz + (M / 64) * 225 * heads * 4;
}

int mphsir_prof_enable(int kid) {
    mphsir::g_prof.kid = kid;
    mphsir::g_prof.used = 0;
    return MPHSIR_OK;
}

int mphsir_prof_read(int* launches, float* total_ms) {
    using mphsir::g_prof;
    float sum = 0.f;
    int n = 0;
    for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
        float ms = 0.f;
        if (hipEventSynchronize(g_prof.pool[i + 1]) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms, g_prof.pool[i], g_prof.pool[i + 1]) != hipSuccess) continue;
        sum += ms;
        ++n;
    }
    g_prof.used = 0;
    if (launches) *launches = n;
    if (total_ms) *total_ms = sum;
    return MPHSIR_OK;
}

}  // extern "C"
