// flat_adamw: one AdamW step over the flat fp32 parameter arena (all trainable tensors of the model
// laid out back to back), with the data-parallel gradient mean (1/world) folded in.
//
// Replaces torch.optim.AdamW(self.parameters(), lr) of the reference (train.py:69; defaults
// betas (0.9,0.999), eps 1e-8, weight_decay 1e-2, decoupled decay) and the 1/world_size scaling DDP
// applies after its all-reduce (train.py:118, strategy="auto").  Pure HBM streaming: 16 B per lane,
// grid-stride, 4 reads + 3 writes of 4 B per element.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct AdamDev {
    float* p; const float* g; float* m; float* v;
    long n;
    float lr, beta1, beta2, eps, wd, grad_scale, bc1, bc2_sqrt;
    const float* hyper;     // optional device [lr, bias_correction1, sqrt(bias_correction2)]: graph-replay friendly
    const float* scaler;    // optional device loss-scaler state (see mphsir_scaler_update): gradients are divided by scaler[0],
                            // the whole update is skipped when scaler[2] != 0, and the Adam step count is scaler[3] + 1
};

__global__ __launch_bounds__(256) void flat_adamw_kernel(AdamDev a) {
    if (a.hyper) { a.lr = a.hyper[0]; a.bc1 = a.hyper[1]; a.bc2_sqrt = a.hyper[2]; }
    if (a.scaler) {
        if (a.scaler[2] != 0.f) return;                       // a non-finite gradient somewhere: skip the step (GradScaler.step)
        const float t = a.scaler[3] + 1.0f;                   // skipped steps do not advance Adam's step count
        a.bc1 = 1.0f - powf(a.beta1, t);
        a.bc2_sqrt = sqrtf(1.0f - powf(a.beta2, t));
        a.grad_scale /= a.scaler[0];
    }
    const long nvec = a.n / 4;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        f32x4 p = reinterpret_cast<f32x4*>(a.p)[i];
        const f32x4 g = reinterpret_cast<const f32x4*>(a.g)[i];
        f32x4 m = reinterpret_cast<f32x4*>(a.m)[i];
        f32x4 v = reinterpret_cast<f32x4*>(a.v)[i];
        for (int e = 0; e < 4; ++e) {
            const float ge = g[e] * a.grad_scale;
            p[e] *= 1.0f - a.lr * a.wd;
            m[e] = a.beta1 * m[e] + (1.0f - a.beta1) * ge;
            v[e] = a.beta2 * v[e] + (1.0f - a.beta2) * ge * ge;
            const float denom = sqrtf(v[e]) / a.bc2_sqrt + a.eps;
            p[e] -= (a.lr / a.bc1) * (m[e] / denom);
        }
        reinterpret_cast<f32x4*>(a.p)[i] = p;
        reinterpret_cast<f32x4*>(a.m)[i] = m;
        reinterpret_cast<f32x4*>(a.v)[i] = v;
    }
}

// ---- dynamic loss scaling (the reference trains precision="16-mixed", train.py:118 = torch GradScaler) --------------------
// scaler state, device fp32[4]: [0] loss scale, [1] growth tracker (good steps in a row), [2] found-inf flag of the current
// step, [3] optimizer steps actually taken.
__global__ __launch_bounds__(256) void grad_check_kernel(const float* __restrict__ g, long n, float* scaler) {
    const long nvec = n / 4, stride = (long)gridDim.x * 256;
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        // |x| <= FLT_MAX is false for inf and for NaN
        bad = bad || !(fabsf(v[0]) <= 3.402823466e38f) || !(fabsf(v[1]) <= 3.402823466e38f) || !(fabsf(v[2]) <= 3.402823466e38f) ||
              !(fabsf(v[3]) <= 3.402823466e38f);
    }
    if (bad) scaler[2] = 1.0f;          // every writer stores the same value: no ordering needed
}

__global__ void scaler_update_kernel(float* scaler, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (scaler[2] != 0.f) {             // GradScaler.update(): overflow -> back off, restart the streak
        scaler[0] *= backoff;
        scaler[1] = 0.f;
    } else {
        scaler[3] += 1.0f;
        scaler[1] += 1.0f;
        if (scaler[1] >= (float)interval) { scaler[0] *= growth; scaler[1] = 0.f; }
    }
    scaler[2] = 0.f;
}

}  // namespace mphsir

extern "C" int mphsir_grad_check(const float* g, int64_t n, float* scaler, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(g && scaler && n > 0 && n % 4 == 0 && aligned16(g), "grad_check: bad arguments");
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    MPHSIR_LAUNCH(MPHSIR_K_FLAT_ADAMW, grad_check_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g, (long)n, scaler);
    return MPHSIR_OK;
}

extern "C" int mphsir_scaler_update(float* scaler, float growth_factor, float backoff_factor, int32_t growth_interval, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(scaler && growth_factor >= 1.0f && backoff_factor > 0.f && backoff_factor <= 1.0f && growth_interval > 0, "scaler_update: bad arguments");
    MPHSIR_LAUNCH(MPHSIR_K_FLAT_ADAMW, scaler_update_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), scaler, growth_factor,
                  backoff_factor, (int)growth_interval);
    return MPHSIR_OK;
}

extern "C" int mphsir_flat_adamw_scaled(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                        float weight_decay, float grad_scale, const float* hyper, const float* scaler, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(p && g && m && v && scaler, "flat_adamw_scaled: null pointer");
    MPHSIR_REQUIRE(n > 0 && n % 4 == 0, "flat_adamw_scaled: arena length must be a positive multiple of 4 (pad the arena)");
    MPHSIR_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "flat_adamw_scaled: 16-byte alignment required");
    AdamDev d{p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, grad_scale, 1.0f, 1.0f, hyper, scaler};
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    MPHSIR_LAUNCH(MPHSIR_K_FLAT_ADAMW, flat_adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), d);
    return MPHSIR_OK;
}

extern "C" int mphsir_flat_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                 float beta2, float eps, float weight_decay, int32_t step, float grad_scale, const float* hyper,
                                 void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(p && g && m && v, "flat_adamw: null pointer");
    MPHSIR_REQUIRE(n > 0 && n % 4 == 0, "flat_adamw: arena length must be a positive multiple of 4 (pad the arena)");
    MPHSIR_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v), "flat_adamw: 16-byte alignment required");
    MPHSIR_REQUIRE(step >= 1 || hyper, "flat_adamw: step is 1-based");
    AdamDev d{p, g, m, v, (long)n, lr, beta1, beta2, eps, weight_decay, grad_scale,
              1.0f - powf(beta1, (float)(step > 0 ? step : 1)), sqrtf(1.0f - powf(beta2, (float)(step > 0 ? step : 1))), hyper, nullptr};
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    MPHSIR_LAUNCH(MPHSIR_K_FLAT_ADAMW, flat_adamw_kernel, dim3((unsigned)blocks), dim3(256), 0,
                  reinterpret_cast<hipStream_t>(stream), d);
    return MPHSIR_OK;
}
