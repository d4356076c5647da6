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
};

__global__ __launch_bounds__(256) void flat_adamw_kernel(AdamDev a) {
    if (a.hyper) { a.lr = a.hyper[0]; a.bc1 = a.hyper[1]; a.bc2_sqrt = a.hyper[2]; }
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

}  // namespace mphsir

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
              1.0f - powf(beta1, (float)(step > 0 ? step : 1)), sqrtf(1.0f - powf(beta2, (float)(step > 0 ? step : 1))), hyper};
    long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    MPHSIR_LAUNCH(MPHSIR_K_FLAT_ADAMW, flat_adamw_kernel, dim3((unsigned)blocks), dim3(256), 0,
                  reinterpret_cast<hipStream_t>(stream), d);
    return MPHSIR_OK;
}
