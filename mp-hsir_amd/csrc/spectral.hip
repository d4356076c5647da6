// Global ("spectral") channel attention, the north-star kernel family.
//
// Reference op sequence (Spectral_Attention.forward net/MP_HSIR.py:96-114, == Attention/MDTA :301-322,
// and CrossAttention.forward :234-249):  t = dwconv3x3(conv1x1(x));  q,k,v = split(t);
// q,k L2-normalised over the H*W pixels;  A_h = softmax(q_h k_h^T * temperature_h)  (hd x hd per head);
// out = project_out(A v).   Folded form: out = M_b v with the per-sample matrix
// M_b = W_o * blockdiag_h(A_h), so only the raw Gram  G_h = sum_p q_h[:,p] k_h[:,p]^T  and the two
// sum-of-squares vectors need a pass over the pixels.
//
//   dwconv_gram   pass A: depthwise 3x3 of the 1x1-conv output (q,k,v sources may be different
//                 tensors -> cross attention), v written once, q/k kept on-chip (transposed tile in LDS)
//                 and reduced into per-workgroup Gram / sum-of-squares partials by MFMA with
//                 K = pixels.  Deterministic split-K: fixed tile -> workgroup map, no atomics.
//   spectral_fold per (sample, head): ordered reduction of the partials, F.normalize (eps 1e-12)
//                 scaling, temperature, row softmax, fold with project_out -> M_b (compute dtype).
//   pass B        is mphsir_gemm_tok with the per-sample weight M_b (epi 2 adds the PGSSTB branch sum).
#include <stdlib.h>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct GramDev {
    const void* Tq; long ldq; const void* Tk; long ldk; const void* Tv; long ldv;   // 1x1-conv outputs [B*H*W][ld]
    const float* wq; const float* wk; const float* wv; long ldw;                      // depthwise taps [9][ldw] fp32
    void* V; long ldvo;                                                               // [B*H*W][ldvo] out
    float* Gpart;   // [B][nsplit][HEADS][HD][HD]
    float* Spart;   // [B][nsplit][2][C]
    int B, H, W, nsplit;
    void* QK; long ldqk;                                                              // optional [B*H*W][ldqk]: q | k (post-dwconv)
};

// depthwise 3x3 (zero padding) at pixel (y,x) for VEC consecutive channels starting at c0
template <class T>
__device__ __forceinline__ void dw3x3_vec(const T* base, long ld, const float* w9, long ldw, int c0, int y, int x,
                                          int H, int W, float* acc) {
    constexpr int VEC = Vec16<T>::N;
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            const Vec16<T> t = load16<T>(base + ((long)yy * W + xx) * ld + c0);
            const float* w = w9 + ((dy + 1) * 3 + (dx + 1)) * ldw + c0;
            for (int e = 0; e < VEC; ++e) acc[e] += t.get(e) * w[e];
        }
    }
}

template <class T, int C, int HD>
__global__ __launch_bounds__(256) void dwconv_gram_kernel(GramDev a) {
    typedef ElemTraits<T> TR;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr int VEC = Vec16<T>::N;
    constexpr int HEADS = C / HD;
    constexpr int LDT = 64 + PAD;
    constexpr int NT = HD / 16;                          // Gram tiles per side
    constexpr int SLOTS = (NT * NT + 3) / 4;             // tiles per wave
    constexpr int VPH = HD / VEC;                        // channel vectors per head
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* qT = reinterpret_cast<T*>(smem_v);                // [HD][LDT]  q_h^T : channel-major, pixel-contiguous
    T* kT = qT + HD * LDT;                               // [HD][LDT]

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int b = blockIdx.x / a.nsplit, sp = blockIdx.x % a.nsplit;
    const int HW = a.H * a.W, tiles = HW / 64, tpw = tiles / a.nsplit;
    const long img = (long)b * HW;
    const T* Tq = reinterpret_cast<const T*>(a.Tq) + img * a.ldq;
    const T* Tk = reinterpret_cast<const T*>(a.Tk) + img * a.ldk;
    const T* Tv = reinterpret_cast<const T*>(a.Tv) + img * a.ldv;
    T* V = reinterpret_cast<T*>(a.V) + img * a.ldvo;

    f32x4 g[HEADS][SLOTS];
    float ssq[HEADS];                                    // thread t < 2*HD owns row t of [q_h; k_h]
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
        ssq[h] = 0.f;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) g[h][s] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    for (int tile = sp * tpw; tile < (sp + 1) * tpw; ++tile) {
#pragma unroll
        for (int h = 0; h < HEADS; ++h) {
            // depthwise conv of q_h, k_h (-> LDS, transposed) and v_h (-> HBM) for 64 pixels
            for (int idx = tid; idx < 64 * VPH; idx += 256) {
                const int pi = idx / VPH, cv = idx % VPH, c0 = h * HD + cv * VEC;
                const int p = tile * 64 + pi, y = p / a.W, x = p % a.W;
                float acc[VEC];
                dw3x3_vec<T>(Tq, a.ldq, a.wq, a.ldw, c0, y, x, a.H, a.W, acc);
                for (int e = 0; e < VEC; ++e) qT[(cv * VEC + e) * LDT + pi] = from_f32<T>(acc[e]);
                dw3x3_vec<T>(Tk, a.ldk, a.wk, a.ldw, c0, y, x, a.H, a.W, acc);
                for (int e = 0; e < VEC; ++e) kT[(cv * VEC + e) * LDT + pi] = from_f32<T>(acc[e]);
                dw3x3_vec<T>(Tv, a.ldv, a.wv, a.ldw, c0, y, x, a.H, a.W, acc);
                Vec16<T> vo;
                for (int e = 0; e < VEC; ++e) vo.set(e, acc[e]);
                store16<T>(V + (long)p * a.ldvo + c0, vo);
            }
            __syncthreads();
            if (tid < 2 * HD) {   // sum of squares of the (rounded) rows: the values the Gram sees
                const T* row = qT + tid * LDT;       // rows HD..2HD-1 are kT
                float s = 0.f;
                for (int i = 0; i < 64; ++i) { const float v = to_f32(row[i]); s += v * v; }
                ssq[h] += s;
            }
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int t = wv + 4 * s;            // wave-uniform
                if (t < NT * NT) {
                    const int ti = t / NT, tj = t % NT;
#pragma unroll
                    for (int kk = 0; kk < 64; kk += TR::KCHUNK)
                        mma(g[h][s], load_frag<T>(qT, LDT, ti * 16, kk), load_frag<T>(kT, LDT, tj * 16, kk));
                }
            }
            __syncthreads();
        }
    }

    float* Gp = a.Gpart + (long)blockIdx.x * HEADS * HD * HD;
    float* Sp = a.Spart + (long)blockIdx.x * 2 * C;
#pragma unroll
    for (int h = 0; h < HEADS; ++h) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int t = wv + 4 * s;
            if (t < NT * NT) {
                const int ti = t / NT, tj = t % NT;
                for (int r = 0; r < 4; ++r)
                    Gp[(h * HD + ti * 16 + (lane >> 4) * 4 + r) * HD + tj * 16 + (lane & 15)] = g[h][s][r];
            }
        }
        if (tid < 2 * HD) Sp[(tid / HD) * C + h * HD + tid % HD] = ssq[h];
    }
}

// ---- dwconv_gram, second form: the depthwise pass of ALL heads of a 64-pixel tile runs first with a sliding 3x3
// window (one thread = one channel vector x one strip of 8 pixels x one of q/k/v: 3 loads per output instead of 27,
// the first form was L1/TA-bandwidth bound), q and k land transposed in LDS as whole 16-byte rows of 8 pixels, then
// the per-head Gram / sum-of-squares reductions run from LDS exactly as before.
template <class T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
    Vec16<bf16_t> o;
    for (int i = 0; i < 8; ++i) o.set(i, v[i]);
    store16<bf16_t>(p, o);
}
template <> __device__ __forceinline__ void store8<f16_t>(f16_t* p, const float (&v)[8]) {
    Vec16<f16_t> o;
    for (int i = 0; i < 8; ++i) o.set(i, v[i]);
    store16<f16_t>(p, o);
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

template <class T, int C, int HD>
__global__ __launch_bounds__(256) void dwconv_gram2_kernel(GramDev a) {
    typedef ElemTraits<T> TR;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr int VEC = Vec16<T>::N;
    constexpr int HEADS = C / HD;
    constexpr int LDT = 64 + PAD;
    constexpr int NT = HD / 16;
    constexpr int SLOTS = (NT * NT + 3) / 4;
    constexpr int CV = C / VEC;
    constexpr int NSQ = (2 * C + 255) / 256;             // rows of [q;k] per thread for the final sums of squares
    constexpr int NITEM = (3 * CV * 8 + 255) / 256;      // (q|k|v, channel vector, 8-pixel strip) items per thread: fixed per thread
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    // q and k tiles as [pixel][channel] (pitch LDP): a thread's 8 channels of one pixel are one 16-byte store and the lanes of
    // a wave cover consecutive chunks of a pixel row -- conflict-free.  (The first form kept the tiles transposed,
    // [channel][pixel]: its 16-byte stores landed 8 rows = 1280 B apart, all on the same banks: 0.8 conflict cycles per
    // LDS cycle, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE.)  The Gram's operands are K-strided now: load_frag_tr.
    constexpr int LDP = C + PAD;
    T* qT = reinterpret_cast<T*>(smem_v);                // [64][LDP]
    T* kT = qT + 64 * LDP;                               // [64][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int b = blockIdx.x / a.nsplit, sp = blockIdx.x % a.nsplit;
    const int HW = a.H * a.W, tiles = HW / 64, tpw = tiles / a.nsplit;
    const long img = (long)b * HW;
    T* V = reinterpret_cast<T*>(a.V) + img * a.ldvo;

    f32x4 g[HEADS][SLOTS];
    // sums of squares of the (rounded) q / k values: accumulated where the values are produced -- a thread owns the same
    // (channel vector, strip) items in every tile -- and reduced over the 8 strips once at the end.  (Reading the rows back
    // from the transposed LDS tiles, one 2-byte element per lane at a 160-byte pitch, was an 8-way bank conflict and cost
    // more than the Gram MFMAs of the tile.)
    float ssq[NITEM][VEC];
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) g[h][s] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NITEM; ++i)
        for (int e = 0; e < VEC; ++e) ssq[i][e] = 0.f;

    for (int tile = sp * tpw; tile < (sp + 1) * tpw; ++tile) {
#pragma unroll
        for (int slot = 0; slot < NITEM; ++slot) {
            const int it = tid + 256 * slot;
            if (it >= 3 * CV * 8) break;
            const int which = it / (CV * 8), rem = it % (CV * 8), cvec = rem % CV, st = rem / CV;
            const int c0 = cvec * VEC;
            const int p0 = tile * 64 + st * 8, y = p0 / a.W, x0 = p0 % a.W;
            const T* src = (which == 0 ? reinterpret_cast<const T*>(a.Tq) + img * a.ldq
                            : which == 1 ? reinterpret_cast<const T*>(a.Tk) + img * a.ldk
                                         : reinterpret_cast<const T*>(a.Tv) + img * a.ldv) + c0;
            const long ld = which == 0 ? a.ldq : which == 1 ? a.ldk : a.ldv;
            const float* wsrc = (which == 0 ? a.wq : which == 1 ? a.wk : a.wv) + c0;
            float w[9][VEC];
#pragma unroll
            for (int t = 0; t < 9; ++t)
                for (int e = 0; e < VEC; ++e) w[t][e] = wsrc[t * a.ldw + e];
            auto column = [&](int x, Vec16<T> (&col)[3]) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int yy = y + r - 1;
                    if (x >= 0 && x < a.W && yy >= 0 && yy < a.H) col[r] = load16<T>(src + ((long)yy * a.W + x) * ld);
                    else col[r] = Vec16<T>{};
                }
            };
            Vec16<T> cl[3], cm[3], cr[3];
            float out[VEC][8];
            column(x0 - 1, cl);
            column(x0, cm);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                column(x0 + i + 1, cr);
                for (int e = 0; e < VEC; ++e) {
                    float s = 0.f;
#pragma unroll
                    for (int r = 0; r < 3; ++r)
                        s += cl[r].get(e) * w[r * 3][e] + cm[r].get(e) * w[r * 3 + 1][e] + cr[r].get(e) * w[r * 3 + 2][e];
                    out[e][i] = s;
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
            }
            if (which < 2) {
                T* dst = (which == 0 ? qT : kT) + (st * 8) * LDP + c0;
                T* QK = a.QK ? reinterpret_cast<T*>(a.QK) + img * a.ldqk + which * C + c0 : nullptr;      // training: q | k kept for the backward
                for (int e = 0; e < VEC; ++e) {
                    float s2 = 0.f;
                    for (int i = 0; i < 8; ++i) { const float r = to_f32(from_f32<T>(out[e][i])); s2 += r * r; }     // the values the Gram sees
                    ssq[slot][e] += s2;
                }
                for (int i = 0; i < 8; ++i) {
                    Vec16<T> o;
                    for (int e = 0; e < VEC; ++e) o.set(e, out[e][i]);
                    store16<T>(dst + i * LDP, o);
                    if (QK) store16<T>(QK + (long)(p0 + i) * a.ldqk, o);
                }
            } else {
                for (int i = 0; i < 8; ++i) {
                    Vec16<T> o;
                    for (int e = 0; e < VEC; ++e) o.set(e, out[e][i]);
                    store16<T>(V + (long)(p0 + i) * a.ldvo + c0, o);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < HEADS; ++h)
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int t = wv + 4 * s;
                if (t < NT * NT) {
                    const int ti = t / NT, tj = t % NT;
#pragma unroll
                    for (int kk = 0; kk < 64; kk += TR::KCHUNK)
                        mma(g[h][s], load_frag_tr<T>(qT, LDP, h * HD + ti * 16, kk), load_frag_tr<T>(kT, LDP, h * HD + tj * 16, kk));
                }
            }
        __syncthreads();
    }

    float* Gp = a.Gpart + (long)blockIdx.x * HEADS * HD * HD;
    float* Sp = a.Spart + (long)blockIdx.x * 2 * C;
#pragma unroll
    for (int h = 0; h < HEADS; ++h)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int t = wv + 4 * s;
            if (t < NT * NT) {
                const int ti = t / NT, tj = t % NT;
                for (int r = 0; r < 4; ++r)
                    Gp[(h * HD + ti * 16 + (lane >> 4) * 4 + r) * HD + tj * 16 + (lane & 15)] = g[h][s][r];
            }
        }
    // strip partials -> LDS [2C][8] (the q/k tiles are free: the loop ended with a barrier), ordered sum over the 8 strips
    float* red = reinterpret_cast<float*>(smem_v);
#pragma unroll
    for (int slot = 0; slot < NITEM; ++slot) {
        const int it = tid + 256 * slot;
        if (it < 2 * CV * 8) {                               // which < 2
            const int which = it / (CV * 8), rem = it % (CV * 8), cvec = rem % CV, st = rem / CV;
            for (int e = 0; e < VEC; ++e) red[(which * C + cvec * VEC + e) * 8 + st] = ssq[slot][e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NSQ; ++i) {
        const int row = tid + 256 * i;
        if (row < 2 * C) {
            float s = 0.f;
            for (int st = 0; st < 8; ++st) s += red[row * 8 + st];
            Sp[row] = s;                                     // [2][C]: q rows then k rows
        }
    }
}

struct FoldDev {
    const float* Gpart; const float* Spart; int nsplit;
    const float* temperature;   // [HEADS]
    const float* Wo;            // project_out weight [C][C] fp32
    void* Mout;                 // [B][C][C] compute dtype
    void* MTout;                // optional [B][C][C]: M^T (training: dv = d_out M is a token GEMM with weight M^T)
    int B, C, HD;
    float* Gsum; float* Ssum;   // optional: reduced Gram [B][HEADS][HD][HD] and sums of squares [B][2][C]
    int co;                     // output rows of project_out per workgroup
};

// grid = (B*HEADS, C/co): every workgroup redoes the (tiny) reduction + softmax of its head and folds `co` output rows of
// project_out, whose head slice is staged through LDS.  co = 32 while that keeps the grid small, up to 128 for wide nets (at
// C=256 / 8 heads co=32 meant 2048 workgroups of 1024 threads, 8 of them redoing each softmax: 33 us a launch).
constexpr int FOLD_CO = 32, FOLD_THREADS = 1024;      // few workgroups: make each one wide

// HDT > 0: the head width as a compile-time constant (32, 48, 64, 96): the HD-long dot products and row loops unroll
template <class T, int HDT>
__global__ __launch_bounds__(FOLD_THREADS) void spectral_fold_kernel(FoldDev a) {
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    const int HD = HDT > 0 ? HDT : a.HD, C = a.C, HEADS = C / HD;
    float* G = reinterpret_cast<float*>(smem_v);      // [HD][HD+1] -> attention probabilities
    const int LDG = HD + 1;
    float* nq = G + HD * LDG;                         // [HD]
    float* nk = nq + HD;                              // [HD]
    float* Ws = nk + HD;                              // [co][HD+1] project_out rows of this workgroup
    const int CO = a.co;
    const int tid = threadIdx.x, b = blockIdx.x / HEADS, h = blockIdx.x % HEADS, co0 = blockIdx.y * CO;

    for (int i = tid; i < HD * HD; i += FOLD_THREADS) {        // ordered (deterministic) reduction over the splits
        // ordered sum over the splits; 8 independent loads in flight per step (the loads, not the adds, are the latency)
        const float* gp = a.Gpart + ((long)b * a.nsplit * HEADS + h) * HD * HD + i;
        const long gstride = (long)HEADS * HD * HD;
        float s = 0.f;
        int sp = 0;
        for (; sp + 8 <= a.nsplit; sp += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = gp[(sp + u) * gstride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; sp < a.nsplit; ++sp) s += gp[sp * gstride];
        G[(i / HD) * LDG + i % HD] = s;
        if (a.Gsum && blockIdx.y == 0) a.Gsum[((long)b * HEADS + h) * HD * HD + i] = s;
    }
    if (tid < 2 * HD) {
        float s = 0.f;
#pragma unroll 8
        for (int sp = 0; sp < a.nsplit; ++sp) s += a.Spart[((long)b * a.nsplit + sp) * 2 * C + (tid / HD) * C + h * HD + tid % HD];
        nq[tid] = fmaxf(sqrtf(s), 1e-12f);            // F.normalize eps (nk follows nq in memory)
        if (a.Ssum && blockIdx.y == 0) a.Ssum[(long)b * 2 * C + (tid / HD) * C + h * HD + tid % HD] = s;
    }
    for (int i = tid; i < CO * HD; i += FOLD_THREADS)
        Ws[(i / HD) * LDG + i % HD] = a.Wo[(long)(co0 + i / HD) * C + h * HD + i % HD];
    __syncthreads();
    for (int row = tid >> 2; row < (HD + FOLD_THREADS / 4 - 1) / (FOLD_THREADS / 4) * (FOLD_THREADS / 4); row += FOLD_THREADS / 4) {
        // row softmax of G/(nq nk^T) * temperature: 4 adjacent lanes per row (rows >= HD idle but take part in the shuffles)
        const bool on = row < HD;
        const int qd = tid & 3;
        const float tq = on ? a.temperature[h] / nq[row] : 0.f;
        float m = -3.0e38f;
        if (on) for (int j = qd; j < HD; j += 4) m = fmaxf(m, G[row * LDG + j] * tq / nk[j]);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        float den = 0.f;
        if (on) for (int j = qd; j < HD; j += 4) { const float e = expf(G[row * LDG + j] * tq / nk[j] - m); G[row * LDG + j] = e; den += e; }
        den += __shfl_xor(den, 1);
        den += __shfl_xor(den, 2);
        const float inv = 1.0f / den;
        if (on) for (int j = qd; j < HD; j += 4) G[row * LDG + j] *= inv;
    }
    __syncthreads();
    T* M = reinterpret_cast<T*>(a.Mout) + (long)b * C * C;
    for (int o = tid; o < CO * HD; o += FOLD_THREADS) {   // M[co][h*HD+j] = sum_i Wo[co][h*HD+i] A[i][j]
        const int cl = o / HD, j = o % HD;
        float s = 0.f;
        for (int i = 0; i < HD; ++i) s += Ws[cl * LDG + i] * G[i * LDG + j];
        M[(long)(co0 + cl) * C + h * HD + j] = from_f32<T>(s);
        if (a.MTout) reinterpret_cast<T*>(a.MTout)[(long)b * C * C + (long)(h * HD + j) * C + co0 + cl] = from_f32<T>(s);
    }
}

struct GateDev {
    const void* Tin; long ldt;          // [B*H*W][2*HP]
    const float* w9; long ldw;          // [9][2*HP]
    void* U; long ldu;                  // [B*H*W][HP]
    int B, H, W, HP;
    const void* dU; void* dT;           // backward form (tile kernel): dU [B*H*W][HP] in, d(dwconv output) [B*H*W][2*HP] out (both contiguous)
};

// GDFN middle (FFN/FeedForward.forward net/MP_HSIR.py:261-263, :387-389): u = gelu(dw(t)[:HP]) * dw(t)[HP:]
// One thread = one channel vector x one strip of 8 pixels; both halves slide a 3x3 window (3 loads per output).
template <class T>
__global__ __launch_bounds__(256) void dwconv_gate_kernel(GateDev a) {
    constexpr int VEC = Vec16<T>::N, S = 8;
    const int vpp = a.HP / VEC, nsx = a.W / S;
    const long total = (long)a.B * a.H * nsx * vpp;
    const long idx = xcd_contiguous_block() * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c0 = (int)(idx % vpp) * VEC;
    long q = idx / vpp;
    const int x0 = (int)(q % nsx) * S;  q /= nsx;
    const int y = (int)(q % a.H), b = (int)(q / a.H);
    const T* Tin = reinterpret_cast<const T*>(a.Tin) + (long)b * a.H * a.W * a.ldt;
    T* U = reinterpret_cast<T*>(a.U) + (long)b * a.H * a.W * a.ldu + c0;
    float w1[9][VEC], w2[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
        for (int e = 0; e < VEC; ++e) { w1[t][e] = a.w9[t * a.ldw + c0 + e]; w2[t][e] = a.w9[t * a.ldw + a.HP + c0 + e]; }
    auto column = [&](int x, int ch, Vec16<T> (&col)[3]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            if (x >= 0 && x < a.W && yy >= 0 && yy < a.H) col[r] = load16<T>(Tin + ((long)yy * a.W + x) * a.ldt + ch);
            else col[r] = Vec16<T>{};
        }
    };
    Vec16<T> al[3], am[3], ar[3], bl[3], bm[3], br[3];
    column(x0 - 1, c0, al); column(x0, c0, am);
    column(x0 - 1, a.HP + c0, bl); column(x0, a.HP + c0, bm);
#pragma unroll
    for (int i = 0; i < S; ++i) {
        column(x0 + i + 1, c0, ar);
        column(x0 + i + 1, a.HP + c0, br);
        Vec16<T> o;
        for (int e = 0; e < VEC; ++e) {
            float v1 = 0.f, v2 = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                v1 += al[r].get(e) * w1[r * 3][e] + am[r].get(e) * w1[r * 3 + 1][e] + ar[r].get(e) * w1[r * 3 + 2][e];
                v2 += bl[r].get(e) * w2[r * 3][e] + bm[r].get(e) * w2[r * 3 + 1][e] + br[r].get(e) * w2[r * 3 + 2][e];
            }
            o.set(e, Math<T>::gelu(v1) * v2);
        }
        store16<T>(U + ((long)y * a.W + x0 + i) * a.ldu, o);
#pragma unroll
        for (int r = 0; r < 3; ++r) { al[r] = am[r]; am[r] = ar[r]; bl[r] = bm[r]; bm[r] = br[r]; }
    }
}

// Tile form of the GDFN middle (16-bit types, H % 8 == 0, W % 16 == 0): the scheme of dwconv3x3_tile_kernel (dwconv.hip) --
// an 8x16-pixel tile + halo of 48 gelu-side and their 48 partner channels staged in LDS as fp32 through coalesced loads
// that are all in flight at once, each element unpacked once, the 3x3 window sliding over LDS.  The strip form above
// ran at 1.6 TB/s (30 dependent-use loads per thread and half through L1).
constexpr int GT2_TH = 8, GT2_TW = 16, GT2_HW = GT2_TW + 2, GT2_ROWS = (GT2_TH + 2) * GT2_HW, GT2_CH = 48, GT2_LD = 2 * GT2_CH + 4;
constexpr int GT2_THREADS = 512;

// BWD: the same pass recomputes [x1|x2] = dwconv(t) and emits the gate's backward next to u -- d x1 = du x2 gelu'(x1),
// d x2 = du gelu(x1) -- instead of dwconv3x3 writing [x1|x2] to HBM for gdfn_gate_bwd_kernel to read back.
template <class T, bool BWD>
__global__ __launch_bounds__(GT2_THREADS, 4) void dwconv_gate_tile_kernel(GateDev a) {
    constexpr int VEC = Vec16<T>::N;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* Ts = reinterpret_cast<float*>(smem_v);          // [180][GT2_LD]: 48 gelu-side channels | their 48 partners
    float* taps = Ts + GT2_ROWS * GT2_LD;                  // [9][96]
    const int tid = threadIdx.x;
    const int tilesx = a.W / GT2_TW, tiles = (a.H / GT2_TH) * tilesx, nslab = (a.HP + GT2_CH - 1) / GT2_CH;
    const long blk = (gridDim.x & 7) == 0 ? xcd_contiguous_block() : (long)blockIdx.x;
    if (blk >= (long)a.B * tiles * nslab) return;
    const int tile = (int)(blk % tiles), slab = (int)((blk / tiles) % nslab), b = (int)(blk / ((long)tiles * nslab));
    const int ty0 = (tile / tilesx) * GT2_TH, tx0 = (tile % tilesx) * GT2_TW;
    const int cs0 = slab * GT2_CH, cw = (a.HP - cs0) < GT2_CH ? (a.HP - cs0) : GT2_CH, vph = cw / VEC, vpr = 2 * vph;
    const T* Tin = reinterpret_cast<const T*>(a.Tin) + (long)b * a.H * a.W * a.ldt;
    T* U = reinterpret_cast<T*>(a.U) + (long)b * a.H * a.W * a.ldu + cs0;

    constexpr int NV = (GT2_ROWS * (2 * GT2_CH / VEC) + GT2_THREADS - 1) / GT2_THREADS;      // 5
    Vec16<T> xv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + GT2_THREADS * i, r = idx / vpr, v = idx % vpr;
        const int y = ty0 - 1 + r / GT2_HW, x = tx0 - 1 + r % GT2_HW;
        const int ch = v < vph ? cs0 + v * VEC : a.HP + cs0 + (v - vph) * VEC;
        if (r < GT2_ROWS && y >= 0 && y < a.H && x >= 0 && x < a.W) xv[i] = load16<T>(Tin + ((long)y * a.W + x) * a.ldt + ch);
        else xv[i] = Vec16<T>{};
    }
    for (int i = tid; i < 9 * 2 * cw; i += GT2_THREADS) {
        const int t = i / (2 * cw), c = i % (2 * cw);
        taps[t * 2 * GT2_CH + (c < cw ? c : GT2_CH + c - cw)] = a.w9[t * a.ldw + (c < cw ? cs0 + c : a.HP + cs0 + c - cw)];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + GT2_THREADS * i, r = idx / vpr, v = idx % vpr;
        if (r < GT2_ROWS) {
            float* dst = Ts + r * GT2_LD + (v < vph ? v * VEC : GT2_CH + (v - vph) * VEC);
            *reinterpret_cast<f32x4*>(dst) = f32x4{xv[i].get(0), xv[i].get(1), xv[i].get(2), xv[i].get(3)};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{xv[i].get(4), xv[i].get(5), xv[i].get(6), xv[i].get(7)};
        }
    }
    __syncthreads();

    // one thread = (gelu side | partner side, 4 channels, strip of 8 pixels); the two sides of a channel group sit in
    // adjacent lanes and meet through one shuffle per output
    const int qpr = cw / 4;
    const int it = tid >> 1, side = tid & 1;
    const bool on = it < qpr * 16;
    f32x4 res[8];
    int c4 = 0, iy = 0, ix0 = 0;
    if (on) {
        c4 = it % qpr;
        const int st = it / qpr;
        iy = st >> 1; ix0 = (st & 1) * 8;
        const float* tsrc = Ts + (iy * GT2_HW + ix0) * GT2_LD + side * GT2_CH + c4 * 4;
        f32x4 w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const f32x4*>(taps + t * 2 * GT2_CH + side * GT2_CH + c4 * 4);
        auto tvec = [&](int r, int col) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(tsrc + (r * GT2_HW + col) * GT2_LD); };
        f32x4 cl[3], cm[3], cr[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { cl[r] = tvec(r, 0); cm[r] = tvec(r, 1); }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int r = 0; r < 3; ++r) cr[r] = tvec(r, i + 2);
            f32x4 o = cl[0] * w[0];
            o = __builtin_elementwise_fma(cm[0], w[1], o);
            o = __builtin_elementwise_fma(cr[0], w[2], o);
#pragma unroll
            for (int r = 1; r < 3; ++r) {
                o = __builtin_elementwise_fma(cl[r], w[r * 3], o);
                o = __builtin_elementwise_fma(cm[r], w[r * 3 + 1], o);
                o = __builtin_elementwise_fma(cr[r], w[r * 3 + 2], o);
            }
            res[i] = o;
#pragma unroll
            for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) res[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // gelu side (even lane) keeps pixels 0-3 of the strip, the partner side (odd lane) pixels 4-7: each lane sends the other
    // half of its results across and finishes 4 outputs
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x4 mine = side ? res[4 + i] : res[i], send = side ? res[i] : res[4 + i];
        f32x4 other;
        for (int e = 0; e < 4; ++e) other[e] = __shfl_xor(send[e], 1);
        const f32x4 g = side ? other : mine, p = side ? mine : other;      // g: gelu-side value, p: partner
        f32x4 o;
        const long pix = (long)(ty0 + iy) * a.W + tx0 + ix0 + (side ? 4 : 0) + i;
        if constexpr (BWD) {
            const long row = (long)b * a.H * a.W + pix;
            f32x4 d1 = f32x4{0.f, 0.f, 0.f, 0.f}, d2 = d1;
            if (on) {
                const f32x4 du = load4<T>(reinterpret_cast<const T*>(a.dU) + row * a.HP + cs0 + c4 * 4);
                for (int e = 0; e < 4; ++e) {
                    float ge, dge;
                    Math<T>::gelu_pair(g[e], ge, dge);
                    o[e] = ge * p[e];
                    d1[e] = du[e] * p[e] * dge;
                    d2[e] = du[e] * ge;
                }
                T* dT = reinterpret_cast<T*>(a.dT) + row * 2 * a.HP + cs0 + c4 * 4;
                store4<T>(dT, d1);
                store4<T>(dT + a.HP, d2);
            }
        } else {
            for (int e = 0; e < 4; ++e) o[e] = Math<T>::gelu(g[e]) * p[e];
        }
        if (on) store4<T>(U + pix * a.ldu + c4 * 4, o);
    }
}

template <class T, bool BWD = false> static int launch_gate_tile(const GateDev& d, hipStream_t s) {
    const long nblk = (long)d.B * (d.H / GT2_TH) * (d.W / GT2_TW) * ((d.HP + GT2_CH - 1) / GT2_CH);
    const size_t shmem = ((size_t)GT2_ROWS * GT2_LD + 9 * 2 * GT2_CH) * sizeof(float);
    allow_big_lds(dwconv_gate_tile_kernel<T, BWD>, shmem);
    MPHSIR_LAUNCH(BWD ? MPHSIR_K_GDFN_GATE_BWD : MPHSIR_K_DWCONV_GATE, (dwconv_gate_tile_kernel<T, BWD>), dim3((unsigned)((nblk + 7) / 8 * 8)),
                  dim3(GT2_THREADS), shmem, s, d);
    return MPHSIR_OK;
}

template <class T, int C, int HD>
static int launch_gram(const GramDev& d, hipStream_t s) {
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr size_t gram_tiles = 2 * 64 * (size_t)(C + PAD) * sizeof(T), gram_red = 2 * (size_t)C * 8 * sizeof(float);
    constexpr size_t shmem2 = gram_tiles > gram_red ? gram_tiles : gram_red;      // the strip partials of the sums of squares reuse the tiles
    if constexpr (shmem2 <= 160 * 1024) {
        if (d.W % 8 == 0) {
            allow_big_lds(dwconv_gram2_kernel<T, C, HD>, shmem2);
            MPHSIR_LAUNCH(MPHSIR_K_DWCONV_GRAM, (dwconv_gram2_kernel<T, C, HD>), dim3(d.B * d.nsplit), dim3(256), shmem2, s, d);
            return MPHSIR_OK;
        }
    }
    const size_t shmem = 2 * HD * (64 + PAD) * sizeof(T);
    allow_big_lds(dwconv_gram_kernel<T, C, HD>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_DWCONV_GRAM, (dwconv_gram_kernel<T, C, HD>), dim3(d.B * d.nsplit), dim3(256), shmem, s, d);
    return MPHSIR_OK;
}

template <class T>
static int dispatch_gram(const GramDev& d, int C, int HD, hipStream_t s) {
#define MPHSIR_GRAM_CASE(c, hd) if (C == c && HD == hd) return launch_gram<T, c, hd>(d, s);
    MPHSIR_GRAM_CASE(32, 16) MPHSIR_GRAM_CASE(64, 16) MPHSIR_GRAM_CASE(128, 16)
    MPHSIR_GRAM_CASE(32, 32) MPHSIR_GRAM_CASE(64, 32) MPHSIR_GRAM_CASE(64, 64) MPHSIR_GRAM_CASE(128, 32)
    MPHSIR_GRAM_CASE(128, 64) MPHSIR_GRAM_CASE(256, 32)
    MPHSIR_GRAM_CASE(96, 48) MPHSIR_GRAM_CASE(192, 48) MPHSIR_GRAM_CASE(192, 96) MPHSIR_GRAM_CASE(384, 48)
#undef MPHSIR_GRAM_CASE
    set_error("dwconv_gram: (C=%d, head_dim=%d) not instantiated", C, HD);
    return MPHSIR_EINVAL;
}

}  // namespace mphsir

extern "C" int mphsir_dwconv_gram(const mphsir_gram_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "dwconv_gram");
    MPHSIR_REQUIRE(a && a->Tq && a->Tk && a->Tv && a->wq && a->wk && a->wv && a->V && a->Gpart && a->Spart, "dwconv_gram: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "dwconv_gram: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2;
    MPHSIR_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && ((int64_t)a->H * a->W) % 64 == 0, "dwconv_gram: H*W must be a multiple of 64");
    MPHSIR_REQUIRE(a->heads > 0 && a->C % a->heads == 0, "dwconv_gram: C %% heads != 0");
    const int tiles = a->H * a->W / 64;
    MPHSIR_REQUIRE(a->nsplit > 0 && tiles % a->nsplit == 0, "dwconv_gram: nsplit=%d must divide the %d pixel tiles", a->nsplit, tiles);
    MPHSIR_REQUIRE(aligned16(a->Tq) && aligned16(a->Tk) && aligned16(a->Tv) && aligned16(a->V) && (a->ldq * esz) % 16 == 0 &&
                       (a->ldk * esz) % 16 == 0 && (a->ldv * esz) % 16 == 0 && (a->ldvo * esz) % 16 == 0,
                   "dwconv_gram: 16-byte alignment required");
    GramDev d{a->Tq, (long)a->ldq, a->Tk, (long)a->ldk, a->Tv, (long)a->ldv, a->wq, a->wk, a->wv, (long)a->ldw,
              a->V, (long)a->ldvo, a->Gpart, a->Spart, a->B, a->H, a->W, a->nsplit, a->QK, (long)a->ldqk};
    MPHSIR_REQUIRE(!a->QK || mphsir_dwconv_gram_keeps_qk(a->C, a->W, dtype), "dwconv_gram: this shape cannot emit q|k (ask mphsir_dwconv_gram_keeps_qk)");
    MPHSIR_REQUIRE(!a->QK || (aligned16(a->QK) && (a->ldqk * esz) % 16 == 0 && a->ldqk >= 2 * a->C), "dwconv_gram: QK must be 16-byte aligned, ldqk >= 2C");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return MPHSIR_DISPATCH_T(dtype, (dispatch_gram<T_>(d, a->C, a->C / a->heads, s)));
}

extern "C" int mphsir_dwconv_gram_keeps_qk(int32_t C, int32_t W, int dtype) {
    // the sliding-window form (the one that can also emit q | k) needs the q and k tiles of all heads in LDS
    const size_t esz = dtype == MPHSIR_F32 ? 4 : 2;
    return (W % 8 == 0 && 2 * 64 * ((size_t)C + mphsir::LDS_PAD_BYTES / esz) * esz <= 160 * 1024) ? 1 : 0;
}

extern "C" int mphsir_spectral_fold(const mphsir_fold_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "spectral_fold");
    MPHSIR_REQUIRE(a && a->Gpart && a->Spart && a->temperature && a->Wo && a->M, "spectral_fold: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "spectral_fold: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(a->B > 0 && a->heads > 0 && a->C % a->heads == 0 && a->nsplit > 0, "spectral_fold: bad shape");
    const int HD = a->C / a->heads;
    MPHSIR_REQUIRE(HD <= 128, "spectral_fold: head_dim %d > 128", HD);
    MPHSIR_REQUIRE((a->Gsum == nullptr) == (a->Ssum == nullptr), "spectral_fold: Gsum and Ssum go together");
    MPHSIR_REQUIRE(a->C % FOLD_CO == 0, "spectral_fold: C must be a multiple of %d", FOLD_CO);
    int co = FOLD_CO;                                  // rows per workgroup: grow while the grid stays >= 256 workgroups
    while (co < 128 && a->C % (2 * co) == 0 && (long)a->B * a->heads * (a->C / (2 * co)) >= 256) co *= 2;
    FoldDev d{a->Gpart, a->Spart, a->nsplit, a->temperature, a->Wo, a->M, a->MT, a->B, a->C, HD, a->Gsum, a->Ssum, co};
    const size_t shmem = ((size_t)HD * (HD + 1) + 2 * HD + (size_t)co * (HD + 1)) * sizeof(float);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const dim3 grid(a->B * a->heads, a->C / co);
#define MPHSIR_FOLD_LAUNCH(T_, HD_)                                                                                       \
    do {                                                                                                                \
        allow_big_lds(spectral_fold_kernel<T_, HD_>, shmem);                                                            \
        MPHSIR_LAUNCH(MPHSIR_K_SPECTRAL_FOLD, (spectral_fold_kernel<T_, HD_>), grid, dim3(FOLD_THREADS), shmem, s, d);   \
        return MPHSIR_OK;                                                                                               \
    } while (0)
#define MPHSIR_FOLD_HD(T_)                                 \
    do {                                                   \
        if (HD == 32) MPHSIR_FOLD_LAUNCH(T_, 32);          \
        if (HD == 48) MPHSIR_FOLD_LAUNCH(T_, 48);          \
        if (HD == 64) MPHSIR_FOLD_LAUNCH(T_, 64);          \
        if (HD == 96) MPHSIR_FOLD_LAUNCH(T_, 96);          \
        MPHSIR_FOLD_LAUNCH(T_, 0);                         \
    } while (0)
    if (dtype == MPHSIR_F32) MPHSIR_FOLD_HD(float);
    if (dtype == MPHSIR_BF16) MPHSIR_FOLD_HD(bf16_t);
    MPHSIR_FOLD_HD(f16_t);
#undef MPHSIR_FOLD_HD
#undef MPHSIR_FOLD_LAUNCH
}

extern "C" int mphsir_dwconv_gate(const mphsir_gate_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "dwconv_gate");
    MPHSIR_REQUIRE(a && a->T && a->w9 && a->U, "dwconv_gate: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "dwconv_gate: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2;
    MPHSIR_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->HP > 0 && a->HP % 8 == 0, "dwconv_gate: bad shape");
    MPHSIR_REQUIRE(aligned16(a->T) && aligned16(a->U) && (a->ldt * esz) % 16 == 0 && (a->ldu * esz) % 16 == 0, "dwconv_gate: 16-byte alignment required");
    GateDev d{a->T, (long)a->ldt, a->w9, (long)a->ldw, a->U, (long)a->ldu, a->B, a->H, a->W, a->HP, nullptr, nullptr};
    MPHSIR_REQUIRE(a->W % 8 == 0, "dwconv_gate: W must be a multiple of 8");
    const long total = (long)a->B * a->H * (a->W / 8) * (a->HP / (16 / esz));
    const long blocks = ((total + 255) / 256 + 7) / 8 * 8;            // multiple of 8: XCD-contiguous order
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype != MPHSIR_F32 && a->H % GT2_TH == 0 && a->W % GT2_TW == 0)
        return dtype == MPHSIR_BF16 ? launch_gate_tile<bf16_t>(d, s) : launch_gate_tile<f16_t>(d, s);
    if (dtype == MPHSIR_F32)
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV_GATE, (dwconv_gate_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    else if (dtype == MPHSIR_BF16)
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV_GATE, (dwconv_gate_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    else
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV_GATE, (dwconv_gate_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    return MPHSIR_OK;
}

extern "C" int mphsir_dwconv_gate_bwd_fits(int32_t H, int32_t W, int32_t HP, int dtype) {
    return ((dtype == MPHSIR_BF16 || dtype == MPHSIR_F16) && H > 0 && W > 0 && H % mphsir::GT2_TH == 0 && W % mphsir::GT2_TW == 0 && HP > 0 && HP % 8 == 0) ? 1 : 0;
}

extern "C" int mphsir_dwconv_gate_bwd(const void* T, const float* w9, int64_t ldw, const void* dU, void* U, void* dT, int32_t B, int32_t H, int32_t W,
                                      int32_t HP, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(T && w9 && dU && U && dT, "dwconv_gate_bwd: null pointer");
    MPHSIR_REQUIRE(B > 0 && mphsir_dwconv_gate_bwd_fits(H, W, HP, dtype),
                   "dwconv_gate_bwd: (H=%d, W=%d, HP=%d, dtype=%d) not covered (16-bit types, H %% 8 == 0, W %% 16 == 0, HP %% 8 == 0: otherwise "
                   "mphsir_dwconv3x3 + mphsir_gdfn_gate_bwd)", H, W, HP, dtype);
    MPHSIR_REQUIRE(aligned16(T) && aligned16(dU) && aligned16(U) && aligned16(dT) && ldw >= 2 * HP, "dwconv_gate_bwd: 16-byte alignment / ldw");
    GateDev d{T, 2L * HP, w9, (long)ldw, U, (long)HP, B, H, W, HP, dU, dT};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == MPHSIR_BF16 ? launch_gate_tile<bf16_t, true>(d, s) : launch_gate_tile<f16_t, true>(d, s);
}
