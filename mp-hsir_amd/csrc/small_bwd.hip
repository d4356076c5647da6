// Backward of the two "tiny math" stages of a PGSSTB block, one launch each instead of ~40 library ops:
//
//   spectral_fold_bwd  backward of mphsir_spectral_fold for one (sample, head) per workgroup: from dM = dL/dM_b it
//       recomputes the hd x hd attention A and produces dG, d(sum q^2), d(sum k^2), d temperature and the
//       per-sample part of d project_out.  dG and the norm terms are emitted as ONE per-sample matrix
//       W2_b [2C][2C] so that  [dq | dk] = [q | k] W2_b^T  is a single token GEMM (mphsir_gemm_tok):
//           rows c      (q gradients): [ diag(2 dsq) | Nq   ],  Nq[c][c'] = dG_h[i][j] inside head blocks
//           rows C + c' (k gradients): [ Nq^T        | diag(2 dsk) ]
//       (reference forward: Spectral_Attention.forward net/MP_HSIR.py:104-113).
//   pg_gate_bwd  backward of the local spectral-prompt gate (PG_Spectral_Attention.forward :132-152) per window:
//       d mu and the per-window left/right factor rows whose token-reduction product (ONE mphsir_gemm_tn over the
//       windows) contains every parameter gradient of the branch as a sub-block.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct FoldBwdDev {
    const float* Gpart; const float* Spart; int nsplit;
    const float* temperature; const float* Wo;
    const float* dM;            // [B][C][C] fp32, or [B][dm_nsplit][C][C]: the split partials of the token-reduction GEMM that produced it
    int dm_nsplit;              // (summed in split order while they are staged: the ordered-sum launch between the two kernels is gone)
    const float* dm_scale;      // optional [B]: dM_b is multiplied by dm_scale[b] (the DropPath factor of a d_out that was handed over unscaled)
    const void* DO; long lddo; const void* V; long ldv; int N;      // N > 0 (16-bit types, small images): dM is not read but FORMED here -- dM_b =
                                                                      // d_out_b^T v_b over the sample's N tokens (rows b N .. of DO [.][lddo], V [.][ldv])
    void* W2;                   // [B][2C][2C] compute dtype
    int w2_blocks;              // only the head's own column blocks are written (what spectral_dqkv_bwd reads)
    float* dWo;                 // [B][C][C] fp32 per-sample partial (column block of this head written by its WG)
    float* dtemp;               // [B][HEADS]
    int B, C, HD;
};

// One workgroup per (sample, head).  The two products with the head slices of project_out and dM,
//     dA[i][j]   = sum_co Wo[co][hHD+i] dM[co][hHD+j]          (K = output channels, streamed in chunks of FB_CO rows)
//     dWo[co][i] = sum_j  dM[co][hHD+j] A[i][j]                (K = head dim)
// run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32; the first one gathers its K-strided fragments with scalar
// LDS reads, no transposed copy); the softmax forward / backward rows are shared by 4 lanes each.  Everything is
// fp32 and every sum has a fixed order: bitwise reproducible.
constexpr int FB_THREADS = 1024, FB_WAVES = FB_THREADS / 64;   // only B*heads (64..256) workgroups exist: make each one wide
__host__ __device__ constexpr int fb_co(int hd) { return hd > 64 ? 32 : 128; }     // rows per chunk = 16 per wave (LDS budget at head_dim 96)
// tokens per stage of the in-kernel dM: as many as the LDS holds beside the fp32 tiles -- a stage costs one exposed round trip to L2 / HBM
// (one tile in flight: the 128-register budget of a 1024-thread workgroup has no room for a deeper prefetch), so fewer, larger stages
__host__ __device__ constexpr int fb_tk(int hd) { return hd == 48 ? 128 : 64; }      // (head_dim 32 at 128 / 256 tokens per stage: 10 / 18 spilled registers)

// W2_b rows of one head: q rows h HD + i and k rows C + h HD + j (see the file header).  16-byte stores: a thread builds VEC consecutive
// columns of a row (element-wise 2-byte stores made this phase 5-10 us of the launch).  blocks: only the row's two column blocks of this
// head -- [h HD, (h+1) HD) and [C + h HD, C + (h+1) HD) -- the rest of W2_b is zero and the fused backward never reads it.
// (Not inlined: inside the kernel it cost the 32-wide heads two spilled registers at the 128 a 1024-thread workgroup has.)
template <class T>
__device__ __attribute__((noinline)) void fold_bwd_write_w2(T* W2, int blocks, const float* D, const float* sq, const float* dn, const float* rq, const float* rk,
                                               int HD, int LD, int C, int h, int tid) {
    constexpr int VEC = Vec16<T>::N;
    const int C2 = 2 * C, hg = HD / VEC, gpr = blocks ? 2 * hg : C2 / VEC;      // column groups per row
    // entry (row r of the head's q rows [k rows: kside], column col) of W2_b
    auto entry = [&](int r, int col, bool kside) __attribute__((always_inline)) -> float {
        const int cq = col - h * HD, ck = col - C - h * HD;              // position inside the head's q-column / k-column block
        if (!kside) {
            if (cq == r) return sq[r] > 1e-24f ? dn[r] * rq[r] : 0.f;                      // 2 * dsq = dnq / nq
            return (ck >= 0 && ck < HD) ? D[r * LD + ck] : 0.f;                            // dG[i][j]
        }
        if (ck == r) return sq[HD + r] > 1e-24f ? dn[HD + r] * rk[r] : 0.f;
        return (cq >= 0 && cq < HD) ? D[cq * LD + r] : 0.f;                                // Nq^T
    };
#pragma unroll 1
    for (int o = tid; o < 2 * HD * gpr; o += FB_THREADS) {
        const int rr = o / gpr, g = o % gpr;
        const bool kside = rr >= HD;
        const int r = kside ? rr - HD : rr;
        const int c0g = blocks ? (g < hg ? h * HD + g * VEC : C + h * HD + (g - hg) * VEC) : g * VEC;
        Vec16<T> ov;
#pragma unroll
        for (int e = 0; e < VEC; ++e) ov.set(e, entry(r, c0g + e, kside));
        store16<T>(W2 + (long)((kside ? C : 0) + h * HD + r) * C2 + c0g, ov);
    }
}

template <class T, int HDT>      // HDT > 0: the head width as a compile-time constant (loops over it unroll), 0: runtime
__global__ __launch_bounds__(FB_THREADS) void spectral_fold_bwd_kernel(FoldBwdDev a) {
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    const int HD = HDT > 0 ? HDT : a.HD, C = a.C, HEADS = C / HD, LD = HD + 8, NT = HD / 16, FB_CO = fb_co(HD);
    float* G = reinterpret_cast<float*>(smem_v);      // raw Gram                  [HD][LD]
    float* A = G + HD * LD;                           // probabilities             [HD][LD]
    float* D = A + HD * LD;                           // dA -> dGtilde -> dG       [HD][LD]
    float* Ws = D + HD * LD;                          // [FB_CO][LD] rows of Wo[:, head]
    float* Ms = Ws + FB_CO * LD;                      // [FB_CO][LD] rows of dM[:, head]
    float* nq = Ms + FB_CO * LD;                      // [2*HD] clamped norms (nq | nk)
    float* rn = nq + 2 * HD;                          // [2*HD] their reciprocals
    float* sq = rn + 2 * HD;                          // [2*HD] raw sums of squares
    float* dn = sq + 2 * HD;                          // [2*HD] d nq | d nk
    float* red = dn + 2 * HD;                         // [HD] per-row d temperature terms
    // in-kernel dM (a.N > 0): token tiles of d_out[:, c0 .. c0 + FB_CO) and v[:, head] in the storage type, 64 tokens per stage
    constexpr int FB_TK = fb_tk(HDT > 0 ? HDT : 64), FB_PAD = 8, FB_NR = FB_TK / 64;
    const int LDD = FB_CO + FB_PAD, LDV = HD + FB_PAD;
    T* Dt = reinterpret_cast<T*>(red + ((HD + 3) & ~3));          // [FB_TK][LDD]
    T* Vt = Dt + FB_TK * LDD;                                      // [FB_TK][LDV]
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform(), b = blockIdx.x / HEADS, h = blockIdx.x % HEADS;
    const float temp = a.temperature[h];
    const float dms = a.dm_scale ? a.dm_scale[b] : 1.f;

    for (int i = tid; i < HD * HD; i += FB_THREADS) {        // ordered sum over the splits (1 when the forward saved the sums)
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
        G[(i / HD) * LD + i % HD] = s;
    }
    if (tid < 2 * HD) {
        float s = 0.f;
#pragma unroll 8
        for (int sp = 0; sp < a.nsplit; ++sp) s += a.Spart[((long)b * a.nsplit + sp) * 2 * C + (tid / HD) * C + h * HD + tid % HD];
        sq[tid] = s;
        const float n = fmaxf(sqrtf(s), 1e-12f);      // F.normalize eps
        nq[tid] = n;
        rn[tid] = 1.0f / n;
    }
    __syncthreads();
    const float* rq = rn;
    const float* rk = rn + HD;
    const int qd = tid & 3;                           // 4 adjacent lanes per row (rows >= HD idle but take part in the shuffles)
    for (int row = tid >> 2; row < (HD + FB_THREADS / 4 - 1) / (FB_THREADS / 4) * (FB_THREADS / 4); row += FB_THREADS / 4) {      // A = softmax_rows(G / (nq nk^T) * temp)
        const bool on = row < HD;
        const float tq = on ? temp * rq[row] : 0.f;
        float m = -3.0e38f;
        if (on) for (int j = qd; j < HD; j += 4) m = fmaxf(m, G[row * LD + j] * tq * rk[j]);
        m = fmaxf(m, __shfl_xor(m, 1));
        m = fmaxf(m, __shfl_xor(m, 2));
        float den = 0.f;
        if (on) for (int j = qd; j < HD; j += 4) { const float e = expf(G[row * LD + j] * tq * rk[j] - m); A[row * LD + j] = e; den += e; }
        den += __shfl_xor(den, 1);
        den += __shfl_xor(den, 2);
        const float inv = 1.0f / den;
        if (on) for (int j = qd; j < HD; j += 4) A[row * LD + j] *= inv;
    }
    const int nsp = a.dm_nsplit > 1 ? a.dm_nsplit : 1;
    const long CC = (long)C * C;
    const float* dM = a.dM + (long)b * nsp * CC;
    float* dWo = a.dWo + (long)b * C * C;
    constexpr int NS = (36 + FB_WAVES - 1) / FB_WAVES;   // dA tiles per wave: t = wv + FB_WAVES*s < NT*NT <= 36 (head_dim <= 96)
    f32x4 accA[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) accA[s] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lr = lane & 15, lk = 4 * (lane >> 4);
    for (int c0 = 0; c0 < C; c0 += FB_CO) {
        __syncthreads();                              // previous chunk consumed (first time: A complete)
        bool formed = false;
        if constexpr (sizeof(T) == 2 && HDT > 0 && HDT <= 64) {
            if (a.N > 0) {
                // ---- dM rows c0 .. c0 + FB_CO of this head, formed here: Ms[rr][cc] = sum_tok d_out[tok][c0 + rr] v[tok][h HD + cc].  Both
                // operands are token-major (K-strided): 64-token tiles go through LDS and are read as MFMA fragments by the transposed
                // reads; the next tile's rows are requested from L2 before this tile's barrier.  (The lower pyramid levels -- 256 / 1024
                // tokens per sample -- where the token-reduction GEMM this replaces was a 15-22 us launch on the critical path.)
                typedef typename ElemTraits<T>::frag_t frag_t;
                const T* DOb = reinterpret_cast<const T*>(a.DO) + (long)b * a.N * a.lddo + c0;
                const T* Vb = reinterpret_cast<const T*>(a.V) + (long)b * a.N * a.ldv + h * HD;
                const int cw = (C - c0) < FB_CO ? (C - c0) : FB_CO;       // valid rows of this chunk
                const int dvr = FB_CO / 8, vvr = HD / 8;                   // 16-byte vectors per token row
                constexpr int NTM = (fb_co(HDT) / 16) * (HDT / 16), TPWM = (NTM + FB_WAVES - 1) / FB_WAVES;
                f32x4 accM[TPWM];
#pragma unroll
                for (int q = 0; q < TPWM; ++q) accM[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                Vec16<T> dreg[FB_NR], vreg;
                // d_out: 64 token rows x FB_CO / 8 vectors = the 1024 threads, FB_NR times; v: FB_TK rows x HD / 8 vectors <= 1024
                const int dtok = tid / dvr, dcol = (tid % dvr) * 8, vtok = tid / vvr, vcol = (tid % vvr) * 8;
                const bool von = tid < FB_TK * vvr;
                auto tload = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
                    for (int r = 0; r < FB_NR; ++r) {
                        const int tk = t0 + r * 64 + dtok;
                        if (dcol < cw && tk < a.N) dreg[r] = load16<T>(DOb + (long)tk * a.lddo + dcol); else dreg[r] = Vec16<T>{};
                    }
                    if (von) { if (t0 + vtok < a.N) vreg = load16<T>(Vb + (long)(t0 + vtok) * a.ldv + vcol); else vreg = Vec16<T>{}; }
                };
                tload(0);
                for (int t0 = 0; t0 < a.N; t0 += FB_TK) {
#pragma unroll
                    for (int r = 0; r < FB_NR; ++r) store16<T>(Dt + (r * 64 + dtok) * LDD + dcol, dreg[r]);
                    if (von) store16<T>(Vt + vtok * LDV + vcol, vreg);
                    __syncthreads();
                    if (t0 + FB_TK < a.N) tload(t0 + FB_TK);
#pragma unroll
                    for (int q = 0; q < TPWM; ++q) {
                        const int tt = wv + FB_WAVES * q;              // wave-uniform: output tile (16 rows of the chunk, 16 head columns)
                        if (tt < (FB_CO / 16) * NT) {
                            const int ti = tt / NT, tj = tt % NT;
#pragma unroll
                            for (int k0 = 0; k0 < FB_TK; k0 += 32)
                                mma(accM[q], load_frag_tr<T>(Dt, LDD, ti * 16, k0), load_frag_tr<T>(Vt, LDV, tj * 16, k0));
                        }
                    }
                    __syncthreads();
                }
#pragma unroll
                for (int q = 0; q < TPWM; ++q) {
                    const int tt = wv + FB_WAVES * q;
                    if (tt < (FB_CO / 16) * NT) {
                        const int ti = tt / NT, tj = tt % NT;
                        for (int r = 0; r < 4; ++r) Ms[(ti * 16 + (lane >> 4) * 4 + r) * LD + tj * 16 + (lane & 15)] = accM[q][r] * dms;
                    }
                }
                for (int idx = tid; idx < FB_CO * (HD / 4); idx += FB_THREADS) {
                    const int rr = idx / (HD / 4), cc = (idx % (HD / 4)) * 4;
                    *reinterpret_cast<f32x4*>(Ws + rr * LD + cc) = c0 + rr < C ? *reinterpret_cast<const f32x4*>(a.Wo + (long)(c0 + rr) * C + h * HD + cc) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
                formed = true;
            }
        }
        if (!formed)
        // Ws / Ms rows of this chunk: 16-byte vectors along the head's columns.  The dM rows are the ordered sum of the token-reduction
        // GEMM's split partials: ALL the splits of a vector are requested before the first is added (in split order, eight at a time) --
        // until round 6 a thread walked its elements one after the other with four loads in flight, 24 dependent round trips at the
        // level-1 shape (12 splits): most of the launch's 36 us.
        for (int idx = tid; idx < FB_CO * (HD / 4); idx += FB_THREADS) {
            const int rr = idx / (HD / 4), cc = (idx % (HD / 4)) * 4;
            const bool in = c0 + rr < C;
            f32x4 wv4 = f32x4{0.f, 0.f, 0.f, 0.f}, m = f32x4{0.f, 0.f, 0.f, 0.f};
            if (in) {
                wv4 = *reinterpret_cast<const f32x4*>(a.Wo + (long)(c0 + rr) * C + h * HD + cc);
                const float* mp = dM + (long)(c0 + rr) * C + h * HD + cc;
                int sp = 0;
                for (; sp + 8 <= nsp; sp += 8) {
                    f32x4 tv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) tv[u] = *reinterpret_cast<const f32x4*>(mp + (sp + u) * CC);
#pragma unroll
                    for (int u = 0; u < 8; ++u) m += tv[u];
                }
                if (sp + 4 <= nsp) {
                    f32x4 tv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) tv[u] = *reinterpret_cast<const f32x4*>(mp + (sp + u) * CC);
#pragma unroll
                    for (int u = 0; u < 4; ++u) m += tv[u];
                    sp += 4;
                }
                for (; sp < nsp; ++sp) m += *reinterpret_cast<const f32x4*>(mp + sp * CC);
            }
            *reinterpret_cast<f32x4*>(Ws + rr * LD + cc) = wv4;
            *reinterpret_cast<f32x4*>(Ms + rr * LD + cc) = m * dms;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int t = wv + FB_WAVES * s;
            if (t < NT * NT) {                        // wave-uniform
                const int ti = t / NT, tj = t % NT;
                for (int k0 = 0; k0 < FB_CO; k0 += 16) {
                    f32x4 fa, fb;
                    for (int e = 0; e < 4; ++e) {
                        fa[e] = Ws[(k0 + lk + e) * LD + ti * 16 + lr];
                        fb[e] = Ms[(k0 + lk + e) * LD + tj * 16 + lr];
                    }
                    mma(accA[s], fa, fb);
                }
            }
        }
        if (wv * 16 < FB_CO && c0 + wv * 16 < C)       // dWo rows c0 + 16*wv .. +15 of this head's columns
            for (int ti = 0; ti < NT; ++ti) {
                f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int k0 = 0; k0 < HD; k0 += 16)
                    mma(acc, *reinterpret_cast<const f32x4*>(Ms + (wv * 16 + lr) * LD + k0 + lk),
                        *reinterpret_cast<const f32x4*>(A + (ti * 16 + lr) * LD + k0 + lk));
                for (int r = 0; r < 4; ++r) {
                    const int co = c0 + wv * 16 + (lane >> 4) * 4 + r;
                    if (co < C) dWo[(long)co * C + h * HD + ti * 16 + lr] = acc[r];
                }
            }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int t = wv + FB_WAVES * s;
        if (t < NT * NT) {
            const int ti = t / NT, tj = t % NT;
            for (int r = 0; r < 4; ++r) D[(ti * 16 + (lane >> 4) * 4 + r) * LD + tj * 16 + lr] = accA[s][r];
        }
    }
    __syncthreads();
    for (int row = tid >> 2; row < (HD + FB_THREADS / 4 - 1) / (FB_THREADS / 4) * (FB_THREADS / 4); row += FB_THREADS / 4) {      // softmax backward per row; logits = Gtilde * temp
        const bool on = row < HD;
        float rs = 0.f;
        if (on) for (int j = qd; j < HD; j += 4) rs += A[row * LD + j] * D[row * LD + j];
        rs += __shfl_xor(rs, 1);
        rs += __shfl_xor(rs, 2);
        float dt = 0.f, dnq = 0.f;
        if (on)
            for (int j = qd; j < HD; j += 4) {
                const float dl = A[row * LD + j] * (D[row * LD + j] - rs);
                const float gt = G[row * LD + j] * rq[row] * rk[j];
                dt += dl * gt;
                const float dgt = dl * temp;
                dnq -= dgt * gt * rq[row];
                D[row * LD + j] = dgt;                // keep dGtilde
            }
        dt += __shfl_xor(dt, 1);  dt += __shfl_xor(dt, 2);
        dnq += __shfl_xor(dnq, 1);  dnq += __shfl_xor(dnq, 2);
        if (on && qd == 0) { red[row] = dt; dn[row] = dnq; }
    }
    __syncthreads();
    for (int row = tid >> 2; row < (HD + FB_THREADS / 4 - 1) / (FB_THREADS / 4) * (FB_THREADS / 4); row += FB_THREADS / 4) {      // d nk[j] = -sum_i dGtilde[i][j] * Gtilde[i][j] / nk[j]
        const bool on = row < HD;                     // "row" = column j here
        float s = 0.f;
        if (on) for (int i = qd; i < HD; i += 4) s -= D[i * LD + row] * G[i * LD + row] * rq[i] * rk[row] * rk[row];
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (on && qd == 0) dn[HD + row] = s;
    }
    if (wv == 0) {
        const float v = wave_sum((lane < HD ? red[lane] : 0.f) + (lane + 64 < HD ? red[lane + 64] : 0.f));
        if (lane == 0) a.dtemp[b * HEADS + h] = v;
    }
    __syncthreads();
    for (int o = tid; o < HD * HD; o += FB_THREADS) {        // dG[i][j] = dGtilde[i][j] / (nq[i] nk[j])
        const int i = o / HD, j = o % HD;
        D[i * LD + j] *= rq[i] * rk[j];
    }
    __syncthreads();
    fold_bwd_write_w2<T>(reinterpret_cast<T*>(a.W2) + (long)b * 4 * C * C, a.w2_blocks, D, sq, dn, rq, rk, HD, LD, C, h, tid);
}

}  // namespace mphsir

extern "C" int mphsir_spectral_fold_bwd(const mphsir_fold_bwd_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "spectral_fold_bwd");
    MPHSIR_REQUIRE(a && a->Gpart && a->Spart && a->temperature && a->Wo && (a->dM || a->N > 0) && a->W2 && a->dWo && a->dtemp, "spectral_fold_bwd: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "spectral_fold_bwd: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(a->B > 0 && a->heads > 0 && a->C % a->heads == 0 && a->nsplit > 0, "spectral_fold_bwd: bad shape");
    const int HD = a->C / a->heads;
    MPHSIR_REQUIRE(HD <= 96, "spectral_fold_bwd: head_dim %d > 96", HD);
    FoldBwdDev d{a->Gpart, a->Spart, a->nsplit, a->temperature, a->Wo, a->dM, a->dM_nsplit, a->dm_scale, a->DO, (long)a->lddo, a->V, (long)a->ldv, a->N, a->W2, a->w2_blocks, a->dWo, a->dtemp, a->B, a->C, HD};
    if (a->N > 0)
        MPHSIR_REQUIRE(dtype != MPHSIR_F32 && (HD == 32 || HD == 48 || HD == 64) && a->DO && a->V && aligned16(a->DO) && aligned16(a->V) &&
                           (a->lddo * 2) % 16 == 0 && (a->ldv * 2) % 16 == 0 && a->lddo >= a->C && a->ldv >= a->C,
                       "spectral_fold_bwd: forming dM in the kernel (N > 0) needs a 16-bit type, a head width of 32 / 48 / 64 and 16-byte aligned DO / V rows");
    MPHSIR_REQUIRE(HD % 16 == 0, "spectral_fold_bwd: head_dim %d must be a multiple of 16", HD);
    const size_t shmem = ((3 * (size_t)HD + 2 * fb_co(HD)) * (HD + 8) + 9 * (size_t)HD + 4) * sizeof(float) +
                         (a->N > 0 ? (size_t)fb_tk(HD) * (fb_co(HD) + 8 + HD + 8) * 2 : 0);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MPHSIR_FB_LAUNCH(T_, HD_)                                                                                                       \
    do {                                                                                                                              \
        allow_big_lds(spectral_fold_bwd_kernel<T_, HD_>, shmem);                                                                      \
        MPHSIR_LAUNCH(MPHSIR_K_FOLD_BWD, (spectral_fold_bwd_kernel<T_, HD_>), dim3(a->B * a->heads), dim3(FB_THREADS), shmem, s, d);   \
        return MPHSIR_OK;                                                                                                             \
    } while (0)
#define MPHSIR_FB_HD(T_)                               \
    do {                                               \
        if (HD == 32) MPHSIR_FB_LAUNCH(T_, 32);        \
        if (HD == 48) MPHSIR_FB_LAUNCH(T_, 48);        \
        if (HD == 64) MPHSIR_FB_LAUNCH(T_, 64);        \
        if (HD == 96) MPHSIR_FB_LAUNCH(T_, 96);        \
        MPHSIR_FB_LAUNCH(T_, 0);                       \
    } while (0)
    if (dtype == MPHSIR_F32) MPHSIR_FB_HD(float);
    if (dtype == MPHSIR_BF16) MPHSIR_FB_HD(bf16_t);
    MPHSIR_FB_HD(f16_t);
#undef MPHSIR_FB_HD
#undef MPHSIR_FB_LAUNCH
}
