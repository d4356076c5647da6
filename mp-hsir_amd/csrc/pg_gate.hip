// pg_gate_fwd / pg_gate_bwd: the local spectral-prompt gate of PG_Spectral_Attention (net/MP_HSIR.py:132-152) and its
// backward, for 16 windows per workgroup.
//
// Per window the gate is a chain of small products on the window mean mu (C):  w = softmax(Wprompt mu) (128),
// s = w^T P (r), q = Wq s, [k;v] = Wkv (Wdown mu), A = softmax_rows(q k^T / sqrt r) (r x r), o = A v,
// g = Wup (Wproj o + b) (C).  Round 1 ran it as the tail of win_attn (one window per workgroup: ~10 barrier-separated
// mat-vecs, every weight re-read from L2 per window; a third of that kernel) and its backward the same way (25 links,
// 0.8 ms per training step).  Here a workgroup owns NWIN = 16 windows, so every link is a small mat-MAT: the two
// C-sized products (Wprompt mu, Wdown mu; in the backward also Wup^T dg and Wprompt^T dlogit) are fp32 MFMA tiles
// (v_mfma_f32_16x16x4_f32: exact f32, windows on the N axis) whose weight fragments are read from L2 once per 16
// windows, the r-sized links run one thread per (window, index), and one barrier serves 16 windows.
// Everything is fp32 (the reference's gate is a handful of fp32 Linear layers on a mean).
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

constexpr int PG_NWIN = 16;     // windows per workgroup = N of the MFMA tiles
constexpr int PG_RMAX = 32;     // low-rank width bound (r = C / compress_ratio: 8..24 in the shipped models)

struct PgDev {
    const float* mu; const float* dgate;
    const float* Wprompt; const float* Pp; const float* Wq; const float* Wkv; const float* Wdown;
    const float* Wpproj; const float* bpproj; const float* Wup;
    float* gate;                // forward:  [nW][C]
    float* dmu;                 // backward: [nW][C]
    void* L; void* R;           // backward: [nW][KL], [nW][KR] factor rows (fp32 or bf16)
    int nW, C, r, KL, KR, lr_bf16;
    int stage_wdn;              // backward: linear_down also staged in LDS (when the budget allows)
    int wpg;                    // windows this workgroup owns (<= PG_NWIN = the MFMA N axis; the rest of it idles): small launches use fewer
    unsigned long long* dbg;    // diagnostics (mphsir_debug): shader-clock stamps of workgroup 0 at the phase boundaries
};
static unsigned long long* g_pg_dbg = nullptr;
#define PG_MARK(k) do { if (a.dbg && blockIdx.x == 0 && threadIdx.x == 0) a.dbg[k] = __builtin_amdgcn_s_memtime(); } while (0)

// fragment of rows row0..row0+15 of a row-major [nrows][ld] fp32 matrix in global memory; rows past nrows-1 are clamped
// (their products land in accumulator rows nobody reads)
__device__ __forceinline__ f32x4 pg_frag_rows(const float* W, int ld, int nrows, int row0, int k0) {
    const int l = lane_id();
    int row = row0 + (l & 15);
    row = row < nrows ? row : nrows - 1;
    return *reinterpret_cast<const f32x4*>(W + (long)row * ld + k0 + 4 * (l >> 4));
}
// fragment of COLUMNS col0..col0+15 of a row-major [K][ld] fp32 matrix in global memory (the operand transposed):
// row (l&15) of the fragment = column col0+(l&15), its K elements = rows k0 + 4*(l>>4) .. +3; columns / rows past the
// matrix are clamped
__device__ __forceinline__ f32x4 pg_frag_cols(const float* W, int ld, int ncols, int nk, int col0, int k0) {
    const int l = lane_id();
    int col = col0 + (l & 15);
    col = col < ncols ? col : ncols - 1;
    const int kb = k0 + 4 * (l >> 4);
    f32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int k = kb + j < nk ? kb + j : nk - 1;
        v[j] = kb + j < nk ? W[(long)k * ld + col] : 0.f;
        (void)k;
    }
    return v;
}

// i / d for 0 <= i < 2^17 and small runtime d without the ~40-instruction integer division: one v_mul_hi_u32
struct FastDiv {
    unsigned d, m;
    __device__ __forceinline__ explicit FastDiv(int dd) : d((unsigned)dd), m(0xFFFFFFFFu / (unsigned)dd + 1u) {}
    __device__ __forceinline__ int div(int i) const { return (int)(((unsigned long long)(unsigned)i * m) >> 32); }
};

struct PgLds {
    float* mu;      // [16][LDC]
    float* dg;      // [16][LDC]   (backward)
    float* w;       // [16][LDW]   prompt logits -> weights
    float* dl;      // [16][LDW]   (backward) d weights -> d logits
    float* P;       // [128][r]    prompt_param staged once
    float* Wq;      // [r][r]  |
    float* Wkv;     // [2r][r] |   the r-sized weights, staged once (one L2 round trip instead of one per link)
    float* Wpp;     // [r][r]  |
    float* bpp;     // [r]     |
    float* Wup;     // [C][r] (+32 floats of slack: column fragments read 16 columns at a time)
    float* Wdn;     // [r][C]      (backward: the rank-r term of d mu)
    float* sm;      // [16][SMW]   small per-window vectors
    float* At;      // [16][LDA]   (backward) attention probabilities, LDA = r*r + 4
    int LDC, LDW, LDA, RP;      // RP = r + 1: the pitch of every r-wide LDS matrix (odd: lanes striding rows or columns never collide)
};
// offsets of the small vectors inside a window's sm row: 15 slots of `pgw` = r floats (kv / dkv take two), + 4 floats so that the
// windows of one wave start a few banks apart (15 r + 4 is never a multiple of 32 for r <= 32) and the per-window broadcast
// reads of the r-sized links do not collide.  (Slots were a fixed 32 floats wide: 31 KB of the kernel's LDS for r = 8, which
// kept it from sharing a CU with the kernels it runs beside on the side stream.)
#define PG_S (0 * pgw)
#define PG_D (1 * pgw)
#define PG_KV (2 * pgw)
#define PG_Q (4 * pgw)
#define PG_O (5 * pgw)
#define PG_O2 (6 * pgw)
#define PG_DO2 (7 * pgw)
#define PG_DO (8 * pgw)
#define PG_DQ (9 * pgw)
#define PG_DKV (10 * pgw)
#define PG_DD (12 * pgw)
#define PG_DS (13 * pgw)
#define PG_RS (14 * pgw)
#define PG_SMW (15 * pgw + 4)

__device__ __forceinline__ PgLds pg_lds(float* base, int C, int r, bool bwd, bool wdn = false) {
    PgLds s;
    const int pgw = r;
    s.LDC = C + 8;                 // 32-byte row padding: conflict-free ds_read_b128 fragments (mphsir_dev.h)
    s.LDW = 128 + 8;
    s.RP = r + 1;
    s.LDA = r * (r + 1) + 4;
    s.mu = base;
    s.dg = s.mu + PG_NWIN * s.LDC;
    s.w = s.dg + (bwd ? PG_NWIN * s.LDC : 0);
    s.dl = s.w + PG_NWIN * s.LDW;
    s.P = s.dl + (bwd ? PG_NWIN * s.LDW : 0);
    s.Wq = s.P + 128 * (r + 1);
    s.Wkv = s.Wq + r * (r + 1);
    s.Wpp = s.Wkv + 2 * r * (r + 1);
    s.bpp = s.Wpp + r * (r + 1) + (4 - (132 * (r + 1)) % 4) % 4;     // keep the following tiles 16-byte aligned
    s.Wup = s.bpp + ((r + 3) & ~3);
    s.Wdn = s.Wup + C * (r + 1) + 32 + (4 - (C * (r + 1)) % 4) % 4;
    s.sm = s.Wdn + (bwd && wdn ? r * C : 0);
    s.At = s.sm + PG_NWIN * PG_SMW;
    return s;
}
static size_t pg_lds_bytes(int C, int r, bool bwd, bool wdn = false) {
    const size_t rp = r + 1;
    const int pgw = r;
    size_t n = (size_t)PG_NWIN * (C + 8) * (bwd ? 2 : 1) + (size_t)PG_NWIN * 136 * (bwd ? 2 : 1) + 132 * rp + 3 + 4 * (size_t)r * rp +
               ((r + 3) & ~3) + (size_t)C * rp + 32 + 3 + (size_t)PG_NWIN * PG_SMW;
    if (bwd) n += (size_t)PG_NWIN * (r * rp + 4) + (wdn ? (size_t)r * C : 0);
    return n * sizeof(float);
}

// acc[t] += W-tile t (16 rows each, read row-wise or column-wise from L2) x the 16-window LDS operand, over K.  A
// workgroup has its CU to itself, so nothing hides an L2 / HBM round trip but the kernel's own ordering: the weight
// fragments of 8 K-steps (128 of K) are requested together (pg_load_chunk), and the first chunk of every product is
// requested at the very top of the kernel, before the staging loads and their barrier.
template <int NT, bool COLS>
__device__ __forceinline__ void pg_load_chunk(f32x4 (&wf)[8][NT], const float* W, int ld, int nrows, int nk, const int (&row0)[NT], int k0, int K) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (k0 + 16 * i < K)
                wf[i][t] = COLS ? pg_frag_cols(W, ld, nrows, nk, row0[t], k0 + 16 * i) : pg_frag_rows(W, ld, nrows, row0[t], k0 + 16 * i);
}
template <int NT>
__device__ __forceinline__ void pg_mma_chunk(f32x4 (&acc)[NT], const f32x4 (&wf)[8][NT], const float* B, int ldb, int k0, int K) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (k0 + 16 * i < K) {
            const f32x4 b = load_frag<float>(B, ldb, 0, k0 + 16 * i);
#pragma unroll
            for (int t = 0; t < NT; ++t) mma(acc[t], wf[i][t], b);
        }
}
// chunk 0 already in `wf` (prefetched); the remaining chunks (K > 128) are loaded as they come
template <int NT, bool COLS>
__device__ __forceinline__ void pg_mfma_rows(f32x4 (&acc)[NT], f32x4 (&wf)[8][NT], const float* W, int ld, int nrows, int nk,
                                             const int (&row0)[NT], const float* B, int ldb, int K) {
    pg_mma_chunk<NT>(acc, wf, B, ldb, 0, K);
    for (int k0 = 128; k0 < K; k0 += 128) {
        pg_load_chunk<NT, COLS>(wf, W, ld, nrows, nk, row0, k0, K);
        pg_mma_chunk<NT>(acc, wf, B, ldb, k0, K);
    }
}

// ---- the forward chain for the workgroup's 16 windows (shared by both kernels); ends with o2 in sm[.][PG_O2] --------------
// R > 0: the low-rank width as a compile-time constant (8, 12, 16 in the shipped nets): every r-long dot product of the chain is
// a per-thread loop over LDS whose trip count the compiler must know to unroll it and keep its reads in flight together
template <bool KEEP_AT, int R>
__device__ __forceinline__ void pg_forward_chain(const PgDev& a, const PgLds& s, int win0) {
    const int C = a.C, r = R > 0 ? R : a.r, pgw = r, tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform(), nwl = a.wpg;
    // the weight fragments of the two C-sized products are requested first ...
    const int rows[2] = {wv * 32, wv * 32 + 16}, rowd[1] = {wv * 16};
    const bool has_d = wv * 16 < r;
    PG_MARK(0);
    f32x4 wfp[8][2], wfd[8][1];
    pg_load_chunk<2, false>(wfp, a.Wprompt, C, 128, 0, rows, 0, C);
    if (has_d) pg_load_chunk<1, false>(wfd, a.Wdown, C, r, 0, rowd, 0, C);
    // ... then everything else the chain will read: the mu tile (windows past nW: zeros), prompt_param, the r-sized
    // weights, linear_up (and linear_down for the backward's rank-r term) -- all in flight together, one barrier
    const FastDiv byC(C), byR(r);
#pragma unroll 4
    for (int i = tid; i < nwl * C; i += 256) {
        const int w = byC.div(i), c = i - w * C;
        s.mu[w * s.LDC + c] = win0 + w < a.nW ? a.mu[(long)(win0 + w) * C + c] : 0.f;
    }
    const int RP = s.RP;
#pragma unroll 4
    for (int i = tid; i < 128 * r; i += 256) { const int q = byR.div(i); s.P[q * RP + i - q * r] = a.Pp[i]; }
    for (int i = tid; i < r * r; i += 256) {
        const int q = byR.div(i), o = q * RP + i - q * r;
        s.Wq[o] = a.Wq[i]; s.Wpp[o] = a.Wpproj[i]; s.Wkv[o] = a.Wkv[i]; s.Wkv[r * RP + o] = a.Wkv[r * r + i];
    }
    if (tid < r) s.bpp[tid] = a.bpproj[tid];
#pragma unroll 4
    for (int i = tid; i < C * r; i += 256) { const int q = byR.div(i); s.Wup[q * RP + i - q * r] = a.Wup[i]; }
    if (tid < 32) s.Wup[C * RP + tid] = 0.f;                    // the slack behind the last row (column fragments overrun by < 32)
    if (KEEP_AT && a.stage_wdn)
        for (int i = tid; i < r * C; i += 256) s.Wdn[i] = a.Wdown[i];
    __syncthreads();
    PG_MARK(1);
    // logits = Wprompt mu (128 rows: two 16-row tiles per wave) and d = Wdown mu (r rows: waves 0..ceil(r/16)-1)
    {
        f32x4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        pg_mfma_rows<2, false>(acc, wfp, a.Wprompt, C, 128, 0, rows, s.mu, s.LDC, C);
        const int w = lane & 15, rr = (lane >> 4) * 4;
        for (int j = 0; j < 4; ++j) {
            s.w[w * s.LDW + wv * 32 + rr + j] = acc[0][j];
            s.w[w * s.LDW + wv * 32 + 16 + rr + j] = acc[1][j];
        }
        if (has_d) {
            f32x4 accd[1] = {{0, 0, 0, 0}};
            pg_mfma_rows<1, false>(accd, wfd, a.Wdown, C, r, 0, rowd, s.mu, s.LDC, C);
            for (int j = 0; j < 4; ++j)
                if (wv * 16 + rr + j < r) s.sm[w * PG_SMW + PG_D + wv * 16 + rr + j] = accd[0][j];
        }
    }
    __syncthreads();
    PG_MARK(2);
    // softmax over the 128 logits: wave wv owns windows 4wv .. 4wv+3
    for (int ww = 0; ww < 4; ++ww) {
        if (wv * 4 + ww >= nwl) break;               // wave-uniform
        float* lw = s.w + (wv * 4 + ww) * s.LDW;
        const float l0 = lw[lane], l1 = lw[lane + 64];
        float m = fmaxf(l0, l1);
        for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
        const float e0 = expf(l0 - m), e1 = expf(l1 - m);
        const float tot = wave_sum(e0 + e1);
        lw[lane] = e0 / tot;
        lw[lane + 64] = e1 / tot;
    }
    // kv = Wkv d  (one thread per (window, row))
    const FastDiv by2R(2 * r);
    for (int i = tid; i < nwl * 2 * r; i += 256) {
        const int w = by2R.div(i), m = i - w * 2 * r;
        const float* d = s.sm + w * PG_SMW + PG_D;
        float acc = 0.f;
        for (int j = 0; j < r; ++j) acc += s.Wkv[m * RP + j] * d[j];
        s.sm[w * PG_SMW + PG_KV + m] = acc;
    }
    __syncthreads();
    PG_MARK(3);
    // s = w^T P
    for (int i = tid; i < nwl * r; i += 256) {
        const int w = byR.div(i), j = i - w * r;
        const float* lw = s.w + w * s.LDW;
        float acc = 0.f;
        for (int p = 0; p < 128; ++p) acc += lw[p] * s.P[p * RP + j];
        s.sm[w * PG_SMW + PG_S + j] = acc;
    }
    __syncthreads();
    PG_MARK(4);
    // q = Wq s
    for (int i = tid; i < nwl * r; i += 256) {
        const int w = byR.div(i), m = i - w * r;
        const float* sv = s.sm + w * PG_SMW + PG_S;
        float acc = 0.f;
        for (int j = 0; j < r; ++j) acc += s.Wq[m * RP + j] * sv[j];
        s.sm[w * PG_SMW + PG_Q + m] = acc;
    }
    __syncthreads();
    PG_MARK(5);
    // o_i = sum_j softmax_j(q_i k_j / sqrt r) v_j
    const float sc = rsqrtf((float)r);
    for (int i = tid; i < nwl * r; i += 256) {
        const int w = byR.div(i), m = i - w * r;
        const float* kv = s.sm + w * PG_SMW + PG_KV;
        const float qs = s.sm[w * PG_SMW + PG_Q + m] * sc;
        float mx = -3.0e38f;
        for (int j = 0; j < r; ++j) mx = fmaxf(mx, qs * kv[j]);
        float den = 0.f, num = 0.f;
        for (int j = 0; j < r; ++j) {
            const float e = expf(qs * kv[j] - mx);
            if (KEEP_AT) s.At[w * s.LDA + m * RP + j] = e;
            den += e;
            num += e * kv[r + j];
        }
        if (KEEP_AT)
            for (int j = 0; j < r; ++j) s.At[w * s.LDA + m * RP + j] /= den;
        s.sm[w * PG_SMW + PG_O + m] = num / den;
    }
    __syncthreads();
    PG_MARK(6);
    // o2 = Wproj o + b
    for (int i = tid; i < nwl * r; i += 256) {
        const int w = byR.div(i), m = i - w * r;
        const float* o = s.sm + w * PG_SMW + PG_O;
        float acc = s.bpp[m];
        for (int j = 0; j < r; ++j) acc += s.Wpp[m * RP + j] * o[j];
        s.sm[w * PG_SMW + PG_O2 + m] = acc;
    }
    __syncthreads();
}

template <int R>
__global__ __launch_bounds__(256) void pg_gate_fwd_kernel(PgDev a) {
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    const int C = a.C, r = R > 0 ? R : a.r, pgw = r, tid = threadIdx.x, nwl = a.wpg, win0 = blockIdx.x * nwl;
    const PgLds s = pg_lds(reinterpret_cast<float*>(smem_v), C, r, false);
    pg_forward_chain<false, R>(a, s, win0);
    PG_MARK(7);
    // g = Wup o2: one thread per (window, channel), coalesced along c
    const FastDiv byC(C);
    if (R > 0 && R % 4 == 0 && 256 % C == 0) {
        // the thread's channel is the same in every round (c = tid % C; C divides 256 -- at C > 256 item tid + 256 k is channel
        // (tid + 256 k) % C, a different one per round: the generic loop below): its Wup row stays in registers, o2 of the round's window
        // comes as 16-byte broadcast reads (the rows of `sm` are 16-byte aligned: 15 r + 4 floats with r % 4 == 0)
        constexpr int RR = R > 0 ? R : 4;
        const int c = tid % C;
        float wrow[RR];
#pragma unroll
        for (int j = 0; j < RR; ++j) wrow[j] = s.Wup[c * s.RP + j];
        for (int i = tid; i < nwl * C; i += 256) {
            const int w = byC.div(i);
            if (win0 + w >= a.nW) continue;
            const float* o2 = s.sm + w * PG_SMW + PG_O2;
            float acc = 0.f;
#pragma unroll
            for (int j4 = 0; j4 < RR / 4; ++j4) {
                const f32x4 ov = *reinterpret_cast<const f32x4*>(o2 + 4 * j4);
                for (int e = 0; e < 4; ++e) acc += wrow[4 * j4 + e] * ov[e];
            }
            a.gate[(long)(win0 + w) * C + c] = acc;
        }
    } else {
        for (int i = tid; i < nwl * C; i += 256) {
            const int w = byC.div(i), c = i - w * C;
            if (win0 + w >= a.nW) continue;
            const float* o2 = s.sm + w * PG_SMW + PG_O2;
            float acc = 0.f;
            for (int j = 0; j < r; ++j) acc += s.Wup[c * s.RP + j] * o2[j];
            a.gate[(long)(win0 + w) * C + c] = acc;
        }
    }
    PG_MARK(8);
}

template <int R>
__global__ __launch_bounds__(256) void pg_gate_bwd_kernel(PgDev a) {
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    const int C = a.C, r = R > 0 ? R : a.r, pgw = r, tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform(), nwl = a.wpg, win0 = blockIdx.x * nwl;
    const PgLds s = pg_lds(reinterpret_cast<float*>(smem_v), C, r, true, a.stage_wdn != 0);
    // Wprompt column fragments of this wave's first two d-mu tiles (used at the very end): requested now, they cost no
    // round trip later
    const int nct = C / 16;
    const int colA[1] = {wv * 16}, colB[1] = {(wv + 4) * 16};
    f32x4 wfa[8][1], wfb[8][1];
    if (wv < nct) pg_load_chunk<1, true>(wfa, a.Wprompt, C, C, 128, colA, 0, 128);
    if (wv + 4 < nct) pg_load_chunk<1, true>(wfb, a.Wprompt, C, C, 128, colB, 0, 128);
    const FastDiv byC(C), byR(r);
#pragma unroll 4
    for (int i = tid; i < nwl * C; i += 256) {
        const int w = byC.div(i), c = i - w * C;
        s.dg[w * s.LDC + c] = win0 + w < a.nW ? a.dgate[(long)(win0 + w) * C + c] : 0.f;
    }
    pg_forward_chain<true, R>(a, s, win0);           // its first barrier also covers the dg tile
    PG_MARK(7);
    const float sc = rsqrtf((float)r);
    // do2 = Wup^T dg  (r rows x 16 windows, K = C): MFMA with the weight read column-wise
    if (wv * 16 < r) {                            // linear_up is in LDS as [C][r]: its column fragments are transposed reads
        f32x4 acc = {0, 0, 0, 0};
        for (int kc = 0; kc < C; kc += 16)
            mma(acc, load_frag_tr<float>(s.Wup, s.RP, wv * 16, kc), load_frag<float>(s.dg, s.LDC, 0, kc));
        const int w = lane & 15, rr = wv * 16 + (lane >> 4) * 4;
        for (int j = 0; j < 4; ++j)
            if (rr + j < r) s.sm[w * PG_SMW + PG_DO2 + rr + j] = acc[j];
    }
    __syncthreads();
    PG_MARK(8);
    for (int i = tid; i < nwl * r; i += 256) {          // do = Wproj^T do2
        const int w = byR.div(i), m = i - w * r;
        const float* do2 = s.sm + w * PG_SMW + PG_DO2;
        float acc = 0.f;
        for (int j = 0; j < r; ++j) acc += s.Wpp[j * s.RP + m] * do2[j];
        s.sm[w * PG_SMW + PG_DO + m] = acc;
    }
    __syncthreads();
    PG_MARK(9);
    // row i: rs_i = sum_j A_ij do_i v_j;  dS_ij = A_ij (do_i v_j - rs_i);  dq_i = sc sum_j dS_ij k_j
    for (int i = tid; i < nwl * r; i += 256) {
        const int w = byR.div(i), m = i - w * r;
        const float* A = s.At + w * s.LDA + m * s.RP;
        const float* kv = s.sm + w * PG_SMW + PG_KV;
        const float dov = s.sm[w * PG_SMW + PG_DO + m];
        float rs = 0.f;
        for (int j = 0; j < r; ++j) rs += A[j] * dov * kv[r + j];
        float acc = 0.f;
        for (int j = 0; j < r; ++j) acc += A[j] * (dov * kv[r + j] - rs) * kv[j];
        s.sm[w * PG_SMW + PG_DQ + m] = acc * sc;
        s.sm[w * PG_SMW + PG_RS + m] = rs;
    }
    __syncthreads();
    PG_MARK(10);
    // column j: dk_j = sc sum_i dS_ij q_i ; dv_j = sum_i A_ij do_i   (dS re-formed from A and the row sums rs_i)
    for (int i = tid; i < nwl * r; i += 256) {
        const int w = byR.div(i), j = i - w * r;
        const float* sm = s.sm + w * PG_SMW;
        const float vj = sm[PG_KV + r + j];
        float dk = 0.f, dv = 0.f;
        for (int m = 0; m < r; ++m) {
            const float A = s.At[w * s.LDA + m * s.RP + j], dov = sm[PG_DO + m];
            dk += A * (dov * vj - sm[PG_RS + m]) * sm[PG_Q + m];
            dv += A * dov;
        }
        s.sm[w * PG_SMW + PG_DKV + j] = dk * sc;
        s.sm[w * PG_SMW + PG_DKV + r + j] = dv;
    }
    __syncthreads();
    PG_MARK(11);
    for (int i = tid; i < nwl * r; i += 256) {          // dd = Wkv^T dkv ; ds = Wq^T dq
        const int w = byR.div(i), j = i - w * r;
        const float* dkv = s.sm + w * PG_SMW + PG_DKV;
        const float* dq = s.sm + w * PG_SMW + PG_DQ;
        float acc = 0.f, acc2 = 0.f;
        for (int m = 0; m < 2 * r; ++m) acc += s.Wkv[m * s.RP + j] * dkv[m];
        for (int m = 0; m < r; ++m) acc2 += s.Wq[m * s.RP + j] * dq[m];
        s.sm[w * PG_SMW + PG_DD + j] = acc;
        s.sm[w * PG_SMW + PG_DS + j] = acc2;
    }
    __syncthreads();
    PG_MARK(12);
    for (int i = tid; i < nwl * 128; i += 256) {        // dw[p] = P[p] . ds
        const int w = i >> 7, p = i & 127;
        const float* ds = s.sm + w * PG_SMW + PG_DS;
        float acc = 0.f;
        for (int j = 0; j < r; ++j) acc += s.P[p * s.RP + j] * ds[j];
        s.dl[w * s.LDW + p] = acc;
    }
    __syncthreads();
    PG_MARK(13);
    for (int ww = 0; ww < 4; ++ww) {                        // dlogit = w (dw - sum w dw)
        const int w = wv * 4 + ww;
        if (w >= nwl) break;                                // wave-uniform
        const float w0 = s.w[w * s.LDW + lane], w1 = s.w[w * s.LDW + lane + 64];
        const float d0 = s.dl[w * s.LDW + lane], d1 = s.dl[w * s.LDW + lane + 64];
        const float tot = wave_sum(w0 * d0 + w1 * d1);
        s.dl[w * s.LDW + lane] = w0 * (d0 - tot);
        s.dl[w * s.LDW + lane + 64] = w1 * (d1 - tot);
    }
    __syncthreads();
    PG_MARK(14);
    // dmu = Wprompt^T dlogit + Wdown^T dd:  C rows x 16 windows, K = 128 by MFMA, the rank-r term in the epilogue
    for (int ct = wv; ct < nct; ct += 4) {
        f32x4 acc[1] = {{0, 0, 0, 0}};
        const int col0[1] = {ct * 16};
        if (ct == wv) pg_mma_chunk<1>(acc, wfa, s.dl, s.LDW, 0, 128);              // prefetched at the top of the kernel
        else if (ct == wv + 4) pg_mma_chunk<1>(acc, wfb, s.dl, s.LDW, 0, 128);
        else {
            f32x4 wf[8][1];
            pg_load_chunk<1, true>(wf, a.Wprompt, C, C, 128, col0, 0, 128);
            pg_mma_chunk<1>(acc, wf, s.dl, s.LDW, 0, 128);
        }
        const int w = lane & 15, c0 = ct * 16 + (lane >> 4) * 4;
        const float* dd = s.sm + w * PG_SMW + PG_DD;
        const float* wdn = a.stage_wdn ? s.Wdn : a.Wdown;
        for (int m = 0; m < r; ++m) {
            const f32x4 wd = *reinterpret_cast<const f32x4*>(wdn + (long)m * C + c0);
            for (int j = 0; j < 4; ++j) acc[0][j] += wd[j] * dd[m];
        }
        if (w < nwl && win0 + w < a.nW) *reinterpret_cast<f32x4*>(a.dmu + (long)(win0 + w) * C + c0) = acc[0];
    }
    PG_MARK(15);
    // ---- factor rows: L = [dg(C) | do2(r) | dkv(2r) | dq(r) | w(128) | dlogit(128) | dd(r) | 0..],
    //                   R = [o2(r) | o(r) | 1 | d(r) | s(r) | ds(r) | mu(C) | 0..]
    // wave wv writes windows wv, wv+4, ...; one short loop per segment (no per-element ladder: the divergent ladder with an
    // LDS read in every arm cost 16 us per launch)
    auto put = [&](void* base, long idx, float v) __attribute__((always_inline)) {
        if (a.lr_bf16 == MPHSIR_BF16) reinterpret_cast<bf16_t*>(base)[idx] = (bf16_t)v;
        else if (a.lr_bf16 == MPHSIR_F16) reinterpret_cast<f16_t*>(base)[idx] = (f16_t)v;
        else reinterpret_cast<float*>(base)[idx] = v;
    };
    for (int w = wv; w < nwl && win0 + w < a.nW; w += 4) {
        const float* sm = s.sm + w * PG_SMW;
        const long L0 = (long)(win0 + w) * a.KL, R0 = (long)(win0 + w) * a.KR;
        for (int c = lane; c < C; c += 64) { put(a.L, L0 + c, s.dg[w * s.LDC + c]); put(a.R, R0 + 5 * r + 1 + c, s.mu[w * s.LDC + c]); }
        for (int c = lane; c < 128; c += 64) { put(a.L, L0 + C + 4 * r + c, s.w[w * s.LDW + c]); put(a.L, L0 + C + 4 * r + 128 + c, s.dl[w * s.LDW + c]); }
        if (lane < r) {
            put(a.L, L0 + C + lane, sm[PG_DO2 + lane]);
            put(a.L, L0 + C + 3 * r + lane, sm[PG_DQ + lane]);
            put(a.L, L0 + C + 4 * r + 256 + lane, sm[PG_DD + lane]);
            put(a.R, R0 + lane, sm[PG_O2 + lane]);
            put(a.R, R0 + r + lane, sm[PG_O + lane]);
            put(a.R, R0 + 2 * r + 1 + lane, sm[PG_D + lane]);
            put(a.R, R0 + 3 * r + 1 + lane, sm[PG_S + lane]);
            put(a.R, R0 + 4 * r + 1 + lane, sm[PG_DS + lane]);
        }
        if (lane < 2 * r) put(a.L, L0 + C + r + lane, sm[PG_DKV + lane]);
        if (lane == 0) put(a.R, R0 + 2 * r, 1.f);
        for (int c = C + 5 * r + 256 + lane; c < a.KL; c += 64) put(a.L, L0 + c, 0.f);      // zero padding of the row tails
        for (int c = 5 * r + 1 + C + lane; c < a.KR; c += 64) put(a.R, R0 + c, 0.f);
    }
    PG_MARK(16);
}

}  // namespace mphsir

namespace mphsir {
// windows per workgroup: 16 fill the MFMA N axis, but the chain is latency-bound and every phase loops over (windows x index)
// items, so a launch that cannot fill the chip anyway takes fewer windows per workgroup -- as many workgroups as fit one round
// (measured, tools/bench_pg.py: 2048 windows 15.3 -> 13.5 us with 8; 512 windows 13.1 -> 10.1 and 128 windows 19.1 -> 14.1 with 2;
// 4096 windows stay at 16: with 8 they need two rounds)
static int pg_windows_per_wg(int nW) {
    int w = PG_NWIN;
    while (w > 2 && (nW + w / 2 - 1) / (w / 2) <= 256) w /= 2;
    return w;
}
void pg_debug_buffer(unsigned long long* p) { g_pg_dbg = p; }
}      // mphsir_debug(MPHSIR_DEBUG_PG_GATE, ...)

extern "C" int mphsir_pg_gate_fwd(const mphsir_pg_fwd_args* a, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "pg_gate_fwd");
    MPHSIR_REQUIRE(a && a->mu && a->Wprompt && a->prompt_param && a->Wq && a->Wkv && a->Wdown && a->Wpproj && a->bpproj && a->Wup && a->gate,
                   "pg_gate_fwd: null pointer");
    MPHSIR_REQUIRE(a->nW > 0 && a->C > 0 && a->C % 16 == 0 && a->r > 0 && a->r <= PG_RMAX, "pg_gate_fwd: need C %% 16 == 0 and 0 < r <= 32");
    PgDev d{a->mu, nullptr, a->Wprompt, a->prompt_param, a->Wq, a->Wkv, a->Wdown, a->Wpproj, a->bpproj, a->Wup, a->gate, nullptr, nullptr,
            nullptr, a->nW, a->C, a->r, 0, 0, 0, 0, pg_windows_per_wg(a->nW), g_pg_dbg};
    const size_t shmem = pg_lds_bytes(a->C, a->r, false);
    MPHSIR_REQUIRE(shmem <= 160 * 1024, "pg_gate_fwd: (C=%d, r=%d) needs %d bytes of LDS", a->C, a->r, (int)shmem);
    const dim3 grid((a->nW + d.wpg - 1) / d.wpg);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MPHSIR_PG_FWD(R_)                                                                        \
    do {                                                                                         \
        allow_big_lds(pg_gate_fwd_kernel<R_>, shmem);                                            \
        MPHSIR_LAUNCH(MPHSIR_K_PG_GATE, pg_gate_fwd_kernel<R_>, grid, dim3(256), shmem, s, d);    \
        return MPHSIR_OK;                                                                        \
    } while (0)
    if (a->r == 8) MPHSIR_PG_FWD(8);
    if (a->r == 12) MPHSIR_PG_FWD(12);
    if (a->r == 16) MPHSIR_PG_FWD(16);
    if (a->r == 24) MPHSIR_PG_FWD(24);        // remote-sensing dec1 / refinement: C = 192, compress_ratio 8
    MPHSIR_PG_FWD(0);
#undef MPHSIR_PG_FWD
}

extern "C" int mphsir_pg_gate_bwd(const mphsir_pg_bwd_args* a, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "pg_gate_bwd");
    MPHSIR_REQUIRE(a && a->mu && a->dgate && a->Wprompt && a->prompt_param && a->Wq && a->Wkv && a->Wdown && a->Wpproj && a->bpproj &&
                       a->Wup && a->dmu && a->L && a->R, "pg_gate_bwd: null pointer");
    MPHSIR_REQUIRE(a->nW > 0 && a->C > 0 && a->C % 16 == 0 && a->r > 0 && a->r <= PG_RMAX, "pg_gate_bwd: need C %% 16 == 0 and 0 < r <= 32");
    MPHSIR_REQUIRE(a->KL >= a->C + 5 * a->r + 256 && a->KR >= 5 * a->r + 1 + a->C, "pg_gate_bwd: factor widths too small");
    MPHSIR_REQUIRE(!a->lr_bf16 || (a->KL % 8 == 0 && a->KR % 8 == 0), "pg_gate_bwd: bf16 factor rows need KL, KR multiples of 8");
    PgDev d{a->mu, a->dgate, a->Wprompt, a->prompt_param, a->Wq, a->Wkv, a->Wdown, a->Wpproj, a->bpproj, a->Wup, nullptr, a->dmu, a->L, a->R,
            a->nW, a->C, a->r, a->KL, a->KR, a->lr_bf16, 0, pg_windows_per_wg(a->nW), g_pg_dbg};
    d.stage_wdn = pg_lds_bytes(a->C, a->r, true, true) <= 160 * 1024 ? 1 : 0;
    const size_t shmem = pg_lds_bytes(a->C, a->r, true, d.stage_wdn != 0);
    MPHSIR_REQUIRE(shmem <= 160 * 1024, "pg_gate_bwd: (C=%d, r=%d) needs %d bytes of LDS", a->C, a->r, (int)shmem);
    const dim3 grid((a->nW + d.wpg - 1) / d.wpg);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MPHSIR_PG_BWD(R_)                                                                            \
    do {                                                                                             \
        allow_big_lds(pg_gate_bwd_kernel<R_>, shmem);                                                \
        MPHSIR_LAUNCH(MPHSIR_K_PG_GATE_BWD, pg_gate_bwd_kernel<R_>, grid, dim3(256), shmem, s, d);    \
        return MPHSIR_OK;                                                                            \
    } while (0)
    if (a->r == 8) MPHSIR_PG_BWD(8);
    if (a->r == 12) MPHSIR_PG_BWD(12);
    if (a->r == 16) MPHSIR_PG_BWD(16);
    if (a->r == 24) MPHSIR_PG_BWD(24);
    MPHSIR_PG_BWD(0);
#undef MPHSIR_PG_BWD
}
