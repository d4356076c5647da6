// gated_mlp_wgrad: the four parameter gradients of  y = x + keep * fc2(value * gelu(gate)),  [value|gate] = fc1(LN(x)),
// WITHOUT the token-sized intermediates ever reaching HBM.
//
// Backward of GatedMlp (reference forward: net/MP_HSIR.py:66-82, called at :719; autograd via train.py:58-67).  Until round 5
// gated_mlp_bwd wrote h = value*gelu(gate) [M][HP], [dval|dgate] [M][2HP] and LN(x) only so that mphsir_gemm_tn could read them
// back: 146 MB per launch at C = 128, M = 131072, 342 MB read again by the token-reduction GEMMs.  The weight gradient of a
// 32-wide slice of the hidden dimension needs nothing but the token tiles LN(x), dm = keep*dy and that slice of the weights:
//
//   grid = R token ranges x S hidden slabs (NCH chunks of 32 hidden units each).  A workgroup keeps its slab's fc1 rows / W2^T rows
//   in REGISTERS (MFMA fragments) and its slice of dW1 / dW2 / db1 (/ db2) as fp32 MFMA accumulators for its whole token range, walks the range in
//   64-token tiles and per tile
//     (a) recomputes value, gate (fc1 on the LN-ed tile) and dh = dm W2 for its slab           [MFMA, K = C],
//     (b) h, dval = dh*gelu(gate), dgate = dh*value*gelu'(gate) -> 16-bit staging tile in LDS  [VALU],
//     (c) dW2[:, slab] += dm^T h,  dW1[slab, :] += [dval|dgate]^T LN(x),  db1 += colsum        [MFMA, K = tokens],
//   and writes ONE fp32 partial per range at the end; mphsir_reduce_parts sums the R partials in order (deterministic).
//   The K = token operands of (c) are read token-major from LDS through ds_read_b64_tr_b16 with a token permutation that
//   makes the reads conflict free (load_frag_trp below); bias gradients are MFMA products with a fragment of ones.
//
// Recomputing (a) costs 6 C HP FLOPs per token on matrix cores that the step leaves 94 % idle; what it saves is
// (3 HP + 2 C) * 2 bytes written and read again per token.  The S workgroups of a range run on ONE XCD (block index mapping
// below) so that the token tiles they share come out of that XCD's L2.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct MlpWgDev {
    const void* XN; const void* DM;                    // [M][C]: LN(x), keep*dy
    const void* W1; const float* b1; const void* W2T;  // [2*HP][C], [2*HP], [HP][C]
    float* dW1p; float* dW2p; float* db1p; float* db2p;  // [R][2*HP][C], [R][C][HP], [R][2*HP], [R][C]
    int M, HP, R, S, tiles_per;
};

// Fragment of a K-STRIDED operand held token-major as [tok][col] in LDS, like load_frag_tr, but with the K slots of the
// 32-token step dealt to tile rows so that the 32 lanes one ds_read_b64_tr_b16 services together touch 8 CONSECUTIVE rows:
// lane group g (K slots 8g..8g+7) reads rows 4g+q (first instruction) and 16+4g+q (second).  With a row pitch of 8*odd dwords
// (every tile here: widths are multiples of 32 elements + 16 elements of padding) 8 consecutive rows x 32 bytes cover all 64
// banks once; load_frag_tr's rows {0-3, 8-11} collide pairwise at such a pitch.  A contraction only needs both operands to
// use the same slot -> token map, which they do by using this one loader.
template <class T>
__device__ __forceinline__ typename ElemTraits<T>::frag_t load_frag_trp(const T* base, int ld, int col0, int k0) {
    const int l = lane_id(), g = l >> 4, q = (l & 15) >> 2, p = l & 3;
    const T* a = base + (size_t)(k0 + 4 * g + q) * ld + col0 + 4 * p;
    const auto lo = lds_read_tr16(a), hi = lds_read_tr16(a + 16 * ld);
    return typename ElemTraits<T>::frag_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <class T, int C, int NCH>
__global__ __launch_bounds__(256 * NCH, 2) void gated_mlp_wgrad_kernel(MlpWgDev a) {
    typedef ElemTraits<T> TR;
    typedef typename TR::frag_t frag_t;
    static_assert(sizeof(T) == 2, "16-bit storage types only");
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T), LDX = C + PAD, LDST = 96 + PAD, VEC = Vec16<T>::N, VPR = C / VEC;
    constexpr int NTHR = 256 * NCH, NKC = C / TR::KCHUNK, NCB = C / 16, NOWN = (NCB + 3) / 4;
    constexpr int NTV = 2 * 64 * VPR, NPT = (NTV + NTHR - 1) / NTHR;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* XNs = reinterpret_cast<T*>(smem_v);              // [64][LDX]  LN(x) tile
    T* DMs = XNs + 64 * LDX;                            // [64][LDX]  dm tile
    T* STb = DMs + 64 * LDX;                            // [NCH][64][LDST] per chunk: h | dval | dgate of the tile's tokens

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int tb = wv & 3, ch = wv >> 2;
    // the S slabs of a token range on one XCD (blocks b and b + 8 share an XCD): they read the same tiles
    const int xcd = blockIdx.x & 7, kk = blockIdx.x >> 3;
    const int rng = (kk / a.S) * 8 + xcd, slab = kk % a.S;
    const int HP = a.HP, j0 = (slab * NCH + ch) * 32;
    const bool active = j0 < HP;                        // a trailing slab may own fewer than NCH chunks
    const T* XN = reinterpret_cast<const T*>(a.XN);
    const T* DM = reinterpret_cast<const T*>(a.DM);
    const long ntiles = a.M / 64;
    long t0 = (long)rng * a.tiles_per, t1 = t0 + a.tiles_per;
    if (t0 > ntiles) t0 = ntiles;
    if (t1 > ntiles) t1 = ntiles;

    // ---- the slab's weights: once, straight into REGISTERS.  In the recompute phase a wave owns one 16-wide half of its chunk (fh) for
    // 32 of the tile's tokens (token blocks 2 th, 2 th + 1): its three weight row blocks -- fc1 value rows, fc1 gate rows, W2^T rows
    // -- are 3 NKC MFMA A fragments that never change, so the phase reads nothing but the token fragments from LDS.
    const int fh = tb & 1, th = tb >> 1;
    T* ST = STb + ch * 64 * LDST;
    const int hr = (lane >> 4) * 4, tk = lane & 15;
    frag_t wA[3][NKC];
    float bv[4], bg[4];
    {
        const T* W1 = reinterpret_cast<const T*>(a.W1);
        const T* W2T = reinterpret_cast<const T*>(a.W2T);
        const int jr = active ? j0 + 16 * fh : 0;           // an inactive wave (trailing slab) loads rows it never uses
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            wA[0][kc] = load_frag<T>(W1, C, jr, kc * TR::KCHUNK);
            wA[1][kc] = load_frag<T>(W1, C, HP + jr, kc * TR::KCHUNK);
            wA[2][kc] = load_frag<T>(W2T, C, jr, kc * TR::KCHUNK);
        }
        for (int r = 0; r < 4; ++r) {
            bv[r] = a.b1[jr + hr + r];
            bg[r] = a.b1[HP + jr + hr + r];
        }
    }

    // ---- token tile: global -> registers (a tile ahead) -> LDS ------------------------------------------------------
    Vec16<T> preg[NPT];
    auto tload = [&](long t) {
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const int v = tid + NTHR * it;
            if (NTV % NTHR == 0 || v < NTV) {
                const int which = v / (64 * VPR), r = (v / VPR) % 64, c = (v % VPR) * VEC;
                preg[it] = load16<T>((which ? DM : XN) + (t * 64 + r) * C + c);
            }
        }
    };
    // (forming dm = keep * dy here instead of reading the dm the data-gradient kernel writes was measured: 100 more VALU instructions
    // per tile and thread, the kernel 2.1 -> 2.8 ms per training step and the step 0.8 ms longer)
    auto tstore = [&]() {
#pragma unroll
        for (int it = 0; it < NPT; ++it) {
            const int v = tid + NTHR * it;
            if (NTV % NTHR == 0 || v < NTV) {
                const int which = v / (64 * VPR), r = (v / VPR) % 64, c = (v % VPR) * VEC;
                store16<T>((which ? DMs : XNs) + r * LDX + c, preg[it]);
            }
        }
    };

    f32x4 accW2[NOWN][2], accV[NOWN][2], accG[NOWN][2], accB1, accB2[NOWN];
#pragma unroll
    for (int i = 0; i < NOWN; ++i) {
        accB2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int f = 0; f < 2; ++f) accW2[i][f] = accV[i][f] = accG[i][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    accB1 = f32x4{0.f, 0.f, 0.f, 0.f};
    frag_t ones;
#pragma unroll
    for (int e = 0; e < TR::EPL; ++e) ones[e] = from_f32<T>(1.0f);
    const bool do_b2 = slab == 0 && ch == 0;

    constexpr bool PRE = C <= 128;                       // C = 192: 72 registers of weight fragments + 84 accumulators leave no room for a tile in flight
    if (PRE && t0 < t1) tload(t0);
    for (long t = t0; t < t1; ++t) {
        if (!PRE) tload(t);
        tstore();
        __syncthreads();                                 // tile visible
        if (PRE && t + 1 < t1) tload(t + 1);             // in flight during the phases below
        if (active) {
            // (a) value, gate, dh of this wave's 16 hidden units x 32 tokens.  The token fragments of K chunk kc + 1 are requested before
            // the MFMAs of chunk kc, and the scheduler is kept from sinking each read down to its use (left alone it serialises
            // read -> wait -> MFMA through one register quad: the whole LDS latency exposed every time at two waves per SIMD).
            frag_t tf[2][4];                          // [buffer][bx of block 0, bx of block 1, bd of block 0, bd of block 1] of one K chunk
            auto fetch = [&](int kc) {
                const int k0 = kc * TR::KCHUNK, b = kc & 1;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    tf[b][u] = load_frag<T>(XNs, LDX, (2 * th + u) * 16, k0);
                    tf[b][2 + u] = load_frag<T>(DMs, LDX, (2 * th + u) * 16, k0);
                }
            };
            f32x4 pv[2], pg[2], pe[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) pv[u] = pg[u] = pe[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            fetch(0);
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const int b = kc & 1;
                if (kc + 1 < NKC) fetch(kc + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    mma(pv[u], wA[0][kc], tf[b][u]);
                    mma(pg[u], wA[1][kc], tf[b][u]);
                    mma(pe[u], wA[2][kc], tf[b][2 + u]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // (b) h | dval | dgate -> staging rows of these tokens (lane: 4 consecutive hidden units of one token)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                f32x4 hh, dv, dg;
                for (int r = 0; r < 4; ++r) {
                    const float va = pv[u][r] + bv[r], ga = pg[u][r] + bg[r];
                    float ea, da;
                    Math<T>::gelu_pair(ga, ea, da);
                    hh[r] = va * ea;
                    dv[r] = pe[u][r] * ea;
                    dg[r] = pe[u][r] * va * da;
                }
                T* row = ST + ((2 * th + u) * 16 + tk) * LDST + 16 * fh + hr;
                store4<T>(row, hh);
                store4<T>(row + 32, dv);
                store4<T>(row + 64, dg);
            }
        }
        __syncthreads();                                 // staging tile complete
        if (active) {
            // (c) K = the tile's 64 tokens, two steps of 32
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int k0 = 32 * ks;
                frag_t hB[2], dvA[2], dgA[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    hB[f] = load_frag_trp<T>(ST, LDST, 16 * f, k0);
                    dvA[f] = load_frag_trp<T>(ST, LDST, 32 + 16 * f, k0);
                    dgA[f] = load_frag_trp<T>(ST, LDST, 64 + 16 * f, k0);
                }
                mma(accB1, tb == 0 ? dvA[0] : (tb == 1 ? dvA[1] : (tb == 2 ? dgA[0] : dgA[1])), ones);      // db1: 16 entries per wave
#pragma unroll
                for (int i = 0; i < NOWN; ++i) {
                    const int cb = tb + 4 * i;
                    if (NCB % 4 == 0 || cb < NCB) {
                        const frag_t dmA = load_frag_trp<T>(DMs, LDX, 16 * cb, k0), xnB = load_frag_trp<T>(XNs, LDX, 16 * cb, k0);
#pragma unroll
                        for (int f = 0; f < 2; ++f) {
                            mma(accW2[i][f], dmA, hB[f]);          // D[c][hidden]
                            mma(accV[i][f], dvA[f], xnB);          // D[hidden][c]
                            mma(accG[i][f], dgA[f], xnB);
                        }
                        if (do_b2) mma(accB2[i], dmA, ones);       // db2 = colsum dm: once per range
                    }
                }
            }
        }
        __syncthreads();                                 // every wave is done with the tiles before the next one lands
    }

    // ---- this range's partial sums ----------------------------------------------------------------------------------
    if (!active) return;
    float* dW1p = a.dW1p + (long)rng * 2 * HP * C;
    float* dW2p = a.dW2p + (long)rng * C * HP;
#pragma unroll
    for (int i = 0; i < NOWN; ++i) {
        const int cb = tb + 4 * i;
        if (NCB % 4 == 0 || cb < NCB) {
#pragma unroll
            for (int f = 0; f < 2; ++f)
                for (int r = 0; r < 4; ++r) {
                    dW2p[(long)(16 * cb + hr + r) * HP + j0 + 16 * f + tk] = accW2[i][f][r];
                    dW1p[(long)(j0 + 16 * f + hr + r) * C + 16 * cb + tk] = accV[i][f][r];
                    dW1p[(long)(HP + j0 + 16 * f + hr + r) * C + 16 * cb + tk] = accG[i][f][r];
                }
            if (do_b2 && tk == 0)
                for (int r = 0; r < 4; ++r) a.db2p[(long)rng * C + 16 * cb + hr + r] = accB2[i][r];
        }
    }
    if (tk == 0) {
        const int jb = (tb < 2 ? j0 + 16 * tb : HP + j0 + 16 * (tb - 2)) + hr;
        for (int r = 0; r < 4; ++r) a.db1p[(long)rng * 2 * HP + jb + r] = accB1[r];
    }
}

template <class T, int C, int NCH>
constexpr size_t mlp_wgrad_lds() {
    constexpr size_t PAD = LDS_PAD_BYTES / sizeof(T), LDX = C + PAD, LDST = 96 + PAD;
    return (2 * 64 * LDX + NCH * 64 * LDST) * sizeof(T);
}

template <class T, int C, int NCH>
static int launch_mlp_wgrad(const MlpWgDev& d, hipStream_t s) {
    if constexpr (sizeof(T) == 2 && mlp_wgrad_lds<T, C, NCH>() <= 160 * 1024) {
        constexpr size_t lds = mlp_wgrad_lds<T, C, NCH>();
        allow_big_lds(gated_mlp_wgrad_kernel<T, C, NCH>, lds);
        MPHSIR_LAUNCH(MPHSIR_K_GATED_MLP_WGRAD, (gated_mlp_wgrad_kernel<T, C, NCH>), dim3(d.R * d.S), dim3(256 * NCH), lds, s, d);
        return MPHSIR_OK;
    }
    set_error("gated_mlp_wgrad: C=%d with %d chunks per workgroup does not fit this element type", C, NCH);
    return MPHSIR_EINVAL;
}

template <class T>
static int dispatch_mlp_wgrad(const MlpWgDev& d, int C, int nch, hipStream_t s) {
    switch (C * 4 + nch) {
        case 32 * 4 + 1: return launch_mlp_wgrad<T, 32, 1>(d, s);
        case 64 * 4 + 1: return launch_mlp_wgrad<T, 64, 1>(d, s);
        case 64 * 4 + 2: return launch_mlp_wgrad<T, 64, 2>(d, s);
        case 96 * 4 + 1: return launch_mlp_wgrad<T, 96, 1>(d, s);
        case 96 * 4 + 2: return launch_mlp_wgrad<T, 96, 2>(d, s);
        case 128 * 4 + 1: return launch_mlp_wgrad<T, 128, 1>(d, s);
        case 128 * 4 + 2: return launch_mlp_wgrad<T, 128, 2>(d, s);
        case 192 * 4 + 1: return launch_mlp_wgrad<T, 192, 1>(d, s);
    }
    set_error("gated_mlp_wgrad: C=%d, chunks_per_wg=%d not instantiated", C, nch);
    return MPHSIR_EINVAL;
}

}  // namespace mphsir

extern "C" int mphsir_gated_mlp_wgrad_fits(int32_t C, int32_t chunks_per_wg, int dtype) {
    if (dtype != MPHSIR_BF16 && dtype != MPHSIR_F16) return 0;
    if (chunks_per_wg == 1) return C == 32 || C == 64 || C == 96 || C == 128 || C == 192;      // C = 256: 68 registers spilled at two waves per SIMD
    if (chunks_per_wg == 2) return C == 64 || C == 96 || C == 128;
    return 0;
}

extern "C" int mphsir_gated_mlp_wgrad(const mphsir_mlp_wgrad_args* a, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_CHECK_ARGS(a, "gated_mlp_wgrad");
    MPHSIR_REQUIRE(a && a->XN && a->DM && a->W1 && a->b1 && a->W2T && a->dW1p && a->dW2p && a->db1p && a->db2p, "gated_mlp_wgrad: null pointer");
    MPHSIR_REQUIRE(dtype == MPHSIR_BF16 || dtype == MPHSIR_F16, "gated_mlp_wgrad: 16-bit element types only (dtype %d)", dtype);
    MPHSIR_REQUIRE(a->M > 0 && a->M % 64 == 0 && a->HP > 0 && a->HP % 32 == 0, "gated_mlp_wgrad: M %% 64 and HP %% 32 must be 0");
    MPHSIR_REQUIRE(a->ranges > 0 && a->ranges % 8 == 0, "gated_mlp_wgrad: ranges must be a positive multiple of 8 (one XCD per range)");
    MPHSIR_REQUIRE(a->chunks_per_wg == 1 || a->chunks_per_wg == 2, "gated_mlp_wgrad: chunks_per_wg must be 1 or 2");
    MPHSIR_REQUIRE(aligned16(a->XN) && aligned16(a->DM) && aligned16(a->W1) && aligned16(a->W2T), "gated_mlp_wgrad: 16-byte alignment required");
    const int nchunks = a->HP / 32, S = (nchunks + a->chunks_per_wg - 1) / a->chunks_per_wg;
    const long ntiles = a->M / 64;
    MlpWgDev d{a->XN, a->DM, a->W1, a->b1, a->W2T, a->dW1p, a->dW2p, a->db1p, a->db2p, (int)a->M, a->HP, a->ranges, S,
               (int)((ntiles + a->ranges - 1) / a->ranges)};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MPHSIR_BF16) return dispatch_mlp_wgrad<bf16_t>(d, a->C, a->chunks_per_wg, s);
    return dispatch_mlp_wgrad<f16_t>(d, a->C, a->chunks_per_wg, s);
}
