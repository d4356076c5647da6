// gemm_tn: token-reduction GEMM  C[n1][n2] = sum_m A[m][n1] * B[m][n2]   (A, B token-major, C fp32)
//
// Every parameter gradient of a Linear / 1x1 conv on the path is of this form (dW = dY^T X over all
// tokens of the local batch), and so is the per-sample dM = d_out^T v of the folded
// channel attention.  M (tokens) is 1e5..1e6 while N1, N2 <= ~1.4k, so the work is split over M:
// grid = (output tiles of 64x64, nsplit, batch); each workgroup walks its token range in 64-row steps,
// transposing both operand tiles into LDS (the MFMA K axis must be lane-contiguous) and accumulating
// its 64x64 tile in registers; partial tiles go to Cpart[batch][split][N1][N2] and are summed by the
// caller in split order: deterministic, no atomics.  HBM-bound (AI = 32..64 FLOP/B); tiles of one
// token range are adjacent in blockIdx so operand re-reads hit L2.
#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct TnDev {
    const void* A; long lda; long abs;
    const void* B; long ldb; long bbs;
    float* Cp;
    float* colsum;      // optional [batch][nsplit][N1]: partial column sums of A (bias gradients for free)
    long M; int N1, N2, nsplit;
    // implicit im2col of a dense 3x3 conv's weight gradient (cC > 0; transposed-read kernel only): B is the conv INPUT
    // X [M = images*cH*cW][cC channels] and column tap*cC + ci of the logical operand is X[m + tap][ci] (zero outside the image)
    int cH, cW, cC;
};

// (Round 4 built the first level of the ordered partial sum INSIDE this launch -- the last of the 8 workgroups of a split group to
// arrive, told by a device-scope counter, added the group's tiles in split order.  Bitwise-correct on hardware and slower: 21.3 ->
// 22.6 ms per training step with write-through partial stores, 32.6 ms with device-scope fences: the last workgroup's sum is a serial
// tail at the end of a 30-60 us launch.  Removed in round 5; DESIGN.md section 5 keeps the measurement.)

// R1, R2 in {1,2}: the workgroup's output tile is (64*R1) x (64*R2); wave w owns rows [16*R1*w, 16*R1*(w+1)).
template <class T, int R1, int R2>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnDev a) {
    typedef ElemTraits<T> TR;
    constexpr int PAD = LDS_PAD_BYTES / sizeof(T);
    constexpr int VEC = Vec16<T>::N;
    constexpr int KT = 64, LDT = KT + PAD, TN1 = 64 * R1, TN2 = 64 * R2;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* At = reinterpret_cast<T*>(smem_v);        // [TN1][LDT]
    T* Bt = At + TN1 * LDT;                      // [TN2][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int t2n = (a.N2 + TN2 - 1) / TN2, ntiles = ((a.N1 + TN1 - 1) / TN1) * t2n;
    // XCD-aware block -> (token split, output tile) map: workgroups are dealt round-robin over the 8 XCDs (blocks L and
    // L+8 share an L2), so tile = (L/8) % ntiles, split = L%8 + 8*(L/(8*ntiles)) puts all tiles of one token range on
    // one XCD: the operand rows they share are fetched from HBM once and re-read from that L2.  (Speed only.)
    const int L = blockIdx.x;
    const int tile = (L >> 3) % ntiles, sp = (L & 7) + 8 * (L / (8 * ntiles)), bz = blockIdx.z;
    if (sp >= a.nsplit) return;
    const int n1_0 = (tile / t2n) * TN1, n2_0 = (tile % t2n) * TN2;
    const long per = ((a.M + a.nsplit - 1) / a.nsplit + KT - 1) / KT * KT;
    const long m_lo = (long)sp * per, m_hi = (m_lo + per < a.M) ? m_lo + per : a.M;
    const T* A = reinterpret_cast<const T*>(a.A) + (long)bz * a.abs;
    const T* B = reinterpret_cast<const T*>(a.B) + (long)bz * a.bbs;

    f32x4 acc[R1][4 * R2];
#pragma unroll
    for (int i = 0; i < R1; ++i)
#pragma unroll
        for (int j = 0; j < 4 * R2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // column sums of A ride along as one extra MFMA per K-chunk against an all-ones fragment (first n2 tile only)
    const bool do_cs = a.colsum != nullptr && (tile % t2n) == 0;
    f32x4 accs[R1];
#pragma unroll
    for (int i = 0; i < R1; ++i) accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    typename TR::frag_t ones;
    for (int e = 0; e < TR::EPL; ++e) ones[e] = from_f32<T>(1.0f);
    // Staging: each item = 4 consecutive token rows of one 16-byte column vector, written per column as ONE 8/16-byte
    // LDS store (the MFMA K axis = tokens must be lane-contiguous, so both operand tiles are transposed on the way in).
    constexpr int VPA = TN1 / VEC, VPB = TN2 / VEC, ITEMS_A = (KT / 4) * VPA, ITEMS_B = (KT / 4) * VPB;
    constexpr int NI = (ITEMS_A + ITEMS_B) / 256;          // items per thread (1 for bf16 64x64 tiles)
    static_assert((ITEMS_A + ITEMS_B) % 256 == 0, "tile staging must divide evenly over the workgroup");
    Vec16<T> x[NI][4];                                     // register stage: next K-step's rows are in flight during the MFMAs
    auto gload = [&](long m0) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int v = tid + 256 * it;
            const bool isb = v >= ITEMS_A;
            const int u = isb ? v - ITEMS_A : v;
            const int rq = u & 15, c = (u >> 4) * VEC;        // token quad fastest: conflict-free transposed LDS stores
            const T* Src = isb ? B : A;
            const long lds = isb ? a.ldb : a.lda;
            const int col = (isb ? n2_0 : n1_0) + c, nmax = isb ? a.N2 : a.N1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = m0 + rq * 4 + i;
                if (m < m_hi && col < nmax) x[it][i] = load16<T>(Src + m * lds + col);
                else x[it][i] = Vec16<T>{};
            }
        }
    };
    gload(m_lo);
    for (long m0 = m_lo; m0 < m_hi; m0 += KT) {
#pragma unroll
        for (int it = 0; it < NI; ++it) {
            const int v = tid + 256 * it;
            const bool isb = v >= ITEMS_A;
            const int u = isb ? v - ITEMS_A : v;
            const int rq = u & 15, c = (u >> 4) * VEC;
            T* Dst = (isb ? Bt : At) + c * LDT + rq * 4;
#pragma unroll
            for (int e = 0; e < VEC; ++e) store_quad(Dst + e * LDT, x[it][0], x[it][1], x[it][2], x[it][3], e);
        }
        __syncthreads();
        if (m0 + KT < m_hi) gload(m0 + KT);
#pragma unroll
        for (int kk = 0; kk < KT; kk += TR::KCHUNK) {
            typename TR::frag_t af[R1];
#pragma unroll
            for (int i = 0; i < R1; ++i) af[i] = load_frag<T>(At, LDT, (wv * R1 + i) * 16, kk);
#pragma unroll
            for (int nt = 0; nt < 4 * R2; ++nt) {
                const typename TR::frag_t bf = load_frag<T>(Bt, LDT, nt * 16, kk);
#pragma unroll
                for (int i = 0; i < R1; ++i) mma(acc[i][nt], af[i], bf);
            }
            if (do_cs)
#pragma unroll
                for (int i = 0; i < R1; ++i) mma(accs[i], af[i], ones);
        }
        __syncthreads();
    }
    float* Cp = a.Cp + (((long)bz * a.nsplit + sp) * a.N1) * a.N2;
#pragma unroll
    for (int i = 0; i < R1; ++i)
#pragma unroll
        for (int nt = 0; nt < 4 * R2; ++nt)
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + (wv * R1 + i) * 16 + (lane >> 4) * 4 + r, n2 = n2_0 + nt * 16 + (lane & 15);
                if (n1 < a.N1 && n2 < a.N2) Cp[(long)n1 * a.N2 + n2] = acc[i][nt][r];
            }
    if (do_cs && (lane & 15) == 0)
#pragma unroll
        for (int i = 0; i < R1; ++i)
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + (wv * R1 + i) * 16 + (lane >> 4) * 4 + r;
                if (n1 < a.N1) a.colsum[((long)bz * a.nsplit + sp) * a.N1 + n1] = accs[i][r];
            }
}

// ---- bf16 fast path: no transposing stores --------------------------------------------------------------------
// Both operands are token-major (K-strided), so the 64x64 kernel above transposes them on the way into LDS with
// 8-byte stores.  gfx950 can transpose on the way OUT instead (ds_read_b64_tr_b16): the tiles are stored exactly as
// they are loaded (16-byte chunks of token rows, XOR-swizzled so that both the stores and the transposed reads are
// bank-conflict free), two LDS stages with ONE barrier per 64-token step, register-staged global loads in flight
// during the MFMAs, and a (W1 x W2) in {64,128}^2 output tile per workgroup so the L1 fill rate (64 B/clk/CU) no
// longer bounds the kernel: at 128x128 a step moves 32 KB for 2.1 MFLOP.
template <int W> __host__ __device__ constexpr int tr_off(int row, int ch);      // byte offset of 16-byte chunk ch of token row `row`
template <> __host__ __device__ constexpr int tr_off<128>(int row, int ch) {       // 256-byte rows (cdna_hip_programming.md T10, image (b))
    return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
template <> __host__ __device__ constexpr int tr_off<64>(int row, int ch) {        // 128-byte rows: two rows per bank sweep
    return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2)));
}

// Fragment (16 columns starting at col0, tokens kk..kk+31) of a swizzled [64][W] image = two transposed reads (token
// rows kk+8*(lane>>4)+q and +4).  With col0 a multiple of 16 and kk a multiple of 32 the swizzle term of tr_off depends
// on the lane only, so the whole address is  lane_base + W*2*kk + ((2*col0) ^ lane_xor)  -- the lane parts are computed
// once per kernel (TrLane), leaving one v_xor + one v_add per read instead of re-deriving the swizzle (the kernel was
// issue-bound: 40 transposed reads per 32 MFMAs).
struct TrLane {
    int base[2];      // byte offset of the lane's row (+0 / +4 rows)
    int lx[2];        // lane part of the chunk swizzle, incl. the 8-byte half
};
template <int W> __device__ __forceinline__ TrLane tr_lane() {
    const int l = lane_id(), g4 = l >> 4, q = (l & 15) >> 2, p = l & 3;
    TrLane t;
    for (int hh = 0; hh < 2; ++hh) {
        const int row = 8 * g4 + q + 4 * hh;
        int sw;
        if (W == 128) sw = ((row & 3) << 2) | ((row >> 2) & 3);
        else sw = (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2);
        t.base[hh] = W * 2 * row;
        t.lx[hh] = (16 * ((p >> 1) ^ sw)) | (8 * (p & 1));
    }
    return t;
}
template <class T, int W> __device__ __forceinline__ typename ElemTraits<T>::frag_t tr_frag(const char* img, const TrLane& t, int col0, int kk) {
    const auto lo = lds_read_tr16(reinterpret_cast<const T*>(img + t.base[0] + W * 2 * kk + ((2 * col0) ^ t.lx[0])));
    const auto hi = lds_read_tr16(reinterpret_cast<const T*>(img + t.base[1] + W * 2 * kk + ((2 * col0) ^ t.lx[1])));
    return typename ElemTraits<T>::frag_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <class T, int W1, int W2>          // T: a 16-bit element type (bf16_t / f16_t)
__device__ __forceinline__ void tn_tr_body(const TnDev& a, char* smem, int L, int bz) {
    constexpr int KT = 64, RW = W1 / 64, NT = W2 / 16;
    constexpr int IMG_A = KT * W1 * 2, IMG_B = KT * W2 * 2, STAGE = IMG_A + IMG_B;
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform();
    const int t2n = (a.N2 + W2 - 1) / W2, ntiles = ((a.N1 + W1 - 1) / W1) * t2n;
    // L: block index inside this problem; XCD-aware map, as in gemm_tn_kernel
    const int tile = (L >> 3) % ntiles, sp = (L & 7) + 8 * (L / (8 * ntiles));
    if (sp >= a.nsplit) return;
    const int n1_0 = (tile / t2n) * W1, n2_0 = (tile % t2n) * W2;
    const long per = ((a.M + a.nsplit - 1) / a.nsplit + KT - 1) / KT * KT;
    const long m_lo = (long)sp * per, m_hi = (m_lo + per < a.M) ? m_lo + per : a.M;
    const T* A = reinterpret_cast<const T*>(a.A) + (long)bz * a.abs;
    const T* B = reinterpret_cast<const T*>(a.B) + (long)bz * a.bbs;

    f32x4 acc[RW][NT];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = a.colsum != nullptr && (tile % t2n) == 0;
    f32x4 accs[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    typename ElemTraits<T>::frag_t ones;
    for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;

    // Staging: thread t moves 16-byte chunk (t % C*) of token rows t / C* + RS*·it of each operand.  Everything that
    // depends on the thread only -- source pointer, swizzled LDS offset (the swizzle term is invariant under the row
    // stride RS*), column guard -- is computed once; a step costs one pointer bump per operand.
    constexpr int CA = W1 / 8, CB = W2 / 8, NIA = KT * CA / 256, NIB = KT * CB / 256, RSA = 256 / CA, RSB = 256 / CB;
    const int rowA = tid / CA, chA = tid % CA, rowB = tid / CB, chB = tid % CB;
    const bool okA = n1_0 + chA * 8 < a.N1, okB = n2_0 + chB * 8 < a.N2;
    const T* pA = A + (m_lo + rowA) * a.lda + n1_0 + chA * 8;
    const T* pB = B + (m_lo + rowB) * a.ldb + n2_0 + chB * 8;
    // conv weight gradient: this thread's 8 columns belong to ONE tap (cC % 8 == 0): its rows are the input rows shifted by the
    // tap, read only where the shifted pixel is inside the image
    int cdy = 0, cdx = 0;
    const int cHW = a.cH * a.cW;
    if (a.cC) {
        const int colB = n2_0 + chB * 8, tap = colB / a.cC, ci = colB - tap * a.cC;
        cdy = tap / 3 - 1;
        cdx = tap - (tap / 3) * 3 - 1;
        pB = B + (m_lo + rowB + (long)cdy * a.cW + cdx) * a.ldb + ci;
    }
    const int ldsA = tr_off<W1>(rowA, chA), ldsB = IMG_A + tr_off<W2>(rowB, chB);
    static_assert(tr_off<W1>(RSA, 0) == RSA * W1 * 2 && tr_off<W2>(RSB, 0) == RSB * W2 * 2, "row stride must not change the swizzle");
    Vec16<T> xA[NIA], xB[NIB];
    auto gload = [&](long m0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NIA; ++it) {
            if (okA && m0 + rowA + RSA * it < m_hi) xA[it] = load16<T>(pA + (long)(RSA * it) * a.lda);
            else xA[it] = Vec16<T>{};
        }
#pragma unroll
        for (int it = 0; it < NIB; ++it) {
            bool ok = okB && m0 + rowB + RSB * it < m_hi;
            if (a.cC) {           // wave-uniform
                const unsigned p = (unsigned)(m0 + rowB + RSB * it) % (unsigned)cHW, py = p / (unsigned)a.cW;     // M < 2^31: 32-bit division
                const int y = (int)py + cdy, x = (int)(p - py * (unsigned)a.cW) + cdx;
                ok = ok && y >= 0 && y < a.cH && x >= 0 && x < a.cW;
            }
            if (ok) xB[it] = load16<T>(pB + (long)(RSB * it) * a.ldb);
            else xB[it] = Vec16<T>{};
        }
        pA += (long)KT * a.lda;
        pB += (long)KT * a.ldb;
    };
    auto sstore = [&](char* stage) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NIA; ++it) store16<T>(reinterpret_cast<T*>(stage + ldsA + RSA * it * W1 * 2), xA[it]);
#pragma unroll
        for (int it = 0; it < NIB; ++it) store16<T>(reinterpret_cast<T*>(stage + ldsB + RSB * it * W2 * 2), xB[it]);
    };
    const TrLane tla = tr_lane<W1>(), tlb = tr_lane<W2>();
    gload(m_lo);
    int buf = 0;
    for (long m0 = m_lo; m0 < m_hi; m0 += KT, buf ^= 1) {
        char* stage = smem + buf * STAGE;
        sstore(stage);
        __syncthreads();         // the only barrier per step: the other stage was last read before the previous barrier
        if (m0 + KT < m_hi) gload(m0 + KT);
#pragma unroll
        for (int kk = 0; kk < KT; kk += 32) {
            typename ElemTraits<T>::frag_t af[RW];
#pragma unroll
            for (int i = 0; i < RW; ++i) af[i] = tr_frag<T, W1>(stage, tla, (wv * RW + i) * 16, kk);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const typename ElemTraits<T>::frag_t bf = tr_frag<T, W2>(stage + IMG_A, tlb, nt * 16, kk);
#pragma unroll
                for (int i = 0; i < RW; ++i) mma(acc[i][nt], af[i], bf);
            }
            if (do_cs)
#pragma unroll
                for (int i = 0; i < RW; ++i) mma(accs[i], af[i], ones);
        }
    }
    float* Cp = a.Cp + (((long)bz * a.nsplit + sp) * a.N1) * a.N2;
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + (wv * RW + i) * 16 + (lane >> 4) * 4 + r, n2 = n2_0 + nt * 16 + (lane & 15);
                if (n1 < a.N1 && n2 < a.N2) {
                    Cp[(long)n1 * a.N2 + n2] = acc[i][nt][r];
                }
            }
    if (do_cs && (lane & 15) == 0)
#pragma unroll
        for (int i = 0; i < RW; ++i)
            for (int r = 0; r < 4; ++r) {
                const int n1 = n1_0 + (wv * RW + i) * 16 + (lane >> 4) * 4 + r;
                if (n1 < a.N1) a.colsum[((long)bz * a.nsplit + sp) * a.N1 + n1] = accs[i][r];
            }
}

template <class T, int W1, int W2>
__global__ __launch_bounds__(256, 2) void gemm_tn_tr_kernel(TnDev a) {      // 2 waves/SIMD = 2 workgroups/CU (2 x 64 KB LDS)
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    tn_tr_body<T, W1, W2>(a, reinterpret_cast<char*>(smem_v), blockIdx.x, blockIdx.z);      // [2 stages][A image | B image]
}

// Grouped form: up to MPHSIR_TN_GROUP_MAX independent token-reduction GEMMs in ONE launch (block ranges per problem).
// The weight-gradient GEMMs of a backward function feed nothing but the final partial reduction, so they are all
// issued together at its end: at the small pyramid levels (8k-32k tokens) each of them alone fills a fraction of the
// chip for ~20 us; together they fill it once.
struct TnGroupDev {
    TnDev p[MPHSIR_TN_GROUP_MAX];
    int blk0[MPHSIR_TN_GROUP_MAX + 1];
    int n;
};

template <class T, int W1, int W2>
__global__ __launch_bounds__(256, 2) void gemm_tn_tr_group_kernel(TnGroupDev g) {
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    // constant indices only (a dynamically indexed by-value kernel argument is copied to scratch by every lane)
    TnDev a = g.p[0];
    int b0 = 0;
#pragma unroll
    for (int k = 1; k < MPHSIR_TN_GROUP_MAX; ++k)
        if (k < g.n && (int)blockIdx.x >= g.blk0[k]) { a = g.p[k]; b0 = g.blk0[k]; }
    tn_tr_body<T, W1, W2>(a, reinterpret_cast<char*>(smem_v), (int)blockIdx.x - b0, 0);
}

// ---- ring form (16-bit types): ONE 512-thread workgroup per CU, token rows by LDS-DMA ------------------------------------------
// A workgroup is EIGHT waves = two groups of four: its token range goes through a ring of 32-token slots filled by LDS-DMA
// (global_load_lds_dwordx4 straight into the swizzled image the transposed reads want: the swizzle sits in the per-lane SOURCE
// address, the destination is lane-linear; no staging registers, RG_RING - 2 slots in flight all the time, retired by a counted
// vmcnt in front of an LDS-only barrier); group g multiplies slot 2 it + g.  Both groups keep a full (W1 x W2) tile of
// accumulators; at the end group 1's tile is added to group 0's through LDS (always in that order: deterministic) and the sum
// leaves as whole 16-byte row chunks: half the partial tiles per launch of two 4-wave workgroups.  Rows outside the token range /
// the image (conv weight gradient) and columns outside the matrix are fetched from a page of zeros.
// Measured on MI355X (tools/bench/bench_tn.py, tools/lab/ring_lab.hip): a bare DMA ring streams 5.9 TB/s with one
// workgroup per CU, and problems with ONE output tile run 1.2-1.7x faster in this form (dWproj 64x64 at M = 131072: 19.1 -> 11.5
// us, 192x64: 23.3 -> 16.6, 128x128: 27.2 -> 21.0).  Problems with several tiles do NOT: the tiles of a token range re-read the
// shared operand (dWqkv 384x128: 134 MB from HBM but 201 MB into the CUs), the kernel above already moves those CU-side bytes at
// 5.4 TB/s, and with one workgroup per CU this form moves them more slowly (56 us against 37).  The host picks the form per
// problem (ops.gemm_tn).
constexpr int RG_ST = 32;            // tokens per ring slot (= one MFMA K chunk)
constexpr int RG_RING = 8;           // ring slots; two are being multiplied, the others are in flight
__device__ __attribute__((aligned(16))) const unsigned char g_tn_zero_page[16] = {0};

template <class T, int W1, int W2>
__device__ __forceinline__ void tn_ring_body(const TnDev& a, char* smem, int L, int bz) {
    constexpr int RW = W1 / 64, NT = W2 / 16;
    constexpr int IMG_A = RG_ST * W1 * 2, IMG_B = RG_ST * W2 * 2, SLOT = IMG_A + IMG_B;
    constexpr int NIA = IMG_A / 1024, NIB = IMG_B / 1024, NI = NIA + NIB;      // DMA instructions (1 KB each) per slot
    constexpr int CMAX = (NI + 7) / 8, CMIN = NI / 8;                          // ... per wave: waves < NI % 8 issue CMAX, the others CMIN
    static_assert(NI >= 8 && CMAX <= 2, "ring form: 8..16 DMA instructions per slot");
    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform(), grp = wv >> 2, wv4 = wv & 3;
    const int t2n = (a.N2 + W2 - 1) / W2, ntiles = ((a.N1 + W1 - 1) / W1) * t2n;
    const int tile = (L >> 3) % ntiles, sp = (L & 7) + 8 * (L / (8 * ntiles));      // XCD-aware map, as in gemm_tn_kernel
    if (sp >= a.nsplit) return;
    const int n1_0 = (tile / t2n) * W1, n2_0 = (tile % t2n) * W2;
    const long per = ((a.M + a.nsplit - 1) / a.nsplit + 63) / 64 * 64;
    const long m_lo = (long)sp * per, m_hi = (m_lo + per < a.M) ? m_lo + per : a.M;
    const int nslots = m_hi > m_lo ? (int)((m_hi - m_lo + RG_ST - 1) / RG_ST) : 0;
    const char* A = reinterpret_cast<const char*>(a.A) + (long)bz * a.abs * 2;
    const char* B = reinterpret_cast<const char*>(a.B) + (long)bz * a.bbs * 2;
    const char* zero = reinterpret_cast<const char*>(g_tn_zero_page);

    // ---- this wave's DMA instructions: q = wv (+ 8); per lane: row inside the slot, source chunk (the swizzle of tr_off applied to
    // the SOURCE column: LDS byte 16 * lane of the instruction's 1 KB holds chunk pos ^ swz(row) of its row)
    const char* src[CMAX];          // source of slot 0 (valid or not: see ok*)
    long step[CMAX];                // bytes per slot
    int row[CMAX];                  // row inside the slot
    bool colok[CMAX], isb[CMAX];
    unsigned ldsoff[CMAX];          // wave-uniform byte offset inside the slot
    int cdy[CMAX], cdx[CMAX];       // conv weight gradient: the tap of this lane's 8 columns of B, per DMA instruction (one shared pair
                                    // was right only while every B instruction of a wave happened to land on the same tap)
    const int cHW = a.cH * a.cW;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        const int q = wv + 8 * c;                       // wave-uniform
        const bool b = q >= NIA;
        const int j = b ? q - NIA : q;
        int r, pos, sw;
        if ((b ? W2 : W1) == 128) { r = 4 * j + (lane >> 4); pos = lane & 15; sw = ((r & 3) << 2) | ((r >> 2) & 3); }
        else { r = 8 * j + (lane >> 3); pos = lane & 7; sw = (((r >> 1) & 1) << 1) | (((r >> 3) & 1) << 2); }
        const int ch = pos ^ sw;
        row[c] = r;
        isb[c] = b;
        ldsoff[c] = (b ? IMG_A : 0) + 1024 * j;
        const int col = (b ? n2_0 : n1_0) + ch * 8;
        colok[c] = q < NI && col < (b ? a.N2 : a.N1);
        const long ld = b ? a.ldb : a.lda;
        step[c] = (long)RG_ST * ld * 2;
        cdy[c] = cdx[c] = 0;
        if (b && a.cC && q < NI) {
            const int tap = col / a.cC, ci = col - tap * a.cC;
            cdy[c] = tap / 3 - 1;
            cdx[c] = tap - (tap / 3) * 3 - 1;
            src[c] = B + ((m_lo + r + (long)cdy[c] * a.cW + cdx[c]) * ld + ci) * 2;
        } else {
            src[c] = (b ? B : A) + ((m_lo + r) * ld + col) * 2;
        }
    }
    auto issue = [&](int sidx) __attribute__((always_inline)) {          // DMA of slot sidx (wave-uniform) into its ring position
        char* slot = smem + (sidx % RG_RING) * SLOT;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (wv + 8 * c >= NI) continue;                              // wave-uniform
            const long m = m_lo + (long)sidx * RG_ST + row[c];
            bool ok = colok[c] && m < m_hi;
            if (isb[c] && a.cC) {                                        // wave-uniform
                const unsigned p = (unsigned)m % (unsigned)cHW, py = p / (unsigned)a.cW;     // M < 2^31: 32-bit division
                const int y = (int)py + cdy[c], x = (int)(p - py * (unsigned)a.cW) + cdx[c];
                ok = ok && y >= 0 && y < a.cH && x >= 0 && x < a.cW;
            }
            const char* g = ok ? src[c] + (long)sidx * step[c] : zero;
            MPHSIR_LDS_DMA16P(g, slot + ldsoff[c]);
        }
    };

    f32x4 acc[RW][NT];
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = a.colsum != nullptr && (tile % t2n) == 0;
    f32x4 accs[RW];
#pragma unroll
    for (int i = 0; i < RW; ++i) accs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    typename ElemTraits<T>::frag_t ones;
    for (int e = 0; e < 8; ++e) ones[e] = (T)1.0f;
    const TrLane tla = tr_lane<W1>(), tlb = tr_lane<W2>();

    constexpr int LEAD = RG_RING - 2;                 // slots issued ahead of the pair being multiplied
    for (int s0 = 0; s0 < LEAD && s0 < nslots; ++s0) issue(s0);
    const int niter = (nslots + 1) / 2;
    const bool many = wv < NI % 8 || NI % 8 == 0;     // this wave issues CMAX instructions per slot (else CMIN)
    for (int it = 0; it < niter; ++it) {
        // slots 2 it, 2 it + 1 must have landed; the LEAD - 2 younger slots stay in flight (vmcnt counts in issue order).  Near
        // the end fewer younger ones exist: wait for everything.
        if (2 * it + LEAD <= nslots) {
            if (many) wait_vmcnt<(LEAD - 2) * CMAX>();
            else wait_vmcnt<(LEAD - 2) * (CMIN > 0 ? CMIN : 1)>();
        } else {
            wait_vmcnt<0>();
        }
        lds_barrier();                                // ... for every wave; and everybody is done with the pair of iteration it - 1
        if (2 * it + LEAD < nslots) issue(2 * it + LEAD);
        if (2 * it + LEAD + 1 < nslots) issue(2 * it + LEAD + 1);
        const int sidx = 2 * it + grp;
        if (sidx < nslots) {
            const char* slot = smem + (sidx % RG_RING) * SLOT;
            typename ElemTraits<T>::frag_t af[RW];
#pragma unroll
            for (int i = 0; i < RW; ++i) af[i] = tr_frag<T, W1>(slot, tla, (wv4 * RW + i) * 16, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const typename ElemTraits<T>::frag_t bf = tr_frag<T, W2>(slot + IMG_A, tlb, nt * 16, 0);
#pragma unroll
                for (int i = 0; i < RW; ++i) mma(acc[i][nt], af[i], bf);
            }
            if (do_cs)
#pragma unroll
                for (int i = 0; i < RW; ++i) mma(accs[i], af[i], ones);
        }
    }
    // ---- the two groups' tiles -> one partial tile: group 1 through LDS (the ring is dead), group 0 adds (fixed order), everybody
    // stores whole 16-byte row chunks
    constexpr int LDC = W2 + 4;
    float* Cs = reinterpret_cast<float*>(smem);                 // [W1][LDC]
    float* Ss = Cs + W1 * LDC;                                  // [W1] column sums
    wait_vmcnt<0>();
    lds_barrier();
    for (int g = 1; g >= 0; --g) {
        if (grp == g) {
#pragma unroll
            for (int i = 0; i < RW; ++i) {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* p = Cs + ((wv4 * RW + i) * 16 + (lane >> 4) * 4 + r) * LDC + nt * 16 + (lane & 15);
                        *p = g ? acc[i][nt][r] : acc[i][nt][r] + *p;
                    }
                if (do_cs && (lane & 15) == 0)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* p = Ss + (wv4 * RW + i) * 16 + (lane >> 4) * 4 + r;
                        *p = g ? accs[i][r] : accs[i][r] + *p;
                    }
            }
        }
        lds_barrier();
    }
    float* Cp = a.Cp + (((long)bz * a.nsplit + sp) * a.N1) * a.N2;
    constexpr int CPR = W2 / 4;                                 // 16-byte chunks per tile row
    for (int idx = tid; idx < W1 * CPR; idx += 512) {
        const int r = idx / CPR, c = (idx % CPR) * 4, n1 = n1_0 + r, n2 = n2_0 + c;
        if (n1 < a.N1 && n2 < a.N2) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(Cs + r * LDC + c);
            *reinterpret_cast<f32x4*>(Cp + (long)n1 * a.N2 + n2) = v;
        }
    }
    if (do_cs && tid < W1 && n1_0 + tid < a.N1) a.colsum[((long)bz * a.nsplit + sp) * a.N1 + n1_0 + tid] = Ss[tid];
}

template <class T, int W1, int W2> constexpr size_t tn_ring_lds() {
    constexpr size_t ring = (size_t)RG_RING * RG_ST * (W1 + W2) * 2, comb = (size_t)W1 * (W2 + 4) * 4 + W1 * 4;
    return ring > comb ? ring : comb;
}

template <class T, int W1, int W2>
__global__ __launch_bounds__(512) void gemm_tn_ring_kernel(TnDev a) {
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    tn_ring_body<T, W1, W2>(a, reinterpret_cast<char*>(smem_v), blockIdx.x, blockIdx.z);
}

template <class T, int W1, int W2>
__global__ __launch_bounds__(512) void gemm_tn_ring_group_kernel(TnGroupDev g) {
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    TnDev a = g.p[0];
    int b0 = 0;
#pragma unroll
    for (int k = 1; k < MPHSIR_TN_GROUP_MAX; ++k)
        if (k < g.n && (int)blockIdx.x >= g.blk0[k]) { a = g.p[k]; b0 = g.blk0[k]; }
    tn_ring_body<T, W1, W2>(a, reinterpret_cast<char*>(smem_v), (int)blockIdx.x - b0, 0);
}

template <class T, int W1, int W2>
static int launch_tn_ring_group(const TnGroupDev& g, hipStream_t s) {
    const size_t shmem = tn_ring_lds<T, W1, W2>();
    allow_big_lds(gemm_tn_ring_group_kernel<T, W1, W2>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GEMM_TN, (gemm_tn_ring_group_kernel<T, W1, W2>), dim3(g.blk0[g.n]), dim3(512), shmem, s, g);
    return MPHSIR_OK;
}

template <class T, int W1, int W2>
static int launch_tn_ring(const TnDev& d, int batch, hipStream_t s) {
    const int ntiles = ((d.N1 + W1 - 1) / W1) * ((d.N2 + W2 - 1) / W2);
    dim3 grid(ntiles * ((d.nsplit + 7) / 8 * 8), 1, batch);
    const size_t shmem = tn_ring_lds<T, W1, W2>();
    allow_big_lds(gemm_tn_ring_kernel<T, W1, W2>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GEMM_TN, (gemm_tn_ring_kernel<T, W1, W2>), grid, dim3(512), shmem, s, d);
    return MPHSIR_OK;
}

template <class T, int W1, int W2>
static int launch_tn_tr_group(const TnGroupDev& g, hipStream_t s) {
    const size_t shmem = 2 * (size_t)(W1 + W2) * 64 * 2;
    allow_big_lds(gemm_tn_tr_group_kernel<T, W1, W2>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GEMM_TN, (gemm_tn_tr_group_kernel<T, W1, W2>), dim3(g.blk0[g.n]), dim3(256), shmem, s, g);
    return MPHSIR_OK;
}

template <class T, int W1, int W2>
static int launch_tn_tr(const TnDev& d, int batch, hipStream_t s) {
    const int ntiles = ((d.N1 + W1 - 1) / W1) * ((d.N2 + W2 - 1) / W2);
    dim3 grid(ntiles * ((d.nsplit + 7) / 8 * 8), 1, batch);
    const size_t shmem = 2 * (size_t)(W1 + W2) * 64 * 2;
    allow_big_lds(gemm_tn_tr_kernel<T, W1, W2>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GEMM_TN, (gemm_tn_tr_kernel<T, W1, W2>), grid, dim3(256), shmem, s, d);
    return MPHSIR_OK;
}

template <class T, int R1, int R2>
static int launch_tn(const TnDev& d, int batch, hipStream_t s) {
    constexpr int esz = sizeof(T), pad = LDS_PAD_BYTES / esz;
    const int ntiles = ((d.N1 + 64 * R1 - 1) / (64 * R1)) * ((d.N2 + 64 * R2 - 1) / (64 * R2));
    dim3 grid(ntiles * ((d.nsplit + 7) / 8 * 8), 1, batch);
    const size_t shmem = (size_t)(64 * R1 + 64 * R2) * (64 + pad) * esz;
    allow_big_lds(gemm_tn_kernel<T, R1, R2>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_GEMM_TN, (gemm_tn_kernel<T, R1, R2>), grid, dim3(256), shmem, s, d);
    return MPHSIR_OK;
}

}  // namespace mphsir

extern "C" int mphsir_gemm_tn(const void* A, int64_t lda, int64_t a_batch_stride, const void* B, int64_t ldb, int64_t b_batch_stride,
                              float* Cpart, float* colsum_part, int64_t M, int32_t N1, int32_t N2, int32_t nsplit, int32_t batch, int32_t tile128,
                              int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(A && B && Cpart, "gemm_tn: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "gemm_tn: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2, vec = 16 / esz;
    MPHSIR_REQUIRE(M > 0 && N1 > 0 && N2 > 0 && N1 % vec == 0 && N2 % vec == 0 && nsplit > 0 && batch > 0 && nsplit < 65536 && batch < 65536,
                   "gemm_tn: bad shape (N1, N2 must be multiples of %d)", vec);
    MPHSIR_REQUIRE(aligned16(A) && aligned16(B) && (lda * esz) % 16 == 0 && (ldb * esz) % 16 == 0 &&
                       (a_batch_stride * esz) % 16 == 0 && (b_batch_stride * esz) % 16 == 0, "gemm_tn: 16-byte alignment required");
    TnDev d{A, (long)lda, (long)a_batch_stride, B, (long)ldb, (long)b_batch_stride, Cpart, colsum_part, (long)M, N1, N2, nsplit};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool big = tile128 > 0;
    if (dtype == MPHSIR_F32) return big ? launch_tn<float, 2, 2>(d, batch, s) : launch_tn<float, 1, 1>(d, batch, s);
    if (!big) return dtype == MPHSIR_BF16 ? launch_tn<bf16_t, 1, 1>(d, batch, s) : launch_tn<f16_t, 1, 1>(d, batch, s);
    // 16-bit "big": the transposed-read kernel (1) or its ring form (2); each operand's tile width follows its matrix width
#define MPHSIR_TN_RING(T16)                                                                                             \
    if (N1 > 64) return N2 > 64 ? launch_tn_ring<T16, 128, 128>(d, batch, s) : launch_tn_ring<T16, 128, 64>(d, batch, s); \
    return N2 > 64 ? launch_tn_ring<T16, 64, 128>(d, batch, s) : launch_tn_ring<T16, 64, 64>(d, batch, s);
    if (tile128 == 2) {
        if (dtype == MPHSIR_BF16) { MPHSIR_TN_RING(bf16_t) }
        MPHSIR_TN_RING(f16_t)
    }
#undef MPHSIR_TN_RING
#define MPHSIR_TN_TR(T16)                                                                                             \
    if (N1 > 64) return N2 > 64 ? launch_tn_tr<T16, 128, 128>(d, batch, s) : launch_tn_tr<T16, 128, 64>(d, batch, s); \
    return N2 > 64 ? launch_tn_tr<T16, 64, 128>(d, batch, s) : launch_tn_tr<T16, 64, 64>(d, batch, s);
    if (dtype == MPHSIR_BF16) { MPHSIR_TN_TR(bf16_t) }
    MPHSIR_TN_TR(f16_t)
#undef MPHSIR_TN_TR
}

extern "C" int mphsir_gemm_tn_group(const mphsir_gemm_tn_problem* probs, int32_t n, int32_t form, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(probs && n > 0 && n <= MPHSIR_TN_GROUP_MAX, "gemm_tn_group: 1..%d problems per call", MPHSIR_TN_GROUP_MAX);
    MPHSIR_REQUIRE(dtype == MPHSIR_BF16 || dtype == MPHSIR_F16, "gemm_tn_group: 16-bit element types only (the transposed-LDS-read kernel)");
    MPHSIR_REQUIRE(form == 1 || form == 2, "gemm_tn_group: form 1 (transposed-read kernel) or 2 (ring form)");
    TnGroupDev g;
    g.n = n;
    bool wide1 = false, wide2 = false;
    for (int k = 0; k < n; ++k) { wide1 = wide1 || probs[k].N1 > 64; wide2 = wide2 || probs[k].N2 > 64; }
    const int W1 = wide1 ? 128 : 64, W2 = wide2 ? 128 : 64;
    int blocks = 0;
    for (int k = 0; k < n; ++k) {
        const mphsir_gemm_tn_problem& q = probs[k];
        MPHSIR_REQUIRE(q.A && q.B && q.Cpart && q.M > 0 && q.N1 > 0 && q.N2 > 0 && q.N1 % 8 == 0 && q.N2 % 8 == 0 && q.nsplit > 0 && q.nsplit < 65536,
                       "gemm_tn_group: bad problem %d", k);
        MPHSIR_REQUIRE(aligned16(q.A) && aligned16(q.B) && (q.lda * 2) % 16 == 0 && (q.ldb * 2) % 16 == 0, "gemm_tn_group: 16-byte alignment required");
        g.p[k] = TnDev{q.A, (long)q.lda, 0, q.B, (long)q.ldb, 0, q.Cpart, q.colsum_part, (long)q.M, q.N1, q.N2, q.nsplit};
        g.blk0[k] = blocks;
        const int ntiles = ((q.N1 + W1 - 1) / W1) * ((q.N2 + W2 - 1) / W2);
        blocks += ntiles * ((q.nsplit + 7) / 8 * 8);
    }
    g.blk0[n] = blocks;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MPHSIR_TN_GRPR(T16)                                                                                       \
    if (wide1) return wide2 ? launch_tn_ring_group<T16, 128, 128>(g, s) : launch_tn_ring_group<T16, 128, 64>(g, s); \
    return wide2 ? launch_tn_ring_group<T16, 64, 128>(g, s) : launch_tn_ring_group<T16, 64, 64>(g, s);
    if (form == 2) {
        if (dtype == MPHSIR_BF16) { MPHSIR_TN_GRPR(bf16_t) }
        MPHSIR_TN_GRPR(f16_t)
    }
#undef MPHSIR_TN_GRPR
#define MPHSIR_TN_GRP(T16)                                                                                    \
    if (wide1) return wide2 ? launch_tn_tr_group<T16, 128, 128>(g, s) : launch_tn_tr_group<T16, 128, 64>(g, s); \
    return wide2 ? launch_tn_tr_group<T16, 64, 128>(g, s) : launch_tn_tr_group<T16, 64, 64>(g, s);
    if (dtype == MPHSIR_BF16) { MPHSIR_TN_GRP(bf16_t) }
    MPHSIR_TN_GRP(f16_t)
#undef MPHSIR_TN_GRP
}

extern "C" int mphsir_conv3x3_wgrad(const void* dY, int64_t lddy, const void* X, int64_t ldx, float* Cpart, int32_t B, int32_t H, int32_t W,
                                    int32_t Cout, int32_t Cin, int32_t nsplit, int32_t form, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(dY && X && Cpart, "conv3x3_wgrad: null pointer");
    MPHSIR_REQUIRE(dtype == MPHSIR_BF16 || dtype == MPHSIR_F16, "conv3x3_wgrad: 16-bit element types only (otherwise mphsir_im2col3x3 + mphsir_gemm_tn)");
    MPHSIR_REQUIRE(B > 0 && H > 0 && W > 0 && Cout > 0 && Cin > 0 && Cout % 8 == 0 && Cin % 8 == 0 && nsplit > 0 && nsplit < 65536,
                   "conv3x3_wgrad: bad shape (Cout, Cin must be multiples of 8)");
    MPHSIR_REQUIRE(aligned16(dY) && aligned16(X) && (lddy * 2) % 16 == 0 && (ldx * 2) % 16 == 0, "conv3x3_wgrad: 16-byte alignment required");
    TnDev d{dY, (long)lddy, 0, X, (long)ldx, 0, Cpart, nullptr, (long)B * H * W, Cout, 9 * Cin, nsplit, H, W, Cin};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
#define MPHSIR_CW_RING(T16) return Cout > 64 ? launch_tn_ring<T16, 128, 128>(d, 1, s) : launch_tn_ring<T16, 64, 128>(d, 1, s);
    if (form == 2) {
        if (dtype == MPHSIR_BF16) { MPHSIR_CW_RING(bf16_t) }
        MPHSIR_CW_RING(f16_t)
    }
#undef MPHSIR_CW_RING
#define MPHSIR_CW_TR(T16) return Cout > 64 ? launch_tn_tr<T16, 128, 128>(d, 1, s) : launch_tn_tr<T16, 64, 128>(d, 1, s);
    if (dtype == MPHSIR_BF16) { MPHSIR_CW_TR(bf16_t) }
    MPHSIR_CW_TR(f16_t)
#undef MPHSIR_CW_TR
}
