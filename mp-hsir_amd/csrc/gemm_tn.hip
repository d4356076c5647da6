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
};

// R1, R2 in {1,2}: the workgroup's output tile is (64*R1) x (64*R2); wave w owns rows [16*R1*w, 16*R1*(w+1)).
template <class T, int R1, int R2>
__global__ __launch_bounds__(256) void gemm_tn_kernel(TnDev a) {
    typedef ElemTraits<T> TR;
    constexpr int PAD = 16 / sizeof(T);
    constexpr int VEC = Vec16<T>::N;
    constexpr int KT = 64, LDT = KT + PAD, TN1 = 64 * R1, TN2 = 64 * R2;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* At = reinterpret_cast<T*>(smem_v);        // [TN1][LDT]
    T* Bt = At + TN1 * LDT;                      // [TN2][LDT]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
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

template <class T, int R1, int R2>
static int launch_tn(const TnDev& d, int batch, hipStream_t s) {
    constexpr int esz = sizeof(T), vec = 16 / esz;
    const int ntiles = ((d.N1 + 64 * R1 - 1) / (64 * R1)) * ((d.N2 + 64 * R2 - 1) / (64 * R2));
    dim3 grid(ntiles * ((d.nsplit + 7) / 8 * 8), 1, batch);
    const size_t shmem = (size_t)(64 * R1 + 64 * R2) * (64 + vec) * esz;
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
    MPHSIR_REQUIRE(dtype == MPHSIR_F32 || dtype == MPHSIR_BF16, "gemm_tn: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2, vec = 16 / esz;
    MPHSIR_REQUIRE(M > 0 && N1 > 0 && N2 > 0 && N1 % vec == 0 && N2 % vec == 0 && nsplit > 0 && batch > 0 && nsplit < 65536 && batch < 65536,
                   "gemm_tn: bad shape (N1, N2 must be multiples of %d)", vec);
    MPHSIR_REQUIRE(aligned16(A) && aligned16(B) && (lda * esz) % 16 == 0 && (ldb * esz) % 16 == 0 &&
                       (a_batch_stride * esz) % 16 == 0 && (b_batch_stride * esz) % 16 == 0, "gemm_tn: 16-byte alignment required");
    TnDev d{A, (long)lda, (long)a_batch_stride, B, (long)ldb, (long)b_batch_stride, Cpart, colsum_part, (long)M, N1, N2, nsplit};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const bool big = tile128 > 0;
    if (dtype == MPHSIR_F32) return big ? launch_tn<float, 2, 2>(d, batch, s) : launch_tn<float, 1, 1>(d, batch, s);
    return big ? launch_tn<bf16_t, 2, 2>(d, batch, s) : launch_tn<bf16_t, 1, 1>(d, batch, s);
    return MPHSIR_OK;
}
