// qkv_dwconv_gram, ROW-WALKING form: the fused pass A of the global spectral (channel) attention with the 1x1-conv output
// kept in REGISTERS between the matrix cores and the depthwise 3x3.
//
// Reference: Spectral_Attention.forward net/MP_HSIR.py:96-107 (== Attention :301-313):
//     qkv = qkv_dwconv(qkv(x));  q,k,v = chunk(3);  q,k -> F.normalize over pixels;  attn = q @ k^T
//
// The tile form (spectral_fused.hip) sends t = qkv(x) through an fp32 LDS tile between its MFMA phase and its depthwise
// phase, barrier-separated in one workgroup per CU, recomputes a 1.5x halo and re-stages the weights per channel slab.
// Here a workgroup owns a 32-pixel-wide column strip of the image and walks DOWN it, one image row per step:
//   * a wave owns 16 (wide heads: 32) output channels of q, k or v for the whole walk: its 1x1 weights (MFMA A fragments)
//     and its 9 taps per channel are loaded ONCE and stay in registers;
//   * per row the 32 pixels are two interleaved 16-pixel MFMA column blocks (block b, lane j = pixel 2j+b).  The
//     accumulators (lane = 4 channels x 1 pixel) ARE the depthwise input: the row above / below is another register
//     (two partial output rows are kept: every t row is multiplied into the three rows it touches and then dropped, so no
//     row is ever computed twice), the pixel to the left / right is either the other block's register of the SAME lane
//     or one DPP row shift away.  The strip's outer columns (x0-1, x0+32) travel with the row in the ring; their t comes from
//     a third ("edge") MFMA block per row whose lanes 0..7 / 8..15 carry the left / right column, so it sits in lanes 0 / 15
//     where the DPP moves of the taps pick it up as the value a lane without a source keeps;
//   * the x rows arrive by LDS-DMA (global_load_lds, no registers) in an 8-slot ring, issued 8 rows ahead and retired with a
//     counted vmcnt; q,k,v (training: t too) leave through a two-slot LDS row image [pixel][channel]: the Gram reads it
//     pixel-strided (ds_read_b64_tr_b16) and the global stores are 16-byte chunks of >= 96 contiguous bytes per pixel;
//   * Gram tiles and sums of squares are PERSISTENT per-wave accumulators over the whole walk (fixed order, no atomics):
//     one store per workgroup into its Gpart / Spart slot at the end.
// The walk is software-pipelined: step i issues the LDS reads for the MFMAs of row i+1, for the Gram and the stores of
// output row i-3, runs the depthwise taps of row i (VALU) beside those MFMAs, writes output row i-2 to the row image and
// meets ONE workgroup barrier (LDS-only fence: the DMA ring and the global stores stay in flight across it).
#include <stdlib.h>

#include <type_traits>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

constexpr int RW_SW = 32;                         // strip width in pixels
constexpr int RW_RPX = RW_SW + 2;                 // ring row: left outer column | 32 pixels | right outer column
constexpr int RW_RING = 8;                        // ring slots (rows); C = 256: 4 (RwCfg::RING): eight 18 KB rows do not fit beside the row images
constexpr int RW_LEAD = 2;                        // a row is first read one step before the taps run on it
constexpr int RW_UNR = 1;                         // steady steps come in groups of RW_UNR with the low bits of the ring position static

struct RowGramDev {
    const void* X; long ldx;
    const float* ln_w; const float* ln_b;        // optional LayerNorm over C applied to every x row in the ring (LN builds)
    const void* Wqkv;                            // [3C][C]
    const float* w9; long ldw;                   // [9][ldw]: taps of q | k | v channels
    void* V; long ldvo;
    float* Gpart; float* Spart;
    int B, H, W, nseg;                           // nseg row segments per strip: (W / 32) * nseg partial slots per sample
    void* Tout; long ldt;
    void* QKout; long ldqk;
    unsigned long long* dbg;                     // diagnostics (mphsir_debug): shader-clock stamps of workgroup 0, wave 0
};

// waves per workgroup: 3 (q | k | v) x channels of each per workgroup / channels per wave (a function of the head width only)
template <int HD> constexpr int rw_waves() { return 3 * ((HD == 48 ? 48 : (HD == 96 ? 96 : 64)) / (HD == 96 ? 32 : 16)); }

template <class T, int C, int HD> struct RwCfg {
    static constexpr int CT = HD == 48 ? 48 : (HD == 96 ? 96 : 64);   // channels of q (of k, of v) per workgroup
    static constexpr int NBW = HD == 96 ? 2 : 1;             // 16-channel MFMA row blocks per wave
    static constexpr int CW = 16 * NBW;                      // channels per wave
    static constexpr int WPT = CT / CW;                      // waves per q | k | v
    static constexpr int NW = rw_waves<HD>(), THREADS = 64 * NW;
    static constexpr int HPG = CT / HD;                      // heads per workgroup
    static constexpr int NKC = C / 32;                       // K chunks of the 1x1 conv
    static constexpr int CH1 = C / 8 + 1;                    // 16-byte chunks per ring pixel row: C channels + 16 B (pitch / 16 B
                                                             // odd: conflict-free ds_read_b128 of the rows 2j+b)
    static constexpr int XPE = 8 * CH1;                      // elements per ring pixel row
    static constexpr int NCH = RW_RPX * CH1;                 // chunks per ring slot
    static constexpr int NDI = (NCH + 63) / 64, NDQ = (NDI + NW - 1) / NW;    // DMA instructions per row / per wave
    // the counted vmcnt wait in front of the ring barrier tells apart only waves that issue NDQ and NDQ - 1 instructions per row
    static_assert(NDQ <= 2, "row ring: a wave issuing fewer than NDQ - 1 DMA instructions per row would wait on too large a count");
    static constexpr int OPE = 3 * CT + 8;                   // row image [q | k | v] + 16 B
    static constexpr int NT = HD / 16, NTW = HPG * NT * NT, TPW = (NTW + NW - 1) / NW;
    static constexpr int RING = C > 192 ? RW_RING / 2 : RW_RING;      // ring slots
    static constexpr size_t ring_elems = (size_t)RING * NCH * 8, img_elems = (size_t)2 * RW_SW * OPE;
    static constexpr size_t bytes(bool keep) { return (ring_elems + (keep ? 2 : 1) * img_elems) * sizeof(T); }
    static_assert(C % 32 == 0 && HD % 16 == 0 && CT % HD == 0 && CT % CW == 0 && NW <= 16, "shape");
};

unsigned long long* fused_debug_buffer();     // spectral_fused.hip (mphsir_debug)

__device__ __forceinline__ float dpp_row_shr1(float old, float src) {     // lane i <- lane i-1 of its 16-lane row; lane 0 keeps old
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shl1(float old, float src) {     // lane i <- lane i+1; lane 15 keeps old
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x101, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {     // ds_bpermute_b32: the value of lane byte_addr / 4
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ float pick4(f32x4 v, int k) { return k == 0 ? v[0] : (k == 1 ? v[1] : (k == 2 ? v[2] : v[3])); }

// DBG builds only (mphsir_debug armed): shader-clock stamps of workgroup 0 / wave 0 at walk step 9
#define RW_MARK(k) do { if (DBG && blockIdx.x == 0 && tid == 0 && i == 9) a.dbg[k] = __builtin_amdgcn_s_memtime(); } while (0)

// LN (round 5): the LayerNorm prologue of `Attention` (net/MP_HSIR.py:476 with its norm1, :289-322).  A row in the ring is normalised IN
// PLACE, one lane per 16-byte chunk (C / 8 lanes per pixel: a 16- or 8-lane DPP/shuffle group holds a pixel's statistics), one walk
// step after it has landed and one step before the matrix cores first read it -- the ring then keeps one row fewer in flight
// (LEAD 3).  Same statistics as every other LayerNorm of the library: fp32, two passes, biased variance, eps 1e-5.
template <class T, int C, int HD, bool KEEP, bool DBG, bool LN>
__global__ __launch_bounds__(64 * rw_waves<HD>(), (rw_waves<HD>() + 3) / 4) void qkv_dwconv_gram_rows_kernel(RowGramDev a) {
    typedef ElemTraits<T> TR;
    typedef RwCfg<T, C, HD> CF;
    typedef typename TR::frag_t frag_t;
    constexpr int NBW = CF::NBW, CW = CF::CW, CT = CF::CT, HPG = CF::HPG, NKC = CF::NKC, XPE = CF::XPE, OPE = CF::OPE, WPT = CF::WPT, NW = CF::NW;
    constexpr int CH1 = CF::CH1, NCH = CF::NCH, NDQ = CF::NDQ, NT = CF::NT, NTW = CF::NTW, TPW = CF::TPW, THREADS = CF::THREADS;
    constexpr int HEADS = C / HD, SLOTE = NCH * 8;          // elements per ring slot
    constexpr int LEAD = LN ? RW_LEAD + 1 : RW_LEAD;         // walk steps between "landed" and "first read by the matrix cores"
    constexpr int RING = CF::RING;
    static_assert(RING > LEAD && (RING & (RING - 1)) == 0, "ring");
    constexpr int IMG = RW_SW * OPE;                         // elements per row image
    static_assert(sizeof(T) == 2, "16-bit types only");
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* outs = reinterpret_cast<T*>(smem_v);                 // [2][32 rows][OPE]   q | k | v of one output row; row = b*16 + j
    T* tims = outs + CF::img_elems;                          // [2][32 rows][OPE]   KEEP: t = qkv(x) of one row
    T* ring = tims + (KEEP ? CF::img_elems : 0);             // [8 slots][34 pixels][XPE]   x rows, slot = row % 8

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform(), j = lane & 15, g = lane >> 4;
    const int hgroups = HEADS / HPG, strips = a.W / RW_SW;
    const int blk = (gridDim.x & 7) == 0 ? (int)xcd_contiguous_block() : (int)blockIdx.x;
    const int hg = blk % hgroups, seg = (blk / hgroups) % a.nseg, strip = (blk / (hgroups * a.nseg)) % strips, b = blk / (hgroups * a.nseg * strips);
    const int RS = a.H / a.nseg, y0 = seg * RS, x0 = strip * RW_SW, h0 = hg * HPG;
    const long img = (long)b * a.H * a.W;
    const int type = wv / WPT, part = wv % WPT;             // wave -> q|k|v, which CW channels of the group's CT
    const int cl0 = type * CT + part * CW;                   // first local channel (row image column) of this wave
    const int cg0 = type * C + h0 * HD + part * CW;          // first global channel (row of Wqkv, column of w9 / T)
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* Wg = reinterpret_cast<const T*>(a.Wqkv);

    // ---- LDS-DMA of ring rows: instruction d = wv + NW*q of a row covers chunks 64d .. 64d+63 of the slot (linear image:
    // destination = wave-uniform base + lane * 16 B); chunk n = (ring pixel n / CH1, 16-byte column n % CH1), the pad
    // column re-reads the last real one.  Columns outside the image are clamped (their t is zeroed where it is used).  The
    // source is a scalar row pointer that walks down the image (clamped at its first / last row: rows outside it are loaded
    // but never used) plus a per-lane byte offset fixed for the whole walk: no vector arithmetic per row.
    unsigned dma_off[NDQ];
    bool dma_on[NDQ];
#pragma unroll
    for (int q = 0; q < NDQ; ++q) {
        const int n = 64 * (wv + NW * q) + lane;
        dma_on[q] = n < NCH;
        const int rp = (dma_on[q] ? n : 0) / CH1, cv = (dma_on[q] ? n : 0) % CH1;
        int xx = x0 - 1 + rp;
        xx = xx < 0 ? 0 : (xx >= a.W ? a.W - 1 : xx);
        dma_off[q] = (unsigned)(((long)xx * a.ldx + (cv < C / 8 ? cv : C / 8 - 1) * 8) * (long)sizeof(T));
    }
    int dty = y0 - 1;                                        // image row (unclamped) of the next row to request
    const T* drow = X + (img + (long)(dty < 0 ? 0 : dty) * a.W) * a.ldx;
    const long drow_step = (long)a.W * a.ldx;
    auto dma_next = [&](int slot) __attribute__((always_inline)) {
        T* dst = ring + (size_t)slot * SLOTE;
#pragma unroll
        for (int qq = 0; qq < NDQ; ++qq) {
            if (dma_on[qq]) MPHSIR_LDS_DMA16(drow, dma_off[qq], dst + (size_t)(wv + NW * qq) * 512);
        }
        ++dty;
        if (dty >= 1 && dty <= a.H - 1) drow += drow_step;
    };
#pragma unroll
    for (int q = 0; q < RING; ++q) dma_next(q);

    // ---- per-wave constants: 1x1 weights as MFMA A fragments (rows = channels), depthwise taps of the lane's channels
    frag_t Wf[NBW][NKC];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
            Wf[nb][kc] = *reinterpret_cast<const frag_t*>(Wg + (long)(cg0 + 16 * nb + j) * C + 32 * kc + 8 * g);
    float wt[NBW][4][9];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 9; ++t) wt[nb][r][t] = a.w9[t * a.ldw + cg0 + 16 * nb + 4 * g + r];

    // ---- running state
    f32x4 acc[2][NBW];                     // t of the walk row the depthwise taps run on in the next step
    f32x4 Pa[2][NBW], Pb[2][NBW];          // partial outputs of image rows ty-1 (taps 0..5 done) and ty (taps 0..2 done)
    f32x4 eacc[NBW];                       // t of the strip's outer columns for the row in acc: lanes 0..7 left, 8..15 right column
    f32x4 gacc[TPW];                       // this wave's Gram tiles (head, 16 q channels ti, 16 k channels tj)
    f32x4 sqacc[TPW], skacc[TPW];          // q_ti q_ti^T (tiles with tj == 0) and k_tj k_tj^T (ti == 0): their diagonals are the sums
                                           // of squares of F.normalize -- on the matrix cores, which idle, instead of 12 VALU per step
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[bb][nb] = Pa[bb][nb] = Pb[bb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) eacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < TPW; ++s) gacc[s] = sqacc[s] = skacc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* xlane = ring + (2 * j + 1) * XPE + 8 * g;      // pixel 2j of the strip = ring pixel 2j+1
    const int eside = j >> 3;                                // edge block: lanes 0..7 carry the left outer column, 8..15 the right
    const bool ex_ok = eside ? (x0 + RW_SW < a.W) : (x0 > 0);      // one (zero padding of t at the image border otherwise)
    const T* elane = ring + (eside ? (RW_RPX - 1) * XPE : 0) + 8 * g;
    T* wlane = outs + cl0 + 4 * g + j * OPE;                 // this lane's 4 channels of pixel 2j in the row image (block 1: + 16 rows)
    constexpr int CPR = CT / 8;                              // 16-byte chunks per pixel and q | k | v

    // ---- the store chunks of this thread (16 bytes each): offset in the row image, element offset inside the image row of
    // the destination; the destination row is a scalar pointer that moves down one image row per step (set so that step i
    // finds output row i-3 / t row i-1)
    constexpr int NVC = (RW_SW * CPR + THREADS - 1) / THREADS, NQC = (RW_SW * 2 * CPR + THREADS - 1) / THREADS, NTC = (RW_SW * 3 * CPR + THREADS - 1) / THREADS;
    int vl[NVC], ql[KEEP ? NQC : 1], tl[KEEP ? NTC : 1], vo[NVC], qo[KEEP ? NQC : 1], to[KEEP ? NTC : 1];
#pragma unroll
    for (int c = 0; c < NVC; ++c) {
        const int idx = (c * THREADS + tid) % (RW_SW * CPR), rho = idx / CPR, c8 = idx % CPR, p = 2 * (rho & 15) + (rho >> 4);
        vl[c] = rho * OPE + 2 * CT + 8 * c8;
        vo[c] = (int)(p * a.ldvo) + 8 * c8;
    }
    if (KEEP) {
#pragma unroll
        for (int c = 0; c < NQC; ++c) {
            const int idx = (c * THREADS + tid) % (RW_SW * 2 * CPR), rho = idx / (2 * CPR), rem = idx % (2 * CPR), p = 2 * (rho & 15) + (rho >> 4);
            ql[c] = rho * OPE + 8 * rem;
            qo[c] = (int)(p * a.ldqk) + (rem / CPR) * C + 8 * (rem % CPR);
        }
#pragma unroll
        for (int c = 0; c < NTC; ++c) {
            const int idx = (c * THREADS + tid) % (RW_SW * 3 * CPR), rho = idx / (3 * CPR), rem = idx % (3 * CPR), p = 2 * (rho & 15) + (rho >> 4);
            tl[c] = rho * OPE + 8 * rem;
            to[c] = (int)(p * a.ldt) + (rem / CPR) * C + 8 * (rem % CPR);
        }
    }
    T* vrow = reinterpret_cast<T*>(a.V) + (img + (long)(y0 - 4) * a.W + x0) * a.ldvo + h0 * HD;
    T* qrow = KEEP ? reinterpret_cast<T*>(a.QKout) + (img + (long)(y0 - 4) * a.W + x0) * a.ldqk + h0 * HD : nullptr;
    T* trow_g = KEEP ? reinterpret_cast<T*>(a.Tout) + (img + (long)(y0 - 3) * a.W + x0) * a.ldt + h0 * HD : nullptr;
    const long vstep = (long)a.W * a.ldvo, qstep = (long)a.W * a.ldqk, tstep = (long)a.W * a.ldt;

    wait_vmcnt<0>();
    lds_barrier();

    // ---- LayerNorm of one ring slot in place (LN builds): lane = one 16-byte chunk of one of the 34 ring pixels
    constexpr int CPP = C / 8;                               // chunks per pixel = lanes per pixel (8 or 16: one shuffle group)
    static_assert(!LN || ((CPP == 8 || CPP == 16 || CPP == 32) && THREADS % CPP == 0), "LayerNorm prologue: C = 64, 128 or 256");
    auto ln_row = [&](int slot) __attribute__((always_inline)) {
        if constexpr (LN) {
            for (int base = 0; base + (tid & ~63) < RW_RPX * CPP; base += THREADS) {      // whole WAVES run the body (the shuffles need every lane
                const bool valid = base + tid < RW_RPX * CPP;                               // of a group): lanes beyond the row recompute chunk 0
                const int idx = valid ? base + tid : (tid % CPP);
                T* p = ring + (size_t)slot * SLOTE + (idx / CPP) * XPE + (idx % CPP) * 8;
                const Vec16<T> v = load16<T>(p);
                float x[8], sum = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[e] = v.get(e); sum += x[e]; }
#pragma unroll
                for (int m = 1; m < CPP; m <<= 1) sum += __shfl_xor(sum, m);
                const float mean = sum / (float)C;
                float d2 = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[e] -= mean; d2 += x[e] * x[e]; }
#pragma unroll
                for (int m = 1; m < CPP; m <<= 1) d2 += __shfl_xor(d2, m);
                const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
                // weight / bias of the lane's 8 channels: re-read per row (L1 hits) rather than 16 registers held across the whole walk
                const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.ln_w + (idx % CPP) * 8), w1 = *reinterpret_cast<const f32x4*>(a.ln_w + (idx % CPP) * 8 + 4);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.ln_b + (idx % CPP) * 8), b1 = *reinterpret_cast<const f32x4*>(a.ln_b + (idx % CPP) * 8 + 4);
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o.set(e, x[e] * rstd * w0[e] + b0[e]); o.set(4 + e, x[4 + e] * rstd * w1[e] + b1[e]); }
                if (valid) store16<T>(p, o);
            }
        }
    };
    if constexpr (LN) {              // walk row 0 is read by the matrix cores in the first step (i = -1); row i + 2 is normalised in step i
        ln_row(0);
        lds_barrier();
    }

    // One walk step.  STEADY: every part runs and the step's position in the 8-slot ring (U = i % 8) is a compile-time constant,
    // so ring slots, row-image parity and the edge-block phase are immediate offsets / static branches; otherwise the parts
    // are switched by (uniform) flags for the first and last steps.  Walk row i is image row y0 - 1 + i; output row o (image
    // row y0 + o) is complete after the taps of walk row o + 2.
    auto step = [&](int i, auto steady_c, auto u_c) __attribute__((always_inline)) {
        constexpr bool STEADY = decltype(steady_c)::value;
        // ring position of the step: the low bits are a compile-time constant in the steady groups (row-image parity, edge-block
        // phase, half of the slot offsets become immediates), the high bit comes from a scalar
        const int u = STEADY ? (((i & 7) & ~(RW_UNR - 1)) | decltype(u_c)::value) : (i & 7);
        const bool do_mma = STEADY || (i + 1 <= RS + 1 && y0 + i >= 0 && y0 + i < a.H);       // walk row i+1 inside the image
        const bool do_dw = STEADY || (i >= 0 && i <= RS + 1);
        const bool do_out = STEADY || (i >= 2 && i <= RS + 1);                                   // output row i-2 -> row image
        const bool do_gram = STEADY || (i >= 3 && i <= RS + 2);                                  // Gram / stores of output row i-3
        const bool do_tw = KEEP && (STEADY || (i >= 1 && i <= RS));                              // t row i -> t image
        const bool do_ts = KEEP && (STEADY || (i >= 2 && i <= RS + 1));                          // t row i-1 -> HBM
        RW_MARK(0);
        if (STEADY || i >= 0) dma_next(u & (RING - 1));     // slot i % RING is free: row i was last read in step i-1
        if constexpr (LN) ln_row((u + 2) & (RING - 1));  // walk row i + 2: landed before this step began (LEAD 3), first read in step i + 1

        // ---- LDS reads whose results the end of the step needs, issued first so that their latency hides behind the taps:
        // the Gram operands and the store chunks of output row i-3 (row image written in step i-1), training: the chunks of
        // t row i-1.  (The x fragments of walk row i+1 are read after the taps: 8 registers per K chunk held across them is
        // what pushes the wide shapes over the 168 registers three waves per SIMD leave.)
        const T* orow = outs + ((u - 1) & 1) * IMG;
        const T* trow_r = tims + ((u - 1) & 1) * IMG;
        frag_t gq[TPW], gk[TPW];
        Vec16<T> vch[NVC], qch[KEEP ? NQC : 1], tch[KEEP ? NTC : 1];
        if (do_gram) {
#pragma unroll
            for (int s = 0; s < TPW; ++s) {
                const int t = wv + NW * s;                   // wave-uniform
                if (t < NTW) {
                    const int hh = t / (NT * NT), ti = (t / NT) % NT, tj = t % NT;
                    gq[s] = load_frag_tr<T>(orow, OPE, hh * HD + 16 * ti, 0);
                    gk[s] = load_frag_tr<T>(orow, OPE, CT + hh * HD + 16 * tj, 0);
                }
            }
#pragma unroll
            for (int c = 0; c < NVC; ++c)
                if (c * THREADS + tid < RW_SW * CPR) vch[c] = load16<T>(orow + vl[c]);
            if (KEEP) {
#pragma unroll
                for (int c = 0; c < NQC; ++c)
                    if (c * THREADS + tid < RW_SW * 2 * CPR) qch[c] = load16<T>(orow + ql[c]);
            }
        }
        if (do_ts) {
#pragma unroll
            for (int c = 0; c < NTC; ++c)
                if (c * THREADS + tid < RW_SW * 3 * CPR) tch[c] = load16<T>(trow_r + tl[c]);
        }
        RW_MARK(1);

        // ---- depthwise 3x3 of walk row i in registers.  Neighbours of pixel 2j (block 0): left = block 1 of lane j-1, right =
        // block 1 of this lane; of pixel 2j+1 (block 1): left = block 0 of this lane, right = block 0 of lane j+1.  Lane 0's
        // left and lane 15's right neighbour are the strip's outer columns: the edge block of the previous step left their t
        // in exactly those lanes (the `old` operand of the DPP move: what a lane without a source keeps).
        if (do_dw) {
            f32x4 fin[2][NBW];
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                f32x4 L0, R1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    L0[r] = dpp_row_shr1(eacc[nb][r], acc[1][nb][r]);
                    R1[r] = dpp_row_shl1(eacc[nb][r], acc[0][nb][r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float* w = wt[nb][r];
                    const float l0 = L0[r], m0 = acc[0][nb][r], r0 = acc[1][nb][r];        // block 0: left, centre, right
                    const float l1 = acc[0][nb][r], m1 = acc[1][nb][r], r1 = R1[r];        // block 1
                    // tap order 0..8 per output, as a chain (bitwise the order of the tile form)
                    float pc0 = l0 * w[0];   pc0 = fmaf(m0, w[1], pc0);   pc0 = fmaf(r0, w[2], pc0);
                    float pc1 = l1 * w[0];   pc1 = fmaf(m1, w[1], pc1);   pc1 = fmaf(r1, w[2], pc1);
                    float pb0 = fmaf(l0, w[3], Pb[0][nb][r]);   pb0 = fmaf(m0, w[4], pb0);   pb0 = fmaf(r0, w[5], pb0);
                    float pb1 = fmaf(l1, w[3], Pb[1][nb][r]);   pb1 = fmaf(m1, w[4], pb1);   pb1 = fmaf(r1, w[5], pb1);
                    float pa0 = fmaf(l0, w[6], Pa[0][nb][r]);   pa0 = fmaf(m0, w[7], pa0);   pa0 = fmaf(r0, w[8], pa0);
                    float pa1 = fmaf(l1, w[6], Pa[1][nb][r]);   pa1 = fmaf(m1, w[7], pa1);   pa1 = fmaf(r1, w[8], pa1);
                    fin[0][nb][r] = pa0;   fin[1][nb][r] = pa1;          // image row y0 + i - 2 complete
                    Pa[0][nb][r] = pb0;    Pa[1][nb][r] = pb1;           // row y0 + i - 1: taps 0..5
                    Pb[0][nb][r] = pc0;    Pb[1][nb][r] = pc1;           // row y0 + i: taps 0..2
                }
            }
            RW_MARK(2);
            if (do_out) {
                T* wrow = wlane + (u & 1) * IMG;
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) store4<T>(wrow + bb * 16 * OPE + 16 * nb, fin[bb][nb]);
            }
            if (do_tw) {
                T* trow = wlane + (tims - outs) + (u & 1) * IMG;
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int nb = 0; nb < NBW; ++nb) store4<T>(trow + bb * 16 * OPE + 16 * nb, acc[bb][nb]);
            }
        }
        RW_MARK(3);

        // ---- the matrix work of the step: the Gram of output row i-3 (one K chunk of 32 pixels per tile) and the sums of
        // squares (diagonals of q q^T / k k^T) on operands that arrived long ago, the stores, then t of walk row i+1 (used in
        // the NEXT step: nothing waits for it here)
        if (do_gram) {
#pragma unroll
            for (int s = 0; s < TPW; ++s)
                if (wv + NW * s < NTW) {
                    const int t = wv + NW * s;
                    mma(gacc[s], gq[s], gk[s]);
                    if (t % NT == 0) mma(sqacc[s], gq[s], gq[s]);
                    if ((t / NT) % NT == 0) mma(skacc[s], gk[s], gk[s]);
                }
            // v (training: q | k too) to HBM as 16-byte chunks, >= 96 contiguous bytes per pixel
#pragma unroll
            for (int c = 0; c < NVC; ++c)
                if (c * THREADS + tid < RW_SW * CPR) store16<T>(vrow + vo[c], vch[c]);
            if (KEEP) {
#pragma unroll
                for (int c = 0; c < NQC; ++c)
                    if (c * THREADS + tid < RW_SW * 2 * CPR) store16<T>(qrow + qo[c], qch[c]);
            }
        }
        if (do_ts) {
#pragma unroll
            for (int c = 0; c < NTC; ++c)
                if (c * THREADS + tid < RW_SW * 3 * CPR) store16<T>(trow_g + to[c], tch[c]);
        }
        vrow += vstep;
        if (KEEP) { qrow += qstep; trow_g += tstep; }
        f32x4 nacc[2][NBW], neacc[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) nacc[0][nb] = nacc[1][nb] = neacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (do_mma) {
            typedef __attribute__((ext_vector_type(4))) int i32x4;
            const T* xrow = xlane + ((u + 1) & (RING - 1)) * SLOTE;
            const T* erow = elane + ((u + 1) & (RING - 1)) * SLOTE;
            const int emask = ex_ok ? -1 : 0;                // the outer column lies outside the image: t = 0 there
            // two K chunks (6 fragments, 24 registers) at a time: all of a wide row's fragments in flight at once is what tips
            // C = 128 over the 168 registers three waves per SIMD leave
#pragma unroll
            for (int k2 = 0; k2 < NKC; k2 += 2) {
                frag_t xa[2], xb[2], xe[2];
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (k2 + q < NKC) {
                        xa[q] = *reinterpret_cast<const frag_t*>(xrow + 32 * (k2 + q));
                        xb[q] = *reinterpret_cast<const frag_t*>(xrow + XPE + 32 * (k2 + q));
                        xe[q] = __builtin_bit_cast(frag_t, *reinterpret_cast<const i32x4*>(erow + 32 * (k2 + q)) & emask);
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (k2 + q < NKC) {
#pragma unroll
                        for (int nb = 0; nb < NBW; ++nb) {
                            mma(nacc[0][nb], Wf[nb][k2 + q], xa[q]);
                            mma(nacc[1][nb], Wf[nb][k2 + q], xb[q]);
                            mma(neacc[nb], Wf[nb][k2 + q], xe[q]);    // the outer columns of the same row (a third of an MFMA block
                        }                                             // used: the matrix cores are ~15 % busy, a ds_bpermute per value
                    }                                                 // is not free)
                if (k2 + 2 < NKC) asm volatile("" ::: "memory");      // the next pair's reads stay behind this pair's
            }
        }
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[bb][nb] = nacc[bb][nb];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) eacc[nb] = neacc[nb];

        // ---- the row DMA-ed RW_LEAD steps ago must have landed before the barrier: it is read from step i+1 on.  The DMAs of
        // the steps since are younger and stay in flight (vmcnt counts in issue order; this wave's stores only make the wait
        // reach a little further than it has to)
        if (64 * wv < NCH) {                                 // wave-uniform: this wave issues 1 .. NDQ DMA instructions per row
            if (NDQ > 1 && 64 * (wv + NW * (NDQ - 1)) < NCH) wait_vmcnt<(RING - LEAD) * NDQ>();
            else wait_vmcnt<(RING - LEAD) * (NDQ > 1 ? NDQ - 1 : 1)>();
        }
        RW_MARK(4);
        lds_barrier();
        RW_MARK(5);
    };
    auto tail = [&](int i) __attribute__((always_inline)) { step(i, std::false_type{}, std::integral_constant<int, 0>{}); };

    // first steps (parts missing), then whole turns of the ring with everything static, then the last steps
    int i = -1;
#pragma unroll 1
    for (; i < (3 + RW_UNR - 1) / RW_UNR * RW_UNR && i <= RS + 2; ++i) tail(i);     // steady steps start at i >= 3, i % RW_UNR == 0
#pragma unroll 1
    for (; i + RW_UNR - 1 <= RS - 1; i += RW_UNR) {
        step(i + 0, std::true_type{}, std::integral_constant<int, 0>{});
        if (RW_UNR > 1) step(i + 1, std::true_type{}, std::integral_constant<int, 1>{});
        if (RW_UNR > 2) {
            step(i + 2, std::true_type{}, std::integral_constant<int, 2>{});
            step(i + 3, std::true_type{}, std::integral_constant<int, 3>{});
        }
    }
#pragma unroll 1
    for (; i <= RS + 2; ++i) tail(i);
    if (DBG && blockIdx.x == 0 && tid == 0) a.dbg[9] = __builtin_amdgcn_s_memtime();

    // ---- the workgroup's partials: slot (sample, strip, segment); head groups write disjoint heads / channels of it
    const int nsplit = strips * a.nseg, slot = strip * a.nseg + seg;
    float* Gp = a.Gpart + ((long)b * nsplit + slot) * HEADS * HD * HD;
    float* Sp = a.Spart + ((long)b * nsplit + slot) * 2 * C;
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int t = wv + NW * s;
        if (t < NTW) {
            const int hh = t / (NT * NT), ti = (t / NT) % NT, tj = t % NT;
#pragma unroll
            for (int r = 0; r < 4; ++r) Gp[((h0 + hh) * HD + 16 * ti + 4 * g + r) * HD + 16 * tj + j] = gacc[s][r];
        }
    }
#pragma unroll
    for (int s = 0; s < TPW; ++s) {          // diagonals: lane (j, g) holds D[4g + r][j], so row == column where j >> 2 == g, r = j & 3
        const int t = wv + NW * s;
        if (t < NTW && (j >> 2) == g) {
            const int hh = t / (NT * NT), ti = (t / NT) % NT, tj = t % NT;
            if (tj == 0) Sp[(h0 + hh) * HD + 16 * ti + j] = pick4(sqacc[s], j & 3);
            if (ti == 0) Sp[C + (h0 + hh) * HD + 16 * tj + j] = pick4(skacc[s], j & 3);
        }
    }
}

template <class T, int C, int HD>
static int launch_rows(const RowGramDev& d, hipStream_t s) {
    typedef RwCfg<T, C, HD> CF;
    const int grid = d.B * (d.W / RW_SW) * d.nseg * ((C / HD) / CF::HPG);
#define MPHSIR_RW_GO(keep, dbg, ln)                                                                                                    \
    do {                                                                                                                               \
        const size_t shmem = CF::bytes(keep);                                                                                          \
        allow_big_lds(qkv_dwconv_gram_rows_kernel<T, C, HD, keep, dbg, ln>, shmem);                                                    \
        MPHSIR_LAUNCH(MPHSIR_K_QKV_DWCONV_GRAM, (qkv_dwconv_gram_rows_kernel<T, C, HD, keep, dbg, ln>), dim3(grid), dim3(CF::THREADS), \
                      shmem, s, d);                                                                                                    \
    } while (0)
    if constexpr ((C == 64 || C == 128 || C == 256) && HD == 32) {       // the LayerNorm-ed `Attention` of PromptFusion: 32-wide heads
        if (d.ln_w) {
            if (d.Tout) MPHSIR_RW_GO(true, false, true);
            else MPHSIR_RW_GO(false, false, true);
            return MPHSIR_OK;
        }
    }
    if (d.dbg && !d.Tout && std::is_same<T, bf16_t>::value && C <= 128 && HD <= 64) MPHSIR_RW_GO(false, true, false);   // stamped build
    else if (d.Tout) MPHSIR_RW_GO(true, false, false);
    else MPHSIR_RW_GO(false, false, false);
#undef MPHSIR_RW_GO
    return MPHSIR_OK;
}

// shapes: widths whose 1x1 weights fit the wave's registers (C <= 192); heads of 32 / 64 / 48 channels (96-wide heads would need
// 32 channels per wave: over the 168 registers that three waves per SIMD leave); the group of heads one workgroup takes (64
// channels; 48-wide heads: one head) must divide the head count
static bool rows_shape(int C, int HD) {
    return (HD == 32 && (C == 64 || C == 128 || C == 256)) || (HD == 64 && (C == 64 || C == 128)) || (HD == 48 && (C == 96 || C == 192));
}
template <class T> struct RowShapes {
    static int run(const RowGramDev& d, int C, int HD, hipStream_t s) {
#define MPHSIR_RW_CASE(c, hd) if (C == c && HD == hd) return launch_rows<T, c, hd>(d, s);
        MPHSIR_RW_CASE(64, 32) MPHSIR_RW_CASE(128, 32) MPHSIR_RW_CASE(64, 64) MPHSIR_RW_CASE(128, 64)
        MPHSIR_RW_CASE(96, 48) MPHSIR_RW_CASE(192, 48) MPHSIR_RW_CASE(256, 32)
#undef MPHSIR_RW_CASE
        return MPHSIR_EINVAL;
    }
};
template <> struct RowShapes<float> {
    static int run(const RowGramDev&, int, int, hipStream_t) { return MPHSIR_EINVAL; }
};

int rows_form_fits(int C, int heads, int H, int W, int dtype, int ln) {
    if (dtype == MPHSIR_F32 || heads <= 0 || C % heads != 0 || H < 4 || W <= 0 || W % RW_SW != 0) return 0;
    if (ln && !((C == 64 || C == 128 || C == 256) && C / heads == 32)) return 0;      // the LayerNorm prologue: a pixel's C / 8 chunks = one 8- or 16-lane shuffle group; 32-wide heads
    const int HD = C / heads;
    if (!rows_shape(C, HD)) return 0;
    const int hpg = HD == 32 ? 2 : 1;
    return heads % hpg == 0 ? 1 : 0;
}

int rows_form_launch(const mphsir_fused_gram_args* a, int dtype, hipStream_t s) {
    RowGramDev d{a->X, (long)a->ldx, a->ln_w, a->ln_b, a->Wqkv, a->w9, (long)a->ldw, a->V, (long)a->ldvo, a->Gpart, a->Spart,
                 a->B, a->H, a->W, a->row_segments, a->T, (long)a->ldt, a->QK, (long)a->ldqk, fused_debug_buffer()};
    return MPHSIR_DISPATCH_T(dtype, (RowShapes<T_>::run(d, a->C, a->C / a->heads, s)));
}

}  // namespace mphsir

extern "C" int mphsir_qkv_dwconv_gram_rows_fits(int32_t C, int32_t heads, int32_t H, int32_t W, int dtype, int32_t with_ln) {
    if (!MPHSIR_DTYPE_OK(dtype)) return 0;
    return mphsir::rows_form_fits(C, heads, H, W, dtype, with_ln);
}
