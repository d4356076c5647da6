// qkv_dwconv_gram, ROW-WALKING form: the fused pass A of the global spectral (channel) attention with the 1x1-conv output
// kept in REGISTERS between the matrix cores and the depthwise 3x3.
//
// Reference: Spectral_Attention.forward net/MP_HSIR.py:96-107 (== Attention :301-313):
//     qkv = qkv_dwconv(qkv(x));  q,k,v = chunk(3);  q,k -> F.normalize over pixels;  attn = q @ k^T
//
// The tile form (spectral_fused.hip) sends t = qkv(x) through an fp32 LDS tile between its MFMA phase and its depthwise
// phase, barrier-separated in one workgroup per CU, recomputes a 1.5x halo and re-stages the weights per channel slab.
// Here a workgroup owns a 32-pixel-wide column strip of the image and walks DOWN it, one image row per step:
//   * a wave owns 32 output channels (two 16-channel MFMA row blocks) of q, k or v for the whole walk: its 1x1 weights
//     (A fragments) and its 9 taps per channel are loaded ONCE and stay in registers;
//   * per row the 32 pixels are two interleaved 16-pixel MFMA column blocks (block b, lane j = pixel 2j+b).  The
//     accumulators (lane = 4 channels x 1 pixel) ARE the depthwise input: the row above / below is another register
//     (three partial output rows are kept: every t row is multiplied into the rows it touches and then dropped, so no
//     row is ever computed twice), the pixel to the left / right is either the other block's register of the SAME lane
//     or one DPP row shift away.  The strip's outer columns (x0-1, x0+32) come from an "edge" MFMA block computed once
//     per 8 rows (16 pixels = 8 rows x 2 sides, 1/16 extra work) and are dealt to lanes 0 / 15 with one ds_bpermute;
//   * the x rows (optionally LayerNorm-ed) are staged in a two-slot LDS ring by all waves; q,k,v (training: t too) leave
//     through an LDS row image [pixel][channel] so that the Gram reads them pixel-strided (ds_read_b64_tr_b16) and the
//     global stores are whole 16-byte row chunks of >= 128 contiguous bytes per pixel;
//   * Gram tiles and sums of squares are PERSISTENT per-wave accumulators over the whole walk (fixed order, no atomics):
//     one store per workgroup into its Gpart / Spart slot at the end.
// One workgroup barrier per row.  Six waves: q|k|v x two 32-channel halves = 64 channels of each (two heads of 32 or one
// of 64; 48-wide heads: three 16-channel blocks per wave); wider nets split their heads over several workgroups.
#include <stdlib.h>

#include <type_traits>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

constexpr int RW_SW = 32;                         // strip width in pixels
constexpr int RW_WAVES = 6, RW_THREADS = RW_WAVES * 64;

struct RowGramDev {
    const void* X; long ldx;
    const float* ln_w; const float* ln_b;
    const void* Wqkv;                            // [3C][C]
    const float* w9; long ldw;                   // [9][ldw]: taps of q | k | v channels
    void* V; long ldvo;
    float* Gpart; float* Spart;
    int B, H, W, nseg;                           // nseg row segments per strip: (W / 32) * nseg partial slots per sample
    void* Tout; long ldt;
    void* QKout; long ldqk;
    unsigned long long* dbg;                     // diagnostics (mphsir_fused_debug): shader-clock stamps of workgroup 0, wave 0
};
// stamps of step 9 (no edge block) in dbg[0..8] and of step 8 (edge block) in dbg[16..24]
#define RW_MARK(k) do { if (a.dbg && blockIdx.x == 0 && tid == 0 && (i == 9 || i == 8)) a.dbg[(i == 8 ? 16 : 0) + k] = __builtin_amdgcn_s_memtime(); } while (0)

template <class T, int C, int HD> struct RwCfg {
    static constexpr int NBW = HD % 48 == 0 ? 3 : 2;        // 16-channel MFMA row blocks per wave
    static constexpr int CW = 16 * NBW;                      // channels per wave
    static constexpr int CT = 2 * CW;                        // channels of q (of k, of v) per workgroup
    static constexpr int HPG = CT / HD;                      // heads per workgroup
    static constexpr int NKC = C / 32;                       // K chunks of the 1x1 conv
    static constexpr int XPE = C + 8;                        // ring: elements per pixel row (pitch / 16 B odd: conflict-free
                                                             // ds_read_b128 of the rows 2j+b)
    static constexpr int OPE = 3 * CT + 8;                   // row image [q | k | v] + 16 B
    static constexpr int LPP = (C / 8) % 16 == 0 ? 16 : ((C / 8) % 8 == 0 ? 8 : 4);   // loader lanes per pixel
    static constexpr int CPL = (C / 8) / LPP;                // 16-byte chunks per loader lane
    static constexpr int PPP = RW_THREADS / LPP;             // pixels per loader pass
    static constexpr int NPASS = (RW_SW + PPP - 1) / PPP;
    static constexpr int NT = HD / 16, NTW = HPG * NT * NT, TPW = (NTW + RW_WAVES - 1) / RW_WAVES;
    static constexpr size_t ring_elems = (size_t)2 * RW_SW * XPE, img_elems = (size_t)2 * RW_SW * OPE;
    static constexpr size_t bytes(bool keep) { return (ring_elems + (keep ? 2 : 1) * img_elems) * sizeof(T); }
    static_assert(C % 32 == 0 && HD % 16 == 0 && CT % HD == 0 && (C / 8) % LPP == 0 && RW_THREADS % LPP == 0, "shape");
};

unsigned long long* fused_debug_buffer();     // spectral_fused.hip (mphsir_fused_debug)

__device__ __forceinline__ float dpp_row_shr1(float old, float src) {     // lane i <- lane i-1 of its 16-lane row; lane 0 keeps old
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_row_shl1(float old, float src) {     // lane i <- lane i+1; lane 15 keeps old
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src), 0x101, 0xf, 0xf, false));
}

__device__ __forceinline__ float lane_fetch(int byte_addr, float v) {     // ds_bpermute_b32: the value of lane byte_addr / 4
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}

template <class T, int C, int HD, bool LN, bool KEEP>
__global__ __launch_bounds__(RW_THREADS, 2) void qkv_dwconv_gram_rows_kernel(RowGramDev a) {
    typedef ElemTraits<T> TR;
    typedef RwCfg<T, C, HD> CF;
    typedef typename TR::frag_t frag_t;
    constexpr int NBW = CF::NBW, CW = CF::CW, CT = CF::CT, HPG = CF::HPG, NKC = CF::NKC, XPE = CF::XPE, OPE = CF::OPE;
    constexpr int LPP = CF::LPP, CPL = CF::CPL, PPP = CF::PPP, NPASS = CF::NPASS, NT = CF::NT, NTW = CF::NTW, TPW = CF::TPW;
    constexpr int HEADS = C / HD;
    static_assert(sizeof(T) == 2, "16-bit types only");
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    T* ring = reinterpret_cast<T*>(smem_v);                 // [2][32 pixels][XPE]   x rows (LayerNorm-ed), slot = row & 1
    T* outs = ring + CF::ring_elems;                         // [2][32 rows ][OPE]   q | k | v of one output row; row = b*16 + j
    T* tims = outs + CF::img_elems;                          // [2][32 rows ][OPE]   KEEP: t = qkv(x) of one row

    const int tid = threadIdx.x, lane = tid & 63, wv = wave_id_uniform(), j = lane & 15, g = lane >> 4;
    const int hgroups = HEADS / HPG, strips = a.W / RW_SW;
    const int blk = (gridDim.x & 7) == 0 ? (int)xcd_contiguous_block() : (int)blockIdx.x;
    const int hg = blk % hgroups, seg = (blk / hgroups) % a.nseg, strip = (blk / (hgroups * a.nseg)) % strips, b = blk / (hgroups * a.nseg * strips);
    const int RS = a.H / a.nseg, y0 = seg * RS, x0 = strip * RW_SW, h0 = hg * HPG;
    const long img = (long)b * a.H * a.W;
    const int type = wv >> 1, half = wv & 1;                // wave -> q|k|v, lower / upper CW channels of the group's CT
    const int cl0 = type * CT + half * CW;                   // first local channel (row image column) of this wave
    const int cg0 = type * C + h0 * HD + half * CW;          // first global channel (row of Wqkv, column of w9 / T)
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* Wg = reinterpret_cast<const T*>(a.Wqkv);

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

    // ---- the x-row loader: LPP lanes per pixel, CPL 16-byte chunks each (LayerNorm statistics by xor shuffles inside the
    // lane group); row i of the walk is image row y0 - 1 + i
    const int lpx = tid / LPP, lli = tid % LPP;
    Vec16<T> xr[NPASS][CPL];
    auto xload = [&](int i) __attribute__((always_inline)) {
        const int ty = y0 - 1 + i;
        if (ty < 0 || ty >= a.H) return;                    // uniform: rows outside the image are never read (t = 0 there)
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int px = lpx + ps * PPP;
            const bool on = px < RW_SW;
            const T* src = X + (img + (long)ty * a.W + x0 + (on ? px : 0)) * a.ldx;
#pragma unroll
            for (int c = 0; c < CPL; ++c) xr[ps][c] = load16<T>(src + (lli + LPP * c) * 8);
            if (LN) {
                float s = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    for (int e = 0; e < 8; ++e) s += xr[ps][c].get(e);
#pragma unroll
                for (int m = 1; m < LPP; m <<= 1) s += __shfl_xor(s, m);
                const float mean = s / (float)C;
                float d2 = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    for (int e = 0; e < 8; ++e) { const float d = xr[ps][c].get(e) - mean; d2 += d * d; }
#pragma unroll
                for (int m = 1; m < LPP; m <<= 1) d2 += __shfl_xor(d2, m);
                const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
#pragma unroll
                for (int c = 0; c < CPL; ++c)
                    for (int e = 0; e < 8; ++e) {
                        const int k = (lli + LPP * c) * 8 + e;
                        xr[ps][c].set(e, (xr[ps][c].get(e) - mean) * rstd * a.ln_w[k] + a.ln_b[k]);
                    }
            }
        }
    };
    auto xstore = [&](int i) __attribute__((always_inline)) {
        const int ty = y0 - 1 + i;
        if (ty < 0 || ty >= a.H) return;
        T* dst = ring + (size_t)(i & 1) * RW_SW * XPE;
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int px = lpx + ps * PPP;
            if (px < RW_SW) {
#pragma unroll
                for (int c = 0; c < CPL; ++c) store16<T>(dst + px * XPE + (lli + LPP * c) * 8, xr[ps][c]);
            }
        }
    };

    // ---- running state
    f32x4 Pa[2][NBW], Pb[2][NBW];          // partial outputs of image rows ty-1 (taps 0..5 done) and ty (taps 0..2 done)
    f32x4 eacc[NBW];                       // t of the strip's outer columns for 8 rows: lane j = (row j >> 1, side j & 1)
    f32x4 gacc[TPW];                       // this wave's Gram tiles
    float ss[NBW][4];                      // sums of squares of the lane's q / k channels over its pixels
#pragma unroll
    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) Pa[bb][nb] = Pb[bb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        eacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < 4; ++r) ss[nb][r] = 0.f;
    }
#pragma unroll
    for (int s = 0; s < TPW; ++s) gacc[s] = f32x4{0.f, 0.f, 0.f, 0.f};

    xload(0);
    xstore(0);
    xload(1);
    xstore(1);
    __syncthreads();

    const int nit = RS + 2;
#pragma unroll 1
    for (int i = 0; i < nit; ++i) {
        const int ty = y0 - 1 + i;                           // image row of the t row computed in this step
        const bool in_img = ty >= 0 && ty < a.H;
        RW_MARK(0);
        if (i + 2 < nit) xload(i + 2);                       // into registers; to the ring after this step's barrier

        // ---- every 8 steps: t at the strip's outer columns for the next 8 rows (zero outside the image)
        if ((i & 7) == 0) {
            const int ey = ty + (j >> 1), ex = (j & 1) ? x0 + RW_SW : x0 - 1;
            const bool ev = ey >= 0 && ey < a.H && ex >= 0 && ex < a.W;
            const T* src = X + (img + (long)(ev ? ey : 0) * a.W + (ev ? ex : 0)) * a.ldx + 8 * g;
            frag_t ef[NKC];
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) ef[kc] = *reinterpret_cast<const frag_t*>(src + 32 * kc);
            if (LN) {                   // a pixel's channels are spread over the 4 lanes j, j+16, j+32, j+48
                float s = 0.f;
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc)
                    for (int e = 0; e < 8; ++e) s += to_f32(ef[kc][e]);
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                const float mean = s / (float)C;
                float d2 = 0.f;
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc)
                    for (int e = 0; e < 8; ++e) { const float d = to_f32(ef[kc][e]) - mean; d2 += d * d; }
                d2 += __shfl_xor(d2, 16);
                d2 += __shfl_xor(d2, 32);
                const float rstd = rsqrtf(d2 / (float)C + 1e-5f);
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc)
                    for (int e = 0; e < 8; ++e) {
                        const int k = 32 * kc + 8 * g + e;
                        ef[kc][e] = from_f32<T>((to_f32(ef[kc][e]) - mean) * rstd * a.ln_w[k] + a.ln_b[k]);
                    }
            }
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc)
                for (int e = 0; e < 8; ++e)
                    if (!ev) ef[kc][e] = from_f32<T>(0.f);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                eacc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kc = 0; kc < NKC; ++kc) mma(eacc[nb], Wf[nb][kc], ef[kc]);
            }
        }

        RW_MARK(1);
        // ---- t row: acc[b][nb] = W (channels) x x^T (pixels 2j+b): lane = channels 4g..4g+3 of block nb, pixel 2j+b
        f32x4 acc[2][NBW];
#pragma unroll
        for (int bb = 0; bb < 2; ++bb)
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) acc[bb][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (in_img) {
            const T* xrow = ring + (size_t)(i & 1) * RW_SW * XPE + 8 * g;
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                const frag_t xa = *reinterpret_cast<const frag_t*>(xrow + (2 * j) * XPE + 32 * kc);
                const frag_t xb = *reinterpret_cast<const frag_t*>(xrow + (2 * j + 1) * XPE + 32 * kc);
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    mma(acc[0][nb], Wf[nb][kc], xa);
                    mma(acc[1][nb], Wf[nb][kc], xb);
                }
            }
        }

        RW_MARK(2);
        // ---- depthwise 3x3 in registers.  Neighbours of pixel 2j (block 0): left = block 1 of lane j-1, right = block 1 of
        // this lane; of pixel 2j+1 (block 1): left = block 0 of this lane, right = block 0 of lane j+1.  Lane 0's left and
        // lane 15's right neighbour are the strip's outer columns: one bpermute fetches both from the edge block.
        const int esrc = ((lane & 48) | (2 * (i & 7) + (j == 15 ? 1 : 0))) << 2;
        f32x4 fin[2][NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) {
            f32x4 L0, R1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ed = lane_fetch(esrc, eacc[nb][r]);
                L0[r] = dpp_row_shr1(ed, acc[1][nb][r]);
                R1[r] = dpp_row_shl1(ed, acc[0][nb][r]);
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
                fin[0][nb][r] = pa0;   fin[1][nb][r] = pa1;          // image row ty-1 complete
                Pa[0][nb][r] = pb0;    Pa[1][nb][r] = pb1;           // row ty: taps 0..5
                Pb[0][nb][r] = pc0;    Pb[1][nb][r] = pc1;           // row ty+1: taps 0..2
            }
        }

        RW_MARK(3);
        // ---- output row oy = ty - 1 (i >= 2) and, training, the t row ty (1 <= i <= RS) into the LDS row images
        const bool has_out = i >= 2, has_t = KEEP && i >= 1 && i <= RS;
        if (has_out) {
            T* orow = outs + (size_t)(i & 1) * RW_SW * OPE + cl0 + 4 * g;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    store4<T>(orow + (bb * 16 + j) * OPE + 16 * nb, fin[bb][nb]);
                    if (type < 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float q = to_f32(from_f32<T>(fin[bb][nb][r])); ss[nb][r] = fmaf(q, q, ss[nb][r]); }
                    }
                }
        }
        if (has_t) {
            T* trow = tims + (size_t)(i & 1) * RW_SW * OPE + cl0 + 4 * g;
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) store4<T>(trow + (bb * 16 + j) * OPE + 16 * nb, acc[bb][nb]);
        }
        RW_MARK(4);
        __syncthreads();
        RW_MARK(5);

        if (i + 2 < nit) xstore(i + 2);                      // ring slot i & 1 is free: every wave is past its fragment reads
        RW_MARK(6);

        if (has_out) {
            const T* orow = outs + (size_t)(i & 1) * RW_SW * OPE;
            // Gram of the row's 32 pixels (one K chunk): tile t = (head, 16 q channels, 16 k channels)
#pragma unroll
            for (int s = 0; s < TPW; ++s) {
                const int t = wv + RW_WAVES * s;             // wave-uniform
                if (t < NTW) {
                    const int hh = t / (NT * NT), ti = (t / NT) % NT, tj = t % NT;
                    mma(gacc[s], load_frag_tr<T>(orow, OPE, hh * HD + 16 * ti, 0), load_frag_tr<T>(orow, OPE, CT + hh * HD + 16 * tj, 0));
                }
            }
            RW_MARK(7);
            // v (training: q | k too) to HBM as 16-byte chunks, >= 128 contiguous bytes per pixel
            const long pix0 = img + (long)(ty - 1) * a.W + x0;
            constexpr int CPR = CT / 8;                      // chunks per pixel and type
#pragma unroll
            for (int idx0 = 0; idx0 < RW_SW * CPR; idx0 += RW_THREADS) {
                const int idx = idx0 + tid;
                if (idx < RW_SW * CPR) {
                    const int rho = idx / CPR, c8 = idx % CPR, p = 2 * (rho & 15) + (rho >> 4);
                    store16<T>(reinterpret_cast<T*>(a.V) + (pix0 + p) * a.ldvo + h0 * HD + 8 * c8, load16<T>(orow + rho * OPE + 2 * CT + 8 * c8));
                }
            }
            if (KEEP) {
#pragma unroll
                for (int idx0 = 0; idx0 < RW_SW * 2 * CPR; idx0 += RW_THREADS) {
                    const int idx = idx0 + tid;
                    if (idx < RW_SW * 2 * CPR) {
                        const int rho = idx / (2 * CPR), rem = idx % (2 * CPR), ty2 = rem / CPR, c8 = rem % CPR, p = 2 * (rho & 15) + (rho >> 4);
                        store16<T>(reinterpret_cast<T*>(a.QKout) + (pix0 + p) * a.ldqk + ty2 * C + h0 * HD + 8 * c8,
                                   load16<T>(orow + rho * OPE + ty2 * CT + 8 * c8));
                    }
                }
            }
        }
        RW_MARK(8);
        if (has_t) {
            const T* trow = tims + (size_t)(i & 1) * RW_SW * OPE;
            const long pix0 = img + (long)ty * a.W + x0;
            constexpr int CPR = CT / 8;
#pragma unroll
            for (int idx0 = 0; idx0 < RW_SW * 3 * CPR; idx0 += RW_THREADS) {
                const int idx = idx0 + tid;
                if (idx < RW_SW * 3 * CPR) {
                    const int rho = idx / (3 * CPR), rem = idx % (3 * CPR), ty3 = rem / CPR, c8 = rem % CPR, p = 2 * (rho & 15) + (rho >> 4);
                    store16<T>(reinterpret_cast<T*>(a.Tout) + (pix0 + p) * a.ldt + ty3 * C + h0 * HD + 8 * c8,
                               load16<T>(trow + rho * OPE + ty3 * CT + 8 * c8));
                }
            }
        }
    }

    if (a.dbg && blockIdx.x == 0 && tid == 0) a.dbg[9] = __builtin_amdgcn_s_memtime();
    // ---- the workgroup's partials: slot (sample, strip, segment); head groups write disjoint heads / channels of it
    const int nsplit = strips * a.nseg, slot = strip * a.nseg + seg;
    float* Gp = a.Gpart + ((long)b * nsplit + slot) * HEADS * HD * HD;
    float* Sp = a.Spart + ((long)b * nsplit + slot) * 2 * C;
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int t = wv + RW_WAVES * s;
        if (t < NTW) {
            const int hh = t / (NT * NT), ti = (t / NT) % NT, tj = t % NT;
#pragma unroll
            for (int r = 0; r < 4; ++r) Gp[((h0 + hh) * HD + 16 * ti + 4 * g + r) * HD + 16 * tj + j] = gacc[s][r];
        }
    }
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float s2 = ss[nb][r];
            s2 += __shfl_xor(s2, 1);
            s2 += __shfl_xor(s2, 2);
            s2 += __shfl_xor(s2, 4);
            s2 += __shfl_xor(s2, 8);
            if (type < 2 && j == 0) Sp[type * C + h0 * HD + half * CW + 16 * nb + 4 * g + r] = s2;
        }
}

template <class T, int C, int HD>
static int launch_rows(const RowGramDev& d, hipStream_t s) {
    typedef RwCfg<T, C, HD> CF;
    const int grid = d.B * (d.W / RW_SW) * d.nseg * ((C / HD) / CF::HPG);
#define MPHSIR_RW_GO(ln, keep)                                                                                                   \
    do {                                                                                                                         \
        const size_t shmem = CF::bytes(keep);                                                                                    \
        allow_big_lds(qkv_dwconv_gram_rows_kernel<T, C, HD, ln, keep>, shmem);                                                   \
        MPHSIR_LAUNCH(MPHSIR_K_QKV_DWCONV_GRAM, (qkv_dwconv_gram_rows_kernel<T, C, HD, ln, keep>), dim3(grid), dim3(RW_THREADS), \
                      shmem, s, d);                                                                                              \
    } while (0)
    if (d.Tout) { if (d.ln_w) MPHSIR_RW_GO(true, true); else MPHSIR_RW_GO(false, true); }
    else { if (d.ln_w) MPHSIR_RW_GO(true, false); else MPHSIR_RW_GO(false, false); }
#undef MPHSIR_RW_GO
    return MPHSIR_OK;
}

// shapes: widths whose 1x1 weights fit the wave's registers (C <= 192); heads of 32 / 64 (two 16-channel blocks per wave)
// and 48 / 96 (three); the group of heads one workgroup takes must divide the head count
static bool rows_shape(int C, int HD) {
    return (HD == 32 && (C == 64 || C == 128)) || (HD == 64 && (C == 64 || C == 128)) || (HD == 48 && (C == 96 || C == 192)) ||
           (HD == 96 && (C == 96 || C == 192));
}
template <class T> struct RowShapes {
    static int run(const RowGramDev& d, int C, int HD, hipStream_t s) {
#define MPHSIR_RW_CASE(c, hd) if (C == c && HD == hd) return launch_rows<T, c, hd>(d, s);
        MPHSIR_RW_CASE(64, 32) MPHSIR_RW_CASE(128, 32) MPHSIR_RW_CASE(64, 64) MPHSIR_RW_CASE(128, 64)
        MPHSIR_RW_CASE(96, 48) MPHSIR_RW_CASE(192, 48) MPHSIR_RW_CASE(96, 96) MPHSIR_RW_CASE(192, 96)
#undef MPHSIR_RW_CASE
        return MPHSIR_EINVAL;
    }
};
template <> struct RowShapes<float> {
    static int run(const RowGramDev&, int, int, hipStream_t) { return MPHSIR_EINVAL; }
};

int rows_form_fits(int C, int heads, int H, int W, int dtype) {
    if (dtype == MPHSIR_F32 || heads <= 0 || C % heads != 0 || H <= 0 || W <= 0 || W % RW_SW != 0) return 0;
    const int HD = C / heads;
    if (!rows_shape(C, HD)) return 0;
    const int hpg = (HD % 48 == 0 ? 96 : 64) / HD;
    return heads % hpg == 0 ? 1 : 0;
}

int rows_form_launch(const mphsir_fused_gram_args* a, int dtype, hipStream_t s) {
    RowGramDev d{a->X, (long)a->ldx, a->ln_w, a->ln_b, a->Wqkv, a->w9, (long)a->ldw, a->V, (long)a->ldvo, a->Gpart, a->Spart,
                 a->B, a->H, a->W, a->row_segments, a->T, (long)a->ldt, a->QK, (long)a->ldqk, fused_debug_buffer()};
    return MPHSIR_DISPATCH_T(dtype, (RowShapes<T_>::run(d, a->C, a->C / a->heads, s)));
}

}  // namespace mphsir

extern "C" int mphsir_qkv_dwconv_gram_rows_fits(int32_t C, int32_t heads, int32_t H, int32_t W, int dtype) {
    if (!MPHSIR_DTYPE_OK(dtype)) return 0;
    return mphsir::rows_form_fits(C, heads, H, W, dtype);
}
