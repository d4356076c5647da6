// Plain depthwise 3x3 on channels-last cubes: forward / backward-data (same kernel, taps flipped) and
// the weight gradient (two-stage deterministic reduction over pixels).
//
// Used by the backward pass of every `*_dwconv` of the reference (nn.Conv2d(..., groups=channels),
// net/MP_HSIR.py:92,227,230,257,382): d/dx of a depthwise correlation is the depthwise correlation of
// dy with the spatially flipped taps; d/dw[tap][c] = sum over pixels of x[pixel+tap][c] * dy[pixel][c].
// Pure HBM/L2 streaming: 16 B per lane along the channel axis, fp32 accumulation.
#include <stdlib.h>

#include "mphsir_dev.h"
#include "mphsir_host.h"

namespace mphsir {

struct DwDev {
    const void* X; long ldx; const float* w9; long ldw; void* Y; long ldy;
    int B, H, W, C, flip;
};

// One thread = one 16-byte channel vector x one strip of DW_S pixels along x.  The 9 taps (fp32) live in registers and
// the 3x3 neighbourhood slides: 3 new vector loads per output pixel instead of 9 + 18 (taps) -- the first version was
// bound by L1/TA bandwidth (432 B of vector-memory traffic per 16-byte output), not by HBM.
constexpr int DW_S = 8;

template <class T>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwDev a) {
    constexpr int VEC = Vec16<T>::N;
    const int cv = a.C / VEC, nsx = a.W / DW_S;
    const long total = (long)a.B * a.H * nsx * cv;
    const long idx = xcd_contiguous_block() * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c0 = (int)(idx % cv) * VEC;
    long q = idx / cv;
    const int x0 = (int)(q % nsx) * DW_S;  q /= nsx;
    const int y = (int)(q % a.H), b = (int)(q / a.H);
    const T* X = reinterpret_cast<const T*>(a.X) + (long)b * a.H * a.W * a.ldx + c0;
    T* Y = reinterpret_cast<T*>(a.Y) + (long)b * a.H * a.W * a.ldy + c0;
    float w[9][VEC];
#pragma unroll
    for (int t = 0; t < 9; ++t)
        for (int e = 0; e < VEC; ++e) w[t][e] = a.w9[(a.flip ? 8 - t : t) * a.ldw + c0 + e];
    auto column = [&](int x, Vec16<T> (&col)[3]) {      // rows y-1, y, y+1 of column x (zeros outside the image)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = y + r - 1;
            if (x >= 0 && x < a.W && yy >= 0 && yy < a.H) col[r] = load16<T>(X + ((long)yy * a.W + x) * a.ldx);
            else col[r] = Vec16<T>{};
        }
    };
    Vec16<T> cl[3], cm[3], cr[3];
    column(x0 - 1, cl);
    column(x0, cm);
#pragma unroll
    for (int i = 0; i < DW_S; ++i) {
        column(x0 + i + 1, cr);
        float acc[VEC];
        for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            for (int e = 0; e < VEC; ++e)
                acc[e] += cl[r].get(e) * w[r * 3][e] + cm[r].get(e) * w[r * 3 + 1][e] + cr[r].get(e) * w[r * 3 + 2][e];
        Vec16<T> o;
        for (int e = 0; e < VEC; ++e) o.set(e, acc[e]);
        store16<T>(Y + ((long)y * a.W + x0 + i) * a.ldy, o);
#pragma unroll
        for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
    }
}

// ---- tile form (16-bit types, H % 8 == 0, W % 16 == 0, C % 32 == 0) ------------------------------------------------------
// One workgroup (8 waves) = an 8x16-pixel tile + one-pixel halo x a slab of up to 96 channels.  The halo tile goes to LDS as
// fp32 through coalesced 16-byte loads that are all in flight at once (the strip form above issues 30 dependent-use loads per
// thread through L1), each element is unpacked ONCE (the strip form unpacks it 3.75 times), and the 3x3 window slides over
// LDS (f32x4 reads, fma chains) exactly as in the fused spectral pass A (spectral_fused.hip).  Two workgroups per CU.
constexpr int DT_TH = 8, DT_TW = 16, DT_HW = DT_TW + 2, DT_ROWS = (DT_TH + 2) * DT_HW, DT_CS = 96, DT_LD = DT_CS + 4;
constexpr int DT_THREADS = 512;

template <class T>
__global__ __launch_bounds__(DT_THREADS, 4) void dwconv3x3_tile_kernel(DwDev a) {
    constexpr int VEC = Vec16<T>::N;                       // 8
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* Ts = reinterpret_cast<float*>(smem_v);          // [180][DT_LD]
    float* taps = Ts + DT_ROWS * DT_LD;                    // [9][DT_CS]
    const int tid = threadIdx.x;
    const int tilesx = a.W / DT_TW, tiles = (a.H / DT_TH) * tilesx, nslab = (a.C + DT_CS - 1) / DT_CS;
    const long blk = (gridDim.x & 7) == 0 ? xcd_contiguous_block() : (long)blockIdx.x;
    if (blk >= (long)a.B * tiles * nslab) return;
    const int tile = (int)(blk % tiles), slab = (int)((blk / tiles) % nslab), b = (int)(blk / ((long)tiles * nslab));
    const int ty0 = (tile / tilesx) * DT_TH, tx0 = (tile % tilesx) * DT_TW;
    const int cs0 = slab * DT_CS, cw = (a.C - cs0) < DT_CS ? (a.C - cs0) : DT_CS, vpr = cw / VEC;
    const T* X = reinterpret_cast<const T*>(a.X) + (long)b * a.H * a.W * a.ldx + cs0;
    T* Y = reinterpret_cast<T*>(a.Y) + (long)b * a.H * a.W * a.ldy + cs0;

    // ---- stage: halo pixels x channel vectors, every load requested before the first is used.  (A capped, persistent grid
    // with the next tile-slab's loads in flight during the pass was measured no faster: with two workgroups per CU the stage
    // of one already overlaps the pass of the other.)
    constexpr int NV = (DT_ROWS * (DT_CS / VEC) + DT_THREADS - 1) / DT_THREADS;       // 5
    Vec16<T> xv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + DT_THREADS * i, r = idx / vpr, v = idx % vpr;
        const int y = ty0 - 1 + r / DT_HW, x = tx0 - 1 + r % DT_HW;
        if (r < DT_ROWS && y >= 0 && y < a.H && x >= 0 && x < a.W) xv[i] = load16<T>(X + ((long)y * a.W + x) * a.ldx + v * VEC);
        else xv[i] = Vec16<T>{};
    }
    for (int i = tid; i < 9 * cw; i += DT_THREADS) {
        const int t = i / cw, c = i % cw;
        taps[t * DT_CS + c] = a.w9[(a.flip ? 8 - t : t) * a.ldw + cs0 + c];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = tid + DT_THREADS * i, r = idx / vpr, v = idx % vpr;
        if (r < DT_ROWS) {
            float* dst = Ts + r * DT_LD + v * VEC;
            *reinterpret_cast<f32x4*>(dst) = f32x4{xv[i].get(0), xv[i].get(1), xv[i].get(2), xv[i].get(3)};
            *reinterpret_cast<f32x4*>(dst + 4) = f32x4{xv[i].get(4), xv[i].get(5), xv[i].get(6), xv[i].get(7)};
        }
    }
    __syncthreads();

    // ---- 3x3 window over the LDS tile: one thread = 4 channels x a strip of 8 pixels
    const int qpr = cw / 4, it = tid;
    if (it < qpr * 16) {
        const int c4 = it % qpr, st = it / qpr, iy = st >> 1, ix0 = (st & 1) * 8;
        const float* tsrc = Ts + (iy * DT_HW + ix0) * DT_LD + c4 * 4;
        f32x4 w[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) w[t] = *reinterpret_cast<const f32x4*>(taps + t * DT_CS + c4 * 4);
        auto tvec = [&](int r, int col) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(tsrc + (r * DT_HW + col) * DT_LD); };
        f32x4 cl[3], cm[3], cr[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { cl[r] = tvec(r, 0); cm[r] = tvec(r, 1); }
        T* ydst = Y + ((long)(ty0 + iy) * a.W + tx0 + ix0) * a.ldy + c4 * 4;
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
            store4<T>(ydst + (long)i * a.ldy, o);
#pragma unroll
            for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
        }
    }
}

template <class T> static int launch_dw_tile(const DwDev& d, hipStream_t s) {
    const long nblk = (long)d.B * (d.H / DT_TH) * (d.W / DT_TW) * ((d.C + DT_CS - 1) / DT_CS);
    const long blocks = (nblk + 7) / 8 * 8;              // multiple of 8: XCD-contiguous order
    const size_t shmem = ((size_t)DT_ROWS * DT_LD + 9 * DT_CS) * sizeof(float);
    allow_big_lds(dwconv3x3_tile_kernel<T>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_DWCONV, (dwconv3x3_tile_kernel<T>), dim3((unsigned)blocks), dim3(DT_THREADS), shmem, s, d);
    return MPHSIR_OK;
}

struct DwWgDev {
    const void* X; long ldx; const void* dY; long lddy; float* part;   // part [nblk][9][C]
    int B, H, W, C, nblk;
};

// block = 64 channel vectors (lanes) x 4 pixel groups (waves); grid = (nblk, ceil(C/VEC/64))
template <class T>
__global__ __launch_bounds__(256) void dwconv3x3_wgrad_kernel(DwWgDev a) {
    constexpr int VEC = Vec16<T>::N;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* red = reinterpret_cast<float*>(smem_v);        // [64][9*VEC]
    const int lane = threadIdx.x & 63, wv = wave_id_uniform();
    const int cvec = blockIdx.y * 64 + lane, c0 = cvec * VEC;
    const bool live = c0 < a.C;
    const int HW = a.H * a.W;
    const long P = (long)a.B * HW, per = ((P + a.nblk - 1) / a.nblk + DW_S - 1) / DW_S * DW_S;   // whole strips per block
    // neighbouring pixel ranges share their boundary rows: keep them on one XCD (one L2) when the grid allows it
    const long blk = (gridDim.x & 7) == 0 ? xcd_contiguous_block() : (long)blockIdx.x;
    const long p_lo = blk * per, p_hi = (p_lo + per < P) ? p_lo + per : P;
    const T* X = reinterpret_cast<const T*>(a.X);
    const T* dY = reinterpret_cast<const T*>(a.dY);
    float acc[9][VEC];
    for (int t = 0; t < 9; ++t)
        for (int e = 0; e < VEC; ++e) acc[t][e] = 0.f;
    // each wave walks strips of DW_S pixels along x with a sliding 3x3 window: 3 new x vectors + 1 dy vector per pixel
    const long s_lo = p_lo / DW_S, s_hi = p_hi / DW_S;
    if (live)
        for (long sidx = s_lo + wv; sidx < s_hi; sidx += 4) {
            const long pix0 = sidx * DW_S;
            const int b = (int)(pix0 / HW), p = (int)(pix0 % HW), y = p / a.W, x0 = p % a.W;
            const T* base = X + (long)b * HW * a.ldx + c0;
            auto column = [&](int x, Vec16<T> (&col)[3]) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int yy = y + r - 1;
                    if (x >= 0 && x < a.W && yy >= 0 && yy < a.H) col[r] = load16<T>(base + ((long)yy * a.W + x) * a.ldx);
                    else col[r] = Vec16<T>{};
                }
            };
            Vec16<T> cl[3], cm[3], cr[3];
            column(x0 - 1, cl);
            column(x0, cm);
#pragma unroll
            for (int i = 0; i < DW_S; ++i) {
                column(x0 + i + 1, cr);
                const Vec16<T> g = load16<T>(dY + (pix0 + i) * a.lddy + c0);
#pragma unroll
                for (int r = 0; r < 3; ++r)
                    for (int e = 0; e < VEC; ++e) {
                        const float ge = g.get(e);
                        acc[r * 3][e] += cl[r].get(e) * ge;
                        acc[r * 3 + 1][e] += cm[r].get(e) * ge;
                        acc[r * 3 + 2][e] += cr[r].get(e) * ge;
                    }
#pragma unroll
                for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
            }
        }
    // cross-wave reduction in wave order through one [64][9*VEC] buffer (18 KB: keeps 4+ workgroups per CU resident)
    float* mine = red + (lane * 9) * VEC;
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
            if (w == 0) {
                for (int t = 0; t < 9; ++t)
                    for (int e = 0; e < VEC; ++e) mine[t * VEC + e] = acc[t][e];
            } else {
                for (int t = 0; t < 9; ++t)
                    for (int e = 0; e < VEC; ++e) mine[t * VEC + e] += acc[t][e];
            }
        }
        __syncthreads();
    }
    if (wv == 0 && live) {
        float* out = a.part + (long)blockIdx.x * 9 * a.C;
        for (int t = 0; t < 9; ++t)
            for (int e = 0; e < VEC; ++e) out[t * a.C + c0 + e] = mine[t * VEC + e];
    }
}

struct GateBwdDev {
    const void* T; const void* dU; void* U; void* dT; long M; int HP;
};

// backward of u = gelu(x1) * x2 with [x1|x2] = T[:, :HP], T[:, HP:]  (FFN/FeedForward :263, :389); also re-emits u
template <class T>
__global__ __launch_bounds__(256) void gdfn_gate_bwd_kernel(GateBwdDev a) {
    constexpr int VEC = Vec16<T>::N;
    const int vpr = a.HP / VEC;
    const long total = a.M * vpr;
    const T* Tin = reinterpret_cast<const T*>(a.T);
    const T* dU = reinterpret_cast<const T*>(a.dU);
    T* U = reinterpret_cast<T*>(a.U);
    T* dT = reinterpret_cast<T*>(a.dT);
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / vpr;
        const int c0 = (int)(idx % vpr) * VEC;
        const Vec16<T> x1 = load16<T>(Tin + m * 2 * a.HP + c0), x2 = load16<T>(Tin + m * 2 * a.HP + a.HP + c0);
        const Vec16<T> du = load16<T>(dU + m * a.HP + c0);
        Vec16<T> u, d1, d2;
        for (int e = 0; e < VEC; ++e) {
            float g, dgl;
            Math<T>::gelu_pair(x1.get(e), g, dgl);
            u.set(e, g * x2.get(e));
            d1.set(e, du.get(e) * x2.get(e) * dgl);
            d2.set(e, du.get(e) * g);
        }
        store16<T>(U + m * a.HP + c0, u);
        store16<T>(dT + m * 2 * a.HP + c0, d1);
        store16<T>(dT + m * 2 * a.HP + a.HP + c0, d2);
    }
}

// ---- tile form of the weight gradient (16-bit types, H % 8 == 0, W % 16 == 0, C % 32 == 0) -----------------------------
// Workgroup (pblk, slab) walks the 8x16-pixel tiles pblk, pblk + nblk, ... of ALL images for its 96-channel slab: the x halo
// tile goes through LDS as fp32 exactly as in dwconv3x3_tile_kernel (the next tile's loads are in flight during the pass), a
// thread = 4 channels x a strip of 8 pixels keeps its 9 tap sums (f32x4 each) in registers across all its tiles, dy comes
// straight from global (8 bytes per pixel, the lanes of a pixel are contiguous).  One ordered LDS reduction over the 16
// strips at the end -> part[pblk][9][C]: fixed order, no atomics.
// (the 9 tap sums, the window and the prefetched vectors want ~170 VGPRs: one workgroup of 8 waves per CU)
template <class T>
__global__ __launch_bounds__(DT_THREADS, 2) void dwconv3x3_wgrad_tile_kernel(DwWgDev a) {
    constexpr int VEC = Vec16<T>::N;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* Ts = reinterpret_cast<float*>(smem_v);          // [180][DT_LD]; at the end: [16 strips][9][DT_CS] partial sums
    const int tid = threadIdx.x;
    const int tilesx = a.W / DT_TW, tiles = (a.H / DT_TH) * tilesx;
    const long total = (long)a.B * tiles;
    const int slab = blockIdx.y, cs0 = slab * DT_CS, cw = (a.C - cs0) < DT_CS ? (a.C - cs0) : DT_CS, vpr = cw / VEC, qpr = cw / 4;
    const bool on = tid < qpr * 16;
    const int c4 = tid % qpr, st = tid / qpr, iy = st >> 1, ix0 = (st & 1) * 8;
    typedef typename ElemTraits<T>::vec4_t v4_t;

    constexpr int NV = (DT_ROWS * (DT_CS / VEC) + DT_THREADS - 1) / DT_THREADS;       // 5
    Vec16<T> xv[NV];
    v4_t gy[8];
    auto gload = [&](long t) __attribute__((always_inline)) {
        const int b = (int)(t / tiles), tile = (int)(t % tiles);
        const int ty0 = (tile / tilesx) * DT_TH, tx0 = (tile % tilesx) * DT_TW;
        const T* X = reinterpret_cast<const T*>(a.X) + (long)b * a.H * a.W * a.ldx + cs0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + DT_THREADS * i, r = idx / vpr, v = idx % vpr;
            const int y = ty0 - 1 + r / DT_HW, x = tx0 - 1 + r % DT_HW;
            if (r < DT_ROWS && y >= 0 && y < a.H && x >= 0 && x < a.W) xv[i] = load16<T>(X + ((long)y * a.W + x) * a.ldx + v * VEC);
            else xv[i] = Vec16<T>{};
        }
        if (on) {
            const T* g = reinterpret_cast<const T*>(a.dY) + ((long)b * a.H * a.W + (long)(ty0 + iy) * a.W + tx0 + ix0) * a.lddy + cs0 + c4 * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) gy[i] = *reinterpret_cast<const v4_t*>(g + (long)i * a.lddy);
        }
    };
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    long t = blockIdx.x;
    if (t < total) gload(t);
    for (; t < total; t += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + DT_THREADS * i, r = idx / vpr, v = idx % vpr;
            if (r < DT_ROWS) {
                float* dst = Ts + r * DT_LD + v * VEC;
                *reinterpret_cast<f32x4*>(dst) = f32x4{xv[i].get(0), xv[i].get(1), xv[i].get(2), xv[i].get(3)};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{xv[i].get(4), xv[i].get(5), xv[i].get(6), xv[i].get(7)};
            }
        }
        f32x4 g[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) g[i] = f32x4{to_f32(gy[i][0]), to_f32(gy[i][1]), to_f32(gy[i][2]), to_f32(gy[i][3])};
        __syncthreads();
        if (t + gridDim.x < total) gload(t + gridDim.x);
        if (on) {
            const float* tsrc = Ts + (iy * DT_HW + ix0) * DT_LD + c4 * 4;
            auto tvec = [&](int r, int col) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(tsrc + (r * DT_HW + col) * DT_LD); };
            f32x4 cl[3], cm[3], cr[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { cl[r] = tvec(r, 0); cm[r] = tvec(r, 1); }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int r = 0; r < 3; ++r) cr[r] = tvec(r, i + 2);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc[r * 3] = __builtin_elementwise_fma(cl[r], g[i], acc[r * 3]);
                    acc[r * 3 + 1] = __builtin_elementwise_fma(cm[r], g[i], acc[r * 3 + 1]);
                    acc[r * 3 + 2] = __builtin_elementwise_fma(cr[r], g[i], acc[r * 3 + 2]);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
            }
        }
        __syncthreads();           // the tile is free for the next stage (and, after the last tile, for the strip sums)
    }
    float* red = Ts;                                       // [16][9][DT_CS]
    if (on)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) *reinterpret_cast<f32x4*>(red + (st * 9 + tp) * DT_CS + c4 * 4) = acc[tp];
    __syncthreads();
    for (int i = tid; i < 9 * cw; i += DT_THREADS) {
        const int tp = i / cw, c = i % cw;
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += red[(k * 9 + tp) * DT_CS + c];
        a.part[((long)blockIdx.x * 9 + tp) * a.C + cs0 + c] = s;
    }
}

// ---- both gradients of a depthwise 3x3 in one pass (tile form shapes) ------------------------------------------------------------
// dX[p] = sum_taps w[tap] dY[p - tap]   and   dW[tap] = sum_p X[p] dY[p - tap]   read the SAME nine shifted dY values per pixel:
// the dY halo tile goes through LDS (fp32) as in dwconv3x3_tile_kernel, X comes straight from global for the interior pixels
// (8 bytes per pixel and thread), a thread = 4 channels x a strip of 8 pixels produces its dX values and adds X[p] dY[p - tap]
// to its nine tap sums, which stay in registers while the persistent workgroup (pblk, slab) walks its tiles; one ordered LDS
// reduction at the end.  Against dwconv3x3(flip) + dwconv3x3_wgrad: dY is read once instead of twice (5 -> 4 tensor passes),
// one launch instead of two.
struct DwBwdDev {
    const void* X; long ldx; const void* dY; long lddy; const float* w9; long ldw; void* dX; long lddx; float* part;
    int B, H, W, C, nblk;
};

template <class T>
__global__ __launch_bounds__(DT_THREADS, 2) void dwconv3x3_bwd_tile_kernel(DwBwdDev a) {
    constexpr int VEC = Vec16<T>::N;
    HIP_DYNAMIC_SHARED(f32x4, smem_v)
    float* Ts = reinterpret_cast<float*>(smem_v);          // [180][DT_LD] dY halo tile; at the end: [16 strips][9][DT_CS] partial sums
    const int tid = threadIdx.x;
    const int tilesx = a.W / DT_TW, tiles = (a.H / DT_TH) * tilesx;
    const long total = (long)a.B * tiles;
    const int slab = blockIdx.y, cs0 = slab * DT_CS, cw = (a.C - cs0) < DT_CS ? (a.C - cs0) : DT_CS, vpr = cw / VEC, qpr = cw / 4;
    const bool on = tid < qpr * 16;
    const int c4 = tid % qpr, st = tid / qpr, iy = st >> 1, ix0 = (st & 1) * 8;
    typedef typename ElemTraits<T>::vec4_t v4_t;

    constexpr int NV = (DT_ROWS * (DT_CS / VEC) + DT_THREADS - 1) / DT_THREADS;       // 5
    Vec16<T> gv[NV];
    v4_t xy[8];
    auto gload = [&](long t) __attribute__((always_inline)) {
        const int b = (int)(t / tiles), tile = (int)(t % tiles);
        const int ty0 = (tile / tilesx) * DT_TH, tx0 = (tile % tilesx) * DT_TW;
        const T* G = reinterpret_cast<const T*>(a.dY) + (long)b * a.H * a.W * a.lddy + cs0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + DT_THREADS * i, r = idx / vpr, v = idx % vpr;
            const int y = ty0 - 1 + r / DT_HW, x = tx0 - 1 + r % DT_HW;
            if (r < DT_ROWS && y >= 0 && y < a.H && x >= 0 && x < a.W) gv[i] = load16<T>(G + ((long)y * a.W + x) * a.lddy + v * VEC);
            else gv[i] = Vec16<T>{};
        }
        if (on) {
            const T* xs = reinterpret_cast<const T*>(a.X) + ((long)b * a.H * a.W + (long)(ty0 + iy) * a.W + tx0 + ix0) * a.ldx + cs0 + c4 * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) xy[i] = *reinterpret_cast<const v4_t*>(xs + (long)i * a.ldx);
        }
    };
    // window position (r, c) holds dY[p + (r-1, c-1)] = dY[p - tap] for tap (1-r, 1-c): it meets the flipped tap 8 - (3r + c)
    f32x4 w[9], acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        w[t] = on ? *reinterpret_cast<const f32x4*>(a.w9 + (8 - t) * a.ldw + cs0 + c4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    long t = blockIdx.x;
    if (t < total) gload(t);
    for (; t < total; t += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = tid + DT_THREADS * i, r = idx / vpr, v = idx % vpr;
            if (r < DT_ROWS) {
                float* dst = Ts + r * DT_LD + v * VEC;
                *reinterpret_cast<f32x4*>(dst) = f32x4{gv[i].get(0), gv[i].get(1), gv[i].get(2), gv[i].get(3)};
                *reinterpret_cast<f32x4*>(dst + 4) = f32x4{gv[i].get(4), gv[i].get(5), gv[i].get(6), gv[i].get(7)};
            }
        }
        f32x4 xin[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xin[i] = f32x4{to_f32(xy[i][0]), to_f32(xy[i][1]), to_f32(xy[i][2]), to_f32(xy[i][3])};
        const int b = (int)(t / tiles), tile = (int)(t % tiles);
        const int ty0 = (tile / tilesx) * DT_TH, tx0 = (tile % tilesx) * DT_TW;
        __syncthreads();
        if (t + gridDim.x < total) gload(t + gridDim.x);
        if (on) {
            const float* tsrc = Ts + (iy * DT_HW + ix0) * DT_LD + c4 * 4;
            auto tvec = [&](int r, int col) __attribute__((always_inline)) { return *reinterpret_cast<const f32x4*>(tsrc + (r * DT_HW + col) * DT_LD); };
            T* xdst = reinterpret_cast<T*>(a.dX) + ((long)b * a.H * a.W + (long)(ty0 + iy) * a.W + tx0 + ix0) * a.lddx + cs0 + c4 * 4;
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
                store4<T>(xdst + (long)i * a.lddx, o);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    acc[r * 3] = __builtin_elementwise_fma(cl[r], xin[i], acc[r * 3]);
                    acc[r * 3 + 1] = __builtin_elementwise_fma(cm[r], xin[i], acc[r * 3 + 1]);
                    acc[r * 3 + 2] = __builtin_elementwise_fma(cr[r], xin[i], acc[r * 3 + 2]);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) { cl[r] = cm[r]; cm[r] = cr[r]; }
            }
        }
        __syncthreads();           // the tile is free for the next stage (and, after the last tile, for the strip sums)
    }
    float* red = Ts;                                       // [16][9][DT_CS], tap-major in the PARAMETER's tap order
    if (on)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) *reinterpret_cast<f32x4*>(red + (st * 9 + tp) * DT_CS + c4 * 4) = acc[8 - tp];
    __syncthreads();
    for (int i = tid; i < 9 * cw; i += DT_THREADS) {
        const int tp = i / cw, c = i % cw;
        float s = 0.f;
        for (int k = 0; k < 16; ++k) s += red[(k * 9 + tp) * DT_CS + c];
        a.part[((long)blockIdx.x * 9 + tp) * a.C + cs0 + c] = s;
    }
}

template <class T> static int launch_dw_bwd_tile(const DwBwdDev& d, hipStream_t s) {
    const size_t shmem = (size_t)DT_ROWS * DT_LD * sizeof(float);       // >= 16 * 9 * DT_CS floats
    allow_big_lds(dwconv3x3_bwd_tile_kernel<T>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_DWCONV_BWD, (dwconv3x3_bwd_tile_kernel<T>), dim3(d.nblk, (d.C + DT_CS - 1) / DT_CS), dim3(DT_THREADS), shmem, s, d);
    return MPHSIR_OK;
}

template <class T> static int launch_dw_wgrad_tile(const DwWgDev& d, hipStream_t s) {
    const size_t shmem = (size_t)DT_ROWS * DT_LD * sizeof(float);       // >= 16 * 9 * DT_CS floats
    allow_big_lds(dwconv3x3_wgrad_tile_kernel<T>, shmem);
    MPHSIR_LAUNCH(MPHSIR_K_DWCONV_WGRAD, (dwconv3x3_wgrad_tile_kernel<T>), dim3(d.nblk, (d.C + DT_CS - 1) / DT_CS), dim3(DT_THREADS), shmem, s, d);
    return MPHSIR_OK;
}

}  // namespace mphsir

extern "C" int mphsir_gdfn_gate_bwd(const void* T, const void* dU, void* U, void* dT, int64_t M, int32_t HP, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(T && dU && U && dT, "gdfn_gate_bwd: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "gdfn_gate_bwd: dtype %d unsupported", dtype);
    MPHSIR_REQUIRE(M > 0 && HP > 0 && HP % 8 == 0, "gdfn_gate_bwd: bad shape");
    MPHSIR_REQUIRE(aligned16(T) && aligned16(dU) && aligned16(U) && aligned16(dT), "gdfn_gate_bwd: 16-byte alignment required");
    GateBwdDev d{T, dU, U, dT, (long)M, HP};
    const int vec = dtype == MPHSIR_F32 ? 4 : 8;
    long blocks = (M * (HP / vec) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MPHSIR_F32)
        MPHSIR_LAUNCH(MPHSIR_K_GDFN_GATE_BWD, (gdfn_gate_bwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    else if (dtype == MPHSIR_BF16)
        MPHSIR_LAUNCH(MPHSIR_K_GDFN_GATE_BWD, (gdfn_gate_bwd_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    else
        MPHSIR_LAUNCH(MPHSIR_K_GDFN_GATE_BWD, (gdfn_gate_bwd_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    return MPHSIR_OK;
}

extern "C" int mphsir_dwconv3x3_wgrad_tiled(int32_t H, int32_t W, int32_t C, int dtype) {
    // 1 if mphsir_dwconv3x3_wgrad takes the tile form for this shape: the caller then sizes nblk for one workgroup per CU
    // over (nblk, ceil(C/96)) instead of one partial block per 128 pixels
    return ((dtype == MPHSIR_BF16 || dtype == MPHSIR_F16) && H > 0 && W > 0 && H % mphsir::DT_TH == 0 && W % mphsir::DT_TW == 0 &&
            C % 32 == 0) ? 1 : 0;
}

extern "C" int mphsir_dwconv3x3(const void* X, int64_t ldx, const float* w9, int64_t ldw, void* Y, int64_t ldy,
                                int32_t B, int32_t H, int32_t W, int32_t C, int32_t flip, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && w9 && Y, "dwconv3x3: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "dwconv3x3: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2, vec = 16 / esz;
    MPHSIR_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0, "dwconv3x3: C must be a multiple of %d", vec);
    MPHSIR_REQUIRE(aligned16(X) && aligned16(Y) && (ldx * esz) % 16 == 0 && (ldy * esz) % 16 == 0, "dwconv3x3: 16-byte alignment required");
    DwDev d{X, (long)ldx, w9, (long)ldw, Y, (long)ldy, B, H, W, C, flip};
    MPHSIR_REQUIRE(W % DW_S == 0, "dwconv3x3: W must be a multiple of %d", DW_S);
    const long blocks = (((long)B * H * (W / DW_S) * (C / vec) + 255) / 256 + 7) / 8 * 8;      // multiple of 8: XCD-contiguous order
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (dtype != MPHSIR_F32 && H % DT_TH == 0 && W % DT_TW == 0 && C % 32 == 0)
        return dtype == MPHSIR_BF16 ? launch_dw_tile<bf16_t>(d, s) : launch_dw_tile<f16_t>(d, s);
    if (dtype == MPHSIR_F32)
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV, (dwconv3x3_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    else if (dtype == MPHSIR_BF16)
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV, (dwconv3x3_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    else
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV, (dwconv3x3_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, s, d);
    return MPHSIR_OK;
}

extern "C" int mphsir_dwconv3x3_wgrad(const void* X, int64_t ldx, const void* dY, int64_t lddy, float* partial, int32_t nblk,
                                      int32_t B, int32_t H, int32_t W, int32_t C, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && dY && partial, "dwconv3x3_wgrad: null pointer");
    MPHSIR_REQUIRE(MPHSIR_DTYPE_OK(dtype), "dwconv3x3_wgrad: dtype %d unsupported", dtype);
    const int esz = dtype == MPHSIR_F32 ? 4 : 2, vec = 16 / esz;
    MPHSIR_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % vec == 0 && nblk > 0 && W % DW_S == 0, "dwconv3x3_wgrad: bad shape");
    MPHSIR_REQUIRE(aligned16(X) && aligned16(dY) && (ldx * esz) % 16 == 0 && (lddy * esz) % 16 == 0, "dwconv3x3_wgrad: 16-byte alignment required");
    DwWgDev d{X, (long)ldx, dY, (long)lddy, partial, B, H, W, C, nblk};
    const size_t shmem = 64 * 9 * vec * sizeof(float);
    dim3 grid(nblk, (C / vec + 63) / 64);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (mphsir_dwconv3x3_wgrad_tiled(H, W, C, dtype))
        return dtype == MPHSIR_BF16 ? launch_dw_wgrad_tile<bf16_t>(d, s) : launch_dw_wgrad_tile<f16_t>(d, s);
    if (dtype == MPHSIR_F32) {
        allow_big_lds(dwconv3x3_wgrad_kernel<float>, shmem);
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV_WGRAD, (dwconv3x3_wgrad_kernel<float>), grid, dim3(256), shmem, s, d);
    } else if (dtype == MPHSIR_BF16) {
        allow_big_lds(dwconv3x3_wgrad_kernel<bf16_t>, shmem);
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV_WGRAD, (dwconv3x3_wgrad_kernel<bf16_t>), grid, dim3(256), shmem, s, d);
    } else {
        allow_big_lds(dwconv3x3_wgrad_kernel<f16_t>, shmem);
        MPHSIR_LAUNCH(MPHSIR_K_DWCONV_WGRAD, (dwconv3x3_wgrad_kernel<f16_t>), grid, dim3(256), shmem, s, d);
    }
    return MPHSIR_OK;
}

extern "C" int mphsir_dwconv3x3_bwd(const void* X, int64_t ldx, const void* dY, int64_t lddy, const float* w9, int64_t ldw, void* dX, int64_t lddx,
                                    float* partial, int32_t nblk, int32_t B, int32_t H, int32_t W, int32_t C, int dtype, void* stream) {
    using namespace mphsir;
    clear_error();
    MPHSIR_REQUIRE(X && dY && w9 && dX && partial, "dwconv3x3_bwd: null pointer");
    MPHSIR_REQUIRE(B > 0 && nblk > 0 && mphsir_dwconv3x3_wgrad_tiled(H, W, C, dtype),
                   "dwconv3x3_bwd: (H=%d, W=%d, C=%d, dtype=%d) not covered (16-bit types, H %% 8 == 0, W %% 16 == 0, C %% 32 == 0: ask "
                   "mphsir_dwconv3x3_wgrad_tiled; otherwise mphsir_dwconv3x3(flip=1) + mphsir_dwconv3x3_wgrad)", H, W, C, dtype);
    MPHSIR_REQUIRE(aligned16(X) && aligned16(dY) && aligned16(dX) && aligned16(w9) && (ldx * 2) % 16 == 0 && (lddy * 2) % 16 == 0 &&
                       (lddx * 2) % 16 == 0 && (ldw * 4) % 16 == 0, "dwconv3x3_bwd: 16-byte alignment required");
    MPHSIR_REQUIRE(dX != dY, "dwconv3x3_bwd: dX must not alias dY (neighbouring tiles read dY's halo)");
    DwBwdDev d{X, (long)ldx, dY, (long)lddy, w9, (long)ldw, dX, (long)lddx, partial, B, H, W, C, nblk};
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return dtype == MPHSIR_BF16 ? launch_dw_bwd_tile<bf16_t>(d, s) : launch_dw_bwd_tile<f16_t>(d, s);
}
