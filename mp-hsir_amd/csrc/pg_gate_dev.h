// Transposed fp32 mat-vec of the local spectral-prompt gate (PG_Spectral_Attention.forward net/MP_HSIR.py:132-152),
// shared by its forward (tail of win_attn) and its backward (pg_gate_bwd).
#pragma once
#include "mphsir_dev.h"

namespace mphsir {

// y[j] = sum_p x[p] * W[p][j]  for j < ncols; W row-major [np][ncols] in global memory, x / y in LDS, red = 256 floats
// of LDS scratch.  Written as one thread per output with a serial loop over p this is a chain of np dependent L2 round
// trips on ncols lanes; here the p axis is split over 256/ncols thread groups (loads coalesced along j) and the
// partials meet in LDS.  Contains one __syncthreads(); y is written by threads j < ncols (visible after the caller's
// next barrier).
__device__ __forceinline__ void pg_matvec_cols(const float* W, int np, int ncols, const float* x, float* y, float* red) {
    const int tid = threadIdx.x;
    if (ncols >= 256) {                       // already one thread per column and more
        for (int j = tid; j < ncols; j += 256) {
            float acc = 0.f;
            for (int p = 0; p < np; ++p) acc += x[p] * W[(long)p * ncols + j];
            y[j] = acc;
        }
        __syncthreads();
        return;
    }
    const int nparts = 256 / ncols, j = tid % ncols, part = tid / ncols;
    float acc = 0.f;
    if (part < nparts)
        for (int p = part; p < np; p += nparts) acc += x[p] * W[(long)p * ncols + j];
    red[tid] = acc;                           // [part][ncols]: tid = part * ncols + j for part < nparts
    __syncthreads();
    if (tid < ncols) {
        float s = 0.f;
        for (int q = 0; q < nparts; ++q) s += red[q * ncols + tid];
        y[tid] = s;
    }
}

// y[row] = sum_c W[row][c] * x[c]  for row < nrows (nrows <= 128); W row-major [nrows][C] in global memory.  The c axis
// is split over 256/128 = 2 thread groups (each lane still walks its own row: no cross-lane reduction, which measured
// slower than the serial loop it replaced), partials meet in LDS.  One __syncthreads(); y written by threads < nrows.
__device__ __forceinline__ void pg_matvec_rows(const float* W, int nrows, int C, const float* x, float* y, float* red) {
    const int tid = threadIdx.x, row = tid & 127, part = tid >> 7, half = C / 2;
    float acc = 0.f;
    if (row < nrows) {
        const float* wr = W + (long)row * C + part * half;
        const float* xr = x + part * half;
        for (int c = 0; c < half; ++c) acc += wr[c] * xr[c];
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < nrows) y[tid] = red[tid] + red[tid + 128];
}

}  // namespace mphsir
