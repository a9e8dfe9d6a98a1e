// Gaussian random field as four dense fp64 GEMMs on the matrix cores (simulations/ground_truths.py:14-33:
// field = Re ifft2(fft2(white) * amp), min-max normalised).
//
// The white noise is real and, for even n, the amplitude of the reference is real and even in EACH spectral index
// (fft_indices, ground_truths.py:7-11: amp[k][l] depends on k_idx[k]^2 + k_idx[l]^2).  For such a filter the separable
// Hartley transform T = H (x) H, H[j][k] = cos(2 pi j k / n) + sin(2 pi j k / n) (real, symmetric, H H = n I),
// diagonalises it:  T(w)[k][l] = Re w^[k][-l] - Im w^[k][l],  so  T(filtered) = amp .* T(w)  and
//      field = H (amp .* (H w H)) H / n^2          -- four real n x n GEMMs, no complex arithmetic
// (checked against numpy.fft to 2e-15 for n = 10 .. 100; the constant 1 / n^2 is dropped, the field is min-max
// normalised right after).  8 n^3 flops per field instead of the 6 n^3 of the half-spectrum DFT + circular convolutions
// of k_grf_dft.h, but as GEMMs: k_grf_dft runs at 5-13 of the 78 fp64 TFLOP/s (every FMA operand comes out of LDS), and
// with 16-step episodes at 100x100 (BASELINE configs[2]) the ground truths of the resetting envs took as long as the
// step itself.
//
// One 256-thread workgroup per field; X (fp64, padded to NP = 16 TT) lives in LDS and is overwritten in place by every
// GEMM (all waves hold their output tiles in registers across a barrier).  Every GEMM has the form  Y = H X  with the
// wave owning (up to two) 16-row tiles of Y: its fragments of H -- H[k][16 w + (lane & 15)] for all k, 2 NP / 4 doubles
// per lane -- are loaded ONCE and stay in registers for all four GEMMs, so the GEMM loops read nothing but X from LDS,
// in the conflict-free operand form (4 rows of 16 consecutive doubles).  The right-hand multiplications come from
// storing every result TRANSPOSED (H is symmetric):  Y1 = H w,  Y2 = H Y1^T = (H w H)^T,  Y3 = H (amp .* Y2^T),
// Y4 = H Y3^T = field^T.
// v_mfma_f64_16x16x4_f64: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15], one f64 per lane;
// C/D[row = (lane >> 4) + 4 reg][col = lane & 15], 4 f64 per lane.
#pragma once
#include "ipp_common.h"

namespace ipp {

typedef double v4f64 __attribute__((ext_vector_type(4)));

__host__ __device__ inline size_t grf_hartley_lds_bytes(int tt) {
    const size_t np = 16 * (size_t)tt;
    return np * (np + 1) * 8 + 2 * 8 * 8 + 64;
}

// Y = H X for the row tiles {wave, wave + NW} of Y (NW waves per workgroup, OW = ceil(TT / NW) <= 2 owned tiles):
// hreg[o][ks] = H[16 (wave + NW o) + (lane & 15)][4 ks + (lane >> 4)]
// (A fragments; H symmetric, so loaded as row k of H at 16 consecutive columns), X in LDS with leading dimension ld.
// KS: steps of 4 along the contraction index, ceil(n / 4) <= 4 TT: the rows / columns n .. NP-1 of H and X are zero padding, and
// the bound is a template parameter (a run-time bound -- a uniform branch per step -- broke the schedule: 64x64 0.10 -> 0.13 ms):
// n = 100 contracts over 100 instead of 112 (-11 % of the MFMAs), n = 50 over 52 instead of 64 (-19 %)
template <int TT, int NW, int OW, int KS>
__device__ __forceinline__ void grf_gemm(const double (&hreg)[OW][KS], const double* X, int ld, int wave, int lane,
                                         v4f64 (&acc)[OW][TT]) {
    const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
    for (int o = 0; o < OW; ++o)
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[o][t] = (v4f64){0.0, 0.0, 0.0, 0.0};
    const bool two = OW > 1 && wave + NW < TT;  // second owned tile (wave-uniform)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {  // (fully unrolled: hreg needs static indices)
        const double* xr = X + (4 * ks + l4) * ld + l15;  // B[k][col]: 4 rows x 16 consecutive columns
        double xf[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) xf[t] = xr[16 * t];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            acc[0][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(hreg[0][ks], xf[t], acc[0][t], 0, 0, 0);
            if constexpr (OW > 1)
                if (two) acc[1][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(hreg[1][ks], xf[t], acc[1][t], 0, 0, 0);
        }
    }
}

// X <- Y^T (.* scale): accumulator element r of tile (o, t) is Y[16 (wave + NW o) + (lane >> 4) + 4 r][16 t + (lane & 15)]
template <int TT, int NW, int OW>
__device__ __forceinline__ void grf_store_t(double* X, int ld, int wave, int lane, const v4f64 (&acc)[OW][TT],
                                            const double* __restrict__ scale) {
    constexpr int NP = 16 * TT;
#pragma unroll
    for (int o = 0; o < OW; ++o) {
        if (wave + NW * o >= TT) break;
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (wave + NW * o) + (lane >> 4) + 4 * r, col = 16 * t + (lane & 15);
                X[col * ld + row] = scale ? acc[o][t][r] * scale[(size_t)col * NP + row] : acc[o][t][r];
            }
    }
}

template <int TT, int NW, int KS = 4 * TT>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 3 : 1))) void k_grf_hartley(View v, const int* __restrict__ env_ids, int n_items,
                                                     const float* __restrict__ white, const double* __restrict__ Hp,
                                                     const double* __restrict__ ampp, float* __restrict__ gt_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gh[];
    constexpr int NP = 16 * TT, LD = NP + 1, NT = 64 * NW, OW = (TT + NW - 1) / NW;
    static_assert(OW <= 2, "at most two owned tiles per wave");
    const int n = v.W, N = v.N;
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const int env = gt_out ? 0 : (env_ids ? env_ids[item] : item);
    if (env < 0 || env >= v.cap) return;
    double* X = reinterpret_cast<double*>(smem_gh);
    double* red = X + (size_t)NP * LD;  // [2][8] min / max per wave
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const float* __restrict__ wn = white + (size_t)item * N;
    // white noise -> fp64 in LDS: pairs of cells (n is even, so a pair never straddles a row), zero padding to NP x NP
    {
        const float2* __restrict__ w2 = reinterpret_cast<const float2*>(wn);
        const int half = n / 2, n2 = n * half;
#pragma unroll 4
        for (int i = tid; i < n2; i += NT) {
            const float2 wv = w2[i];
            const int y = i / half, x = 2 * (i - y * half);
            X[y * LD + x] = (double)wv.x;
            X[y * LD + x + 1] = (double)wv.y;
        }
        const int padc = NP - n;  // columns n .. NP-1 of the rows < n, then whole rows n .. NP-1
        for (int i = tid; i < n * padc; i += NT) X[(i / padc) * LD + n + i % padc] = 0.0;
        for (int i = tid; i < padc * NP; i += NT) X[(n + i / NP) * LD + i % NP] = 0.0;
    }
    // this wave's fragments of H, once for the four GEMMs
    double hreg[OW][KS];
    {
        const int l15 = lane & 15, l4 = lane >> 4;
#pragma unroll
        for (int o = 0; o < OW; ++o) {
            const int c = 16 * min(wave + NW * o, TT - 1) + l15;  // (waves beyond TT own nothing: their results are dropped)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) hreg[o][ks] = Hp[(size_t)(4 * ks + l4) * NP + c];
        }
    }
    __syncthreads();
    v4f64 acc[OW][TT];
    grf_gemm<TT, NW, OW, KS>(hreg, X, LD, wave, lane, acc);        // Y1 = H w
    __syncthreads();
    grf_store_t<TT, NW, OW>(X, LD, wave, lane, acc, nullptr);  // X = Y1^T
    __syncthreads();
    grf_gemm<TT, NW, OW, KS>(hreg, X, LD, wave, lane, acc);        // Y2 = H Y1^T = (H w H)^T
    __syncthreads();
    grf_store_t<TT, NW, OW>(X, LD, wave, lane, acc, ampp);     // X = amp .* (H w H)
    __syncthreads();
    grf_gemm<TT, NW, OW, KS>(hreg, X, LD, wave, lane, acc);        // Y3
    __syncthreads();
    grf_store_t<TT, NW, OW>(X, LD, wave, lane, acc, nullptr);
    __syncthreads();
    grf_gemm<TT, NW, OW, KS>(hreg, X, LD, wave, lane, acc);        // Y4 = field^T, in registers

    // ---- min-max normalisation to [0, 1] (ground_truths.py:31), fp64 like the reference
    double lo = INFINITY, hi = -INFINITY;
#pragma unroll
    for (int o = 0; o < OW; ++o) {
        if (wave + NW * o >= TT) break;
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (wave + NW * o) + (lane >> 4) + 4 * r, col = 16 * t + (lane & 15);
                if (row < n && col < n) { lo = fmin(lo, acc[o][t][r]); hi = fmax(hi, acc[o][t][r]); }
            }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, off));
        hi = fmax(hi, __shfl_xor(hi, off));
    }
    if (lane == 0) { red[wave] = lo; red[8 + wave] = hi; }
    __syncthreads();
    double dlo = red[0], dhi = red[8];
#pragma unroll
    for (int w = 1; w < NW; ++w) { dlo = fmin(dlo, red[w]); dhi = fmax(dhi, red[8 + w]); }
    const double span = dhi - dlo;
    float* gt = gt_out ? gt_out + (size_t)item * N : gt_plane(v, env);
#pragma unroll
    for (int o = 0; o < OW; ++o) {
        if (wave + NW * o >= TT) break;
#pragma unroll
        for (int t = 0; t < TT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * (wave + NW * o) + (lane >> 4) + 4 * r, col = 16 * t + (lane & 15);
                if (row < n && col < n) gt[col * n + row] = (float)((acc[o][t][r] - dlo) / span);  // (acc holds field^T)
            }
    }
    if (!gt_out)
        for (int i = N + tid; i < v.Npad; i += NT) gt[i] = 0.f;
}

}  // namespace ipp
