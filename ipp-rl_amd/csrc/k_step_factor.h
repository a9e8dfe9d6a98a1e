// Fused factor-state env step: ONE kernel, one 256-thread workgroup per item.
//   phase A  prepare_item (k_prepare.h): footprint, observation, S, Cholesky, L^-1, y, Q -- Q is produced
//            directly in LDS (in place over the gathered HT rows), nothing round-trips through global scratch
//   phase B  gain_tiles (k_gain_factor.h): prior term + streaming of the stored columns of U + epilogue
// Phase A is latency-bound (dependent loads, fp64 9x9 algebra) and phase B is HBM-bound; with several
// workgroups resident per CU one item's prologue runs under other items' streams, which a separate prologue
// kernel cannot do.  The prologue's fp64 scratch aliases the prior table, which is built
// after phase A.
#pragma once
#include "ipp_common.h"
#include "k_gain_factor.h"
#include "k_prepare.h"

namespace ipp {

constexpr int kStepThreads = 256;

template <int MC>
__host__ __device__ constexpr int step_scratch_floats(int lut_cap) {
    return (int)(((prep_small_bytes<MC>() + 15) / 16 * 4) > (size_t)lut_cap ? ((prep_small_bytes<MC>() + 15) / 16 * 4) : (size_t)lut_cap);
}

template <int MC, int VEC>
__global__ __launch_bounds__(kStepThreads, IPP_GF_MINWAVES) void k_step_factor(
    View v, const int* __restrict__ env_ids, int n_items, const double* __restrict__ action,
    const double* __restrict__ prev_action, const float* __restrict__ meas_noise, unsigned flags, int lut_cap,
    int* __restrict__ status_out, float* __restrict__ reward_out) {
    constexpr int QS = (MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sf[];
    const GainLds<MC> lds(smem_sf, v.rank_cap, step_scratch_floats<MC>(lut_cap));
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const int tid = threadIdx.x;

    // ---- phase A: HT rows are gathered into the Q area as HT(i,k) = Qs[k*QS + i] and overwritten by Q row k
    unsigned char* small = reinterpret_cast<unsigned char*>(lds.lut);
    ItemHdr* hs = prepare_item<MC, IPP_FACTOR, kStepThreads>(v, item, env_ids, nullptr, action, prev_action, meas_noise,
                                                              flags, status_out, nullptr, nullptr, nullptr, small,
                                                              lds.Qs, 1, QS, lds.Qs, lds.Ls, nullptr, lds.ys, nullptr,
                                                              lds.span_s);
    __syncthreads();
    const ItemHdr h = *hs;  // registers: the scratch that holds it is about to become the prior table
    if (h.m == 0 || h.status == IPP_STATUS_NOT_PD) {
        if (tid == 0) reward_out[item] = (h.status == IPP_STATUS_NOT_PD) ? NAN : 0.f;
        return;
    }
    __syncthreads();
    if (tid == 0) { *lds.next_tile = 0; *lds.done_waves = 0; }
    const bool use_lut = v.N <= lut_cap;
    if (use_lut) {
        const float s3 = (float)(kSqrt3 * v.res) / h.ls;
        for (int i = tid; i < v.N; i += kStepThreads) {
            const int dr = i / v.W, dc = i - dr * v.W;
            lds.lut[i] = matern_f(dr, dc, s3, h.sv);
        }
    }
    __syncthreads();

    // ---- phase B
    gain_tiles<MC, VEC>(v, h, item, flags, use_lut, lds, reward_out);
}

}  // namespace ipp
