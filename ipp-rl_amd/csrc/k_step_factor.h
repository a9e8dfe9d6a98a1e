// Fused factor-state env step: ONE kernel, one 256-thread workgroup per item.
//   phase A  prepare_item (k_prepare.h): footprint, observation, S, Cholesky, L^-1, y, Q -- the gathered HT rows
//            stay in LDS; Q (17 KB per item) is written to the item's global scratch block, from where phase B
//            reads it back through the scalar cache
//   phase B  gain_tiles (k_gain_factor.h): prior term + streaming of the stored columns of U + epilogue
// Phase A is latency-bound (dependent loads, fp64 9x9 algebra) and phase B is HBM-bound; with several
// workgroups resident per CU one item's prologue runs under other items' streams, which a separate prologue
// kernel cannot do.  The prior table is built over the HT staging rows after phase A.
#pragma once
#include "ipp_common.h"
#include "k_gain_factor.h"
#include "k_prepare.h"

namespace ipp {

constexpr int kStepThreads = 256;

// LDS work area of the fused kernel: HT staging rows during the prologue, then the prior table.
template <int MC>
__host__ __device__ constexpr int step_work_floats(int rank_cap, int lut_cap) {
    return ((rank_cap + 8) * ((MC + 3) & ~3)) > lut_cap ? ((rank_cap + 8) * ((MC + 3) & ~3)) : lut_cap;
}
template <int MC>
__host__ __device__ constexpr int step_small_floats() {
    return (int)((prep_small_bytes<MC>() + 15) / 16 * 4);
}

// q_ro == v.q.  The prologue writes the item's Q rows through v.q (vector stores, complete at the workgroup
// barrier below: __syncthreads waits for vmcnt(0) and the stores are write-through to L2), the tile loop reads
// them through q_ro, which the compiler may treat as read-only and therefore fetches with scalar loads.  The two
// never overlap in time within a workgroup, no other workgroup touches this item's block, blocks are 64-byte
// aligned (no scalar-cache line shared between items) and the scalar cache is invalidated at every kernel launch,
// so it cannot hold lines of a previous step.
template <int MC, int VEC>
__global__ __launch_bounds__(kStepThreads, IPP_GF_MINWAVES) void k_step_factor(
    View v, const float* __restrict__ q_ro, const int* __restrict__ env_ids, int n_items,
    const double* __restrict__ action, const double* __restrict__ prev_action, const float* __restrict__ meas_noise,
    unsigned flags, int lut_cap, int* __restrict__ status_out, float* __restrict__ reward_out) {
    constexpr int QS = (MC + 3) & ~3;
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sf[];
    const GainLds<MC> lds(smem_sf, v.rank_cap, step_work_floats<MC>(v.rank_cap, lut_cap), step_small_floats<MC>());
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const int tid = threadIdx.x;

    // ---- phase A: HT rows are gathered into the LDS work area as HT(i,k) = work[k*QS + i]; Q goes to global scratch
    float* qblk = v.q + (size_t)item * v.q_item;
    ItemHdr* hs = prepare_item<MC, IPP_FACTOR, kStepThreads>(v, item, env_ids, nullptr, action, prev_action, meas_noise,
                                                              flags, status_out, nullptr, nullptr, nullptr, lds.small,
                                                              lds.work, 1, QS, qblk + LQ, lds.Ls, nullptr, lds.ys, nullptr,
                                                              lds.span_s);
    __syncthreads();
    const ItemHdr h = *hs;
    if (h.m == 0 || h.status == IPP_STATUS_NOT_PD) {
        if (tid == 0) reward_out[item] = (h.status == IPP_STATUS_NOT_PD) ? NAN : 0.f;
        return;
    }
    if (tid == 0) { *lds.next_tile = 0; *lds.done_waves = 0; }
    fill_block_tables<MC>(h, lds.fb_yx, lds.fb_w);
    const bool use_lut = v.N <= lut_cap;
    if (use_lut) {  // over the HT staging rows: every thread is past its last read of them (barrier above)
        const float s3 = (float)(kSqrt3 * v.res) / h.ls;
        for (int i = tid; i < v.N; i += kStepThreads) {
            const int dr = i / v.W, dc = i - dr * v.W;
            lds.lut[i] = matern_f(dr, dc, s3, h.sv);
        }
    }
    __syncthreads();

    // ---- phase B
    gain_tiles<MC, VEC, IPP_SF_PIPE>(v, h, item, flags, use_lut, lds, q_ro + (size_t)item * v.q_item + LQ, reward_out);
}

}  // namespace ipp
