// Fused factor-state env step: ONE kernel, one 256-thread workgroup per item.
//   phase A  prepare_item_ex<FRONT_ONLY> (k_prepare.h): footprint, observation, gather of HT = H_F U[F,:]^T into LDS;
//            -HT (17 KB per item) is written to the item's global scratch block, from where phase B reads it back
//            through the scalar cache
//   then     wave 0: S, Cholesky, L^-1, y (solve_wave); waves 1..3 go straight to
//   phase B  gain_tiles<PRE> (k_gain_factor.h): prior term + streaming of the stored columns of U against -HT,
//            L^-1 applied in the tile epilogue, fused reward / diag / mean / append
// Under the streaming load every dependent memory round trip of the prologue costs several microseconds and the
// m x m algebra another ~20 us, during which a workgroup holds its registers without streaming; starting the
// stream before the algebra is done and filling the load latency with the table builds shortens that window.
// With several workgroups resident per CU one item's prologue also runs under other items' streams, which a
// separate prologue kernel cannot do.
#pragma once
#ifndef IPP_SF_LMASK
// 1: adaptive mask of the step's window built in phase A (one pass over mean / diag of the window in front of the stream,
// their updates as no-return atomics); 0: per tile, from the tile's mean / diag loaded under its stream and written back with
// plain stores.  With the two-dimensional windows a tile streams ~30 rows instead of ~80 and the pass in phase A is the
// larger cost: 17.6 -> 18.0 M env-steps/s with 0 (and no second read of mean / diag)
#define IPP_SF_LMASK 0
#endif
#include "ipp_common.h"
#include "k_gain_factor.h"
#include "k_prepare.h"

namespace ipp {

constexpr int kStepThreads = 256;

// LDS work area of the fused kernel: the HT staging rows (rank_cap + 8 rows of QS floats).
template <int MC>
__host__ __device__ constexpr int step_work_floats(int rank_cap) {
    return (rank_cap + 8) * ((MC + 3) & ~3);
}
template <int MC>
__host__ __device__ constexpr int step_small_floats() {
    return (int)((prep_small_bytes<MC>() + 15) / 16 * 4);
}

// Phase A writes the item's -HT rows to its block of v.q (vector stores, complete at the workgroup barrier below:
// __syncthreads waits for vmcnt(0) and the stores are write-through to L2); the tile loop reads them back through
// the constant address space (gain_tiles<QCONST>), i.e. with scalar loads.  The two never overlap in time within a
// workgroup, no other workgroup touches this item's block, blocks are 64-byte aligned (no scalar-cache line shared
// between items) and the scalar cache is invalidated at every kernel launch, so it cannot hold lines of a previous
// step.
template <int MC, int VEC, bool RECT = false>
__global__ __launch_bounds__(kStepThreads, IPP_GF_MINWAVES) void k_step_factor(
    View v, const int* __restrict__ env_ids, int n_items,
    const double* __restrict__ action, const double* __restrict__ prev_action, const float* __restrict__ meas_noise,
    unsigned flags, int lut_rows, int* __restrict__ status_out, float* __restrict__ reward_out, AutoReset ar) {
    constexpr int QS = (MC + 3) & ~3;
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sf[];
    const GainLds<MC> lds(smem_sf, v.rank_cap, step_work_floats<MC>(v.rank_cap), lut_rows * v.W, step_small_floats<MC>(),
                          kStepThreads / kWave, v.win_tiles, IPP_SF_LMASK ? v.win_tiles * kWave : 0, VEC);
    if ((int)blockIdx.x >= n_items) return;
    const int item = launch_item(v, blockIdx.x, n_items);
    const int tid = threadIdx.x;
    if (tid == 0) IPP_MARK(item, 0);

    // ---- phase A: header, gather of HT(i,k) = work[k*QS + i].  While the footprint-dependent loads are
    // in flight the workgroup builds the prior table and the block tables (pure arithmetic on the header).
    // MC = 9: the observation is deferred (prepare_item_ex<DEFER>): wave 1 holds its inputs and evaluates it after the
    // barrier below, while wave 0 factors S and waves 2, 3 already stream; obs_flag publishes z / the innovation.
    constexpr bool kDeferObs = (MC == 9);
    int* obs_flag = reinterpret_cast<int*>(lds.red + 13);  // (red[8..15] are unused by the reduction)
    ObsRegs oregs;
    auto mid = [&](const ItemHdr& hh) {
        if (tid == 0) { *lds.next_tile = 0; *lds.done_waves = 0; *lds.solve_flag = 0; *obs_flag = 0; lds.red[0] = 0.0; lds.red[1] = 0.0; }
        fill_block_tables<MC>(hh, lds.fb_yx, lds.fb_w);
        if (v.rect_meta)  // rectangles of the stored columns (their loads return right behind the gather's first pass)
            for (int k = tid; k < hh.rank; k += kStepThreads) lds.stage_rect(k, (unsigned)v.colrect[(size_t)hh.env * v.rank_cap + k]);
        // adaptive mask of the whole env (rewards.py:11: pre-step mean and pre-step diag), one byte per VEC cells: the
        // tile loop then needs neither mean nor diag (their updates are no-return atomics)
        typedef float cellv __attribute__((ext_vector_type(VEC)));
        const cellv* mean_v = reinterpret_cast<const cellv*>(v.mean + (size_t)hh.env * v.Npad);
        const cellv* diag_v = reinterpret_cast<const cellv*>(v.diag + (size_t)hh.env * v.Npad);
        const bool adaptive = (flags & IPP_ADAPTIVE) != 0;
        constexpr int kMaskPerThread = IPP_SF_LMASK ? 4 : 0;  // groups in flight per thread and pass
        // only the tiles this step touches: [t_lo, t_hi] (the span of the appended columns)
        const int q_lo = hh.t_lo * kWave, q_hi = (hh.t_hi + 1) * kWave;  // groups of VEC cells
        for (int q0 = q_lo + tid; IPP_SF_LMASK && q0 < q_hi; q0 += kMaskPerThread * kStepThreads) {
            cellv mu[kMaskPerThread + 1], dg[kMaskPerThread + 1];
#pragma unroll
            for (int u = 0; u < kMaskPerThread; ++u) {
                const int q = q0 + u * kStepThreads;
                if (adaptive && q < q_hi) { mu[u] = mean_v[q]; dg[u] = diag_v[q]; }
            }
#pragma unroll
            for (int u = 0; u < kMaskPerThread; ++u) {
                const int q = q0 + u * kStepThreads;
                if (q >= q_hi) continue;
                unsigned bits = 0;
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    const bool in = !adaptive || ((double)mu[u][c] + v.kf * (double)dg[u][c] >= v.thr);
                    bits |= (in ? 1u : 0u) << c;
                }
                lds.mask4[q - q_lo] = (unsigned char)bits;
            }
        }
        const float s3 = (float)(kSqrt3 * v.res) / hh.ls;
        for (int i = tid; i < lut_rows * v.W; i += kStepThreads) {
            const int dr = i / v.W, dc = i - dr * v.W;
            lds.lut[i] = matern_f(dr, dc, s3, hh.sv);
        }
    };
    ItemHdr* hs = prepare_item_ex<MC, IPP_FACTOR, kStepThreads, true, decltype(mid), false, false, kDeferObs>(
        v, item, env_ids, nullptr, action, prev_action, meas_noise, flags, status_out, nullptr, nullptr, nullptr, lds.small,
        lds.work, 1, QS, nullptr, lds.Ls, nullptr, lds.ys, nullptr, lds.span_s, mid, nullptr, 0, &oregs);
    const ItemHdr h = uniform_hdr(*hs);  // (every path of the prologue ends with a barrier)
    IPP_TICK_DECL(tick);
    if (h.m == 0) {  // nothing to stream (bad footprint): no step, but a scheduled reset still happens
        if (tid == 0) reward_out[item] = 0.f;
        if (ar.src && tid < kWave) {
            const int k = __builtin_amdgcn_readfirstlane(ar.src[item]);
            if (k >= 0 && h.env >= 0 && h.env < v.cap) wave_reset_env(v, ar, h.env, k, tid);
        }
        return;
    }
    // -HT rows (sign of the downdate folded in, padding lanes zero) and the zero rows behind them -> global scratch
    float* qrows_w = v.q + (size_t)item * v.q_item + LQ;
    for (int idx = tid; idx < (h.rank + 8) * QS; idx += kStepThreads) {
        const int k = idx / QS, i = idx - k * QS;
        qrows_w[idx] = (k < h.rank && i < h.m) ? -lds.work[idx] : 0.f;
    }
    // The rows are read back through the scalar cache, i.e. from L2, not through the vector L1 the stores went
    // through.  __syncthreads() alone does not order that: at workgroup scope hipcc emits no vmcnt wait for global
    // stores on gfx950 (the waves of a workgroup share their vector L1, so the memory model needs none), and a scalar
    // load behind the barrier can reach L2 before a store that is still on its way: one wrong env in ~10^5 item steps
    // at 4096 envs (tests/test_hip_edge_cases.py, full-size fused-vs-exact test).  So: every wave waits for its own
    // stores to be acknowledged, then the barrier, then any line of this block left in the (non-coherent) scalar
    // cache by an earlier launch is dropped.
    if (v.clip_cols) mark_inactive_columns(lds.span_s, lds.work, QS, h.rank, h.m, tid, kStepThreads);  // (published by the barrier below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    IPP_TICK(v, 4, tick);
    if (tid == 0) IPP_MARK(item, 1);

    // ---- wave 0 finishes the m x m algebra while waves 1..3 already stream (they need L^-1 only in a tile epilogue)
    if (tid < kWave) {
        int status;
        if constexpr (MC == 9) status = solve_wave_fast<MC>(v, h, item, flags, lds.small, lds.work, 1, QS, lds.Ls, lds.ys, nullptr, status_out, obs_flag);
        else status = solve_wave<MC>(v, h, item, flags, lds.small, lds.work, lds.Ls, lds.ys, status_out);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (tid == 0) __hip_atomic_store(lds.solve_flag, status == IPP_STATUS_NOT_PD ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        IPP_TICK(v, 5, tick);
        if (tid == 0) IPP_MARK(item, 7);  // m x m algebra done
    } else if (kDeferObs && tid < 2 * kWave) {  // wave 1: the observation, then it streams like waves 2, 3
        if constexpr (kDeferObs) observe_wave<MC>(v, h, flags, lds.small, oregs);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (tid == kWave) __hip_atomic_store(obs_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    // ---- phase B
    gain_tiles<MC, VEC, sf_pipe<MC, VEC>(), true, IPP_SF_LMASK != 0, false, true, true, 0, RECT && !IPP_SF_LMASK>(v, h, item, flags, lut_rows, lds, qrows_w, reward_out, nullptr, nullptr,
                                                                nullptr, nullptr, &ar);
}

}  // namespace ipp
