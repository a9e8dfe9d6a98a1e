// Tree-search step on path-local factor columns (planning/mcts_zero/mcts.py:166-265 descends
// <= episode_horizon levels per simulation, every level one covariance-only predict step from the parent's state,
// and keeps the child's covariance as the new node's state).
//
// A node does not copy its parent: it stores only the m <= MC columns its own step appended (on their tile span)
// and its diagonal.  The state of a node is  P_root - sum over the path's nodes of C_n C_n^T, so a step streams the
// root env's columns followed by the column blocks of the path (ChainCols) -- the root slab is shared by all the
// simulations below it (L2 hits), a node costs (MC + 1) rows instead of a slot copy.
// Same phases as k_step_factor (k_step_factor.h); the root env slot is never written.
#pragma once
#include "k_step_factor.h"

namespace ipp {

struct TreeView {
    float* node_cov;   // [node_cap][MC][Npad] columns appended by the node's step
    float* node_diag;  // [node_cap][Npad]     diag of the node's state
    int* node_meta;    // [node_cap][2]        m, tile span (lo | hi << 16)
    int node_cap;
};

template <int MC, int VEC>
__global__ __launch_bounds__(kStepThreads, IPP_GF_MINWAVES) void k_tree_step(
    View v, TreeView tv, const int* __restrict__ root_ids, const int* __restrict__ path_ids,
    const int* __restrict__ new_ids, int n_items, const double* __restrict__ action, const double* __restrict__ prev_action,
    unsigned flags, int lut_rows, int* __restrict__ status_out, float* __restrict__ reward_out) {
    constexpr int QS = (MC + 3) & ~3;
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tr[];
    const GainLds<MC> lds(smem_tr, v.rank_cap, step_work_floats<MC>(v.rank_cap), lut_rows * v.W, step_small_floats<MC>(),
                          kStepThreads / kWave, v.win_tiles, v.win_tiles * kWave);
    if ((int)blockIdx.x >= n_items) return;
    const int item = xcd_item(blockIdx.x, n_items);
    const int tid = threadIdx.x;

    // ---- the chained state of this item (every thread builds the same, wave-uniform description)
    const int root = min(max(root_ids[item], 0), v.cap - 1);  // (a bad id is reported by the prologue's slot check)
    ChainCols cc;
    cc.root = v.cov + (size_t)root * v.cov_slot;
    cc.root_spans = v.colspan + (size_t)root * v.rank_cap;
    cc.r_root = uni(v.rank[root]);
    cc.depth = 0;
    cc.npad = (size_t)v.Npad;
    int n_cols = cc.r_root;
    const float* parent_diag = v.diag + (size_t)root * v.Npad;
#pragma unroll
    for (int j = 0; j < kTreeDepth; ++j) {
        cc.node[j] = cc.root; cc.off[j] = 0x7fffffff; cc.nspan[j] = 0;
    }
#pragma unroll
    for (int j = 0; j < kTreeDepth; ++j) {
        const int id = uni(path_ids[(size_t)item * kTreeDepth + j]);
        if (id >= 0 && id < tv.node_cap) {
#pragma unroll
            for (int d = 0; d < kTreeDepth; ++d)  // (static indices only: cc lives in registers)
                if (d == cc.depth) {
                    cc.node[d] = tv.node_cov + (size_t)id * MC * v.Npad;
                    cc.off[d] = n_cols;
                    cc.nspan[d] = uni(tv.node_meta[2 * id + 1]);
                }
            n_cols += uni(tv.node_meta[2 * id]);
            cc.depth += 1;
            parent_diag = tv.node_diag + (size_t)id * v.Npad;
        }
    }
    const int new_id = new_ids ? uni(new_ids[item]) : -1;
    const bool expand = new_id >= 0 && new_id < tv.node_cap && !(flags & IPP_PREDICT_ONLY);
    const unsigned flags_eff = (flags | IPP_COV_ONLY | (expand ? 0u : (unsigned)IPP_PREDICT_ONLY)) & ~(unsigned)IPP_UPDATE_PREV;
    float* new_cols = expand ? tv.node_cov + (size_t)new_id * MC * v.Npad : nullptr;
    float* new_diag = expand ? tv.node_diag + (size_t)new_id * v.Npad : nullptr;
    int* new_meta = expand ? tv.node_meta + 2 * new_id : nullptr;

    // ---- phase A; under the footprint-dependent loads: tables, mask bits of the touched tiles (root mean, parent
    // diag), and the new node's diagonal starts as a copy of its parent's (the tile epilogues subtract from it)
    auto mid = [&](const ItemHdr& hh) {
        if (tid == 0) { *lds.next_tile = 0; *lds.done_waves = 0; *lds.solve_flag = 0; lds.red[0] = 0.0; lds.red[1] = 0.0; }
        fill_block_tables<MC>(hh, lds.fb_yx, lds.fb_w);
        typedef float cellv __attribute__((ext_vector_type(VEC)));
        const cellv* mean_v = reinterpret_cast<const cellv*>(v.mean + (size_t)hh.env * v.Npad);
        const cellv* diag_v = reinterpret_cast<const cellv*>(parent_diag);
        const bool adaptive = (flags & IPP_ADAPTIVE) != 0;
        for (int q = hh.t_lo * kWave + tid; q < (hh.t_hi + 1) * kWave; q += kStepThreads) {
            unsigned bits = (1u << VEC) - 1u;
            if (adaptive) {
                const cellv mu = mean_v[q], dg = diag_v[q];
                bits = 0;
#pragma unroll
                for (int c = 0; c < VEC; ++c) bits |= (((double)mu[c] + v.kf * (double)dg[c] >= v.thr) ? 1u : 0u) << c;
            }
            lds.mask4[q - hh.t_lo * kWave] = (unsigned char)bits;
        }
        if (expand) {
            const float4* src = reinterpret_cast<const float4*>(parent_diag);
            float4* dst = reinterpret_cast<float4*>(new_diag);
            for (int q0 = tid; q0 < v.Npad / 4; q0 += 4 * kStepThreads) {
                float4 t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int q = q0 + u * kStepThreads; t[u] = (q < v.Npad / 4) ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int q = q0 + u * kStepThreads; if (q < v.Npad / 4) dst[q] = t[u]; }
            }
        }
        const float s3 = (float)(kSqrt3 * v.res) / hh.ls;
        for (int i = tid; i < lut_rows * v.W; i += kStepThreads) {
            const int dr = i / v.W, dc = i - dr * v.W;
            lds.lut[i] = matern_f(dr, dc, s3, hh.sv);
        }
    };
    ItemHdr* hs = prepare_item_ex<MC, IPP_FACTOR, kStepThreads, true, decltype(mid), true>(
        v, item, root_ids, nullptr, action, prev_action, nullptr, flags_eff, status_out, nullptr, nullptr, nullptr, lds.small,
        lds.work, 1, QS, nullptr, lds.Ls, nullptr, lds.ys, nullptr, lds.span_s, mid, &cc, n_cols);
    const ItemHdr h = uniform_hdr(*hs);
    if (h.m == 0) {
        if (tid == 0) reward_out[item] = 0.f;
        return;
    }
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    if (tid < kWave) {
        int status;
        if constexpr (MC == 9) status = solve_wave_fast<MC>(v, h, item, flags_eff, lds.small, lds.work, 1, QS, lds.Ls, lds.ys, nullptr, status_out);
        else status = solve_wave<MC>(v, h, item, flags_eff, lds.small, lds.work, lds.Ls, lds.ys, status_out);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (tid == 0) __hip_atomic_store(lds.solve_flag, status == IPP_STATUS_NOT_PD ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    gain_tiles<MC, VEC, IPP_SF_PIPE, true, true, true, true>(v, h, item, flags_eff, lut_rows, lds, qrows_w,
                                                         reward_out, &cc, new_cols, new_diag, new_meta);
}

}  // namespace ipp
