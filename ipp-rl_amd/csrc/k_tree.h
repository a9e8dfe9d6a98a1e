// Tree-search step on path-local factor columns (planning/mcts_zero/mcts.py:166-265 descends
// <= episode_horizon levels per simulation, every level one covariance-only predict step from the parent's state,
// and keeps the child's covariance as the new node's state).
//
// A node does not copy its parent: it stores only the m <= MC columns its own step appended and its diagonal, both on
// the tile span of its step only (TreeView).  The state of a node is  P_root - sum over the path's nodes of C_n C_n^T, so a step streams the
// root env's columns followed by the column blocks of the path (ChainCols) -- the root slab is shared by all the
// simulations below it (L2 hits), a node costs (MC + 1) rows instead of a slot copy.
// Same phases as k_step_factor (k_step_factor.h); the root env slot is never written.
#pragma once
#include "k_step_factor.h"

namespace ipp {

// Node storage is COMPACT: a node's step changes the state only on the tiles [t_lo, t_hi] of its window (at most
// View::win_tiles of them), so the node keeps its <= MC columns and its diagonal on those tiles only -- 0.32 MB instead of
// 1.6 MB per node at 200x200.  The diagonal of a node's state on any other tile is that of the nearest ancestor whose
// span covers the tile, or the root env's (diag_source below).
struct TreeView {
    float* node_cov;   // [node_cap][MC][win_cells] columns appended by the node's step, cell c at index c - t_lo * tile_cells
    float* node_diag;  // [node_cap][win_cells]     diag of the node's state on its span
    int* node_meta;    // [node_cap][kNodeMeta]     m, tile span (lo | hi << 16), parent node (-1: child of the root env), root env, rectangle (rect_pack)
    int node_cap;
    int win_cells;     // View::win_tiles * View::tile_cells
};

// The chained state of an item (every thread builds the same, wave-uniform description): column blocks and diagonal
// sources of the path's nodes behind the root env's.  Returns the column count of the state; parent_id = the deepest
// path node (-1: the item steps from the root env itself).
template <int MC>
__device__ __forceinline__ int tree_chain(const View& v, const TreeView& tv, const int item, const int* __restrict__ root_ids,
                                          const int* __restrict__ path_ids, ChainCols& cc, DiagChain& dc, int& root, int& parent_id) {
    root = min(max(root_ids[item], 0), v.cap - 1);  // (a bad id is reported by the prologue's slot check)
    cc.root = v.cov + (size_t)root * v.cov_slot;
    cc.root_spans = v.colspan + (size_t)root * v.rank_cap;
    cc.root_rects = v.colrect + (size_t)root * v.rank_cap;
    cc.r_root = uni(v.rank[root]);
    cc.depth = 0;
    cc.npad = (size_t)v.Npad;
    cc.nstride = (size_t)tv.win_cells;
    int n_cols = cc.r_root;
    dc.root_diag = v.diag + (size_t)root * v.Npad;
    dc.depth = 0;
    parent_id = -1;
#pragma unroll
    for (int j = 0; j < kTreeDepth; ++j) {
        cc.node[j] = cc.root; cc.off[j] = 0x7fffffff; cc.nspan[j] = 0; cc.nrect[j] = kRectFull;
        dc.node[j] = dc.root_diag; dc.nspan[j] = 0xffff; dc.nrect[j] = kRectFull;  // (lo 0xffff > hi 0: covers nothing)
    }
    // Two rounds of loads (all the path ids, then all the nodes' records) in front of branch-free bookkeeping: behind
    // a per-level `if (id valid)` they are 2 kTreeDepth dependent scalar round trips.
    int pid[kTreeDepth], pm[kTreeDepth], psp[kTreeDepth], prc[kTreeDepth];
#pragma unroll
    for (int j = 0; j < kTreeDepth; ++j) pid[j] = uni(path_ids[(size_t)item * kTreeDepth + j]);
#pragma unroll
    for (int j = 0; j < kTreeDepth; ++j) {
        const int idc = min(max(pid[j], 0), tv.node_cap - 1);
        pm[j] = uni(tv.node_meta[kNodeMeta * idc]);
        psp[j] = uni(tv.node_meta[kNodeMeta * idc + 1]);
        prc[j] = uni(tv.node_meta[kNodeMeta * idc + 4]);
    }
#pragma unroll
    for (int j = 0; j < kTreeDepth; ++j) {
        const int id = pid[j];
        if (id >= 0 && id < tv.node_cap) {
            const int sp = psp[j];
            const size_t shift = (size_t)(sp & 0xffff) * v.tile_cells;  // first cell of the node's span
#pragma unroll
            for (int d = 0; d < kTreeDepth; ++d)  // (static indices only: cc / dc live in registers)
                if (d == cc.depth) {
                    cc.node[d] = tv.node_cov + (size_t)id * MC * tv.win_cells - shift;
                    cc.off[d] = n_cols;
                    cc.nspan[d] = sp;
                    cc.nrect[d] = (unsigned)prc[j];
                    dc.node[d] = tv.node_diag + (size_t)id * tv.win_cells - shift;
                    dc.nspan[d] = sp;
                    dc.nrect[d] = (unsigned)prc[j];
                }
            n_cols += pm[j];
            cc.depth += 1;
            dc.depth += 1;
            parent_id = id;
        }
    }
    return n_cols;
}

template <int MC, int VEC, bool RECT = false>
__global__ __launch_bounds__(kStepThreads, IPP_GF_MINWAVES) void k_tree_step(
    View v, TreeView tv, const int* __restrict__ root_ids, const int* __restrict__ path_ids,
    const int* __restrict__ new_ids, int n_items, const double* __restrict__ action, const double* __restrict__ prev_action,
    unsigned flags, int lut_rows, int* __restrict__ status_out, float* __restrict__ reward_out) {
    constexpr int QS = (MC + 3) & ~3;
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tr[];
    const GainLds<MC> lds(smem_tr, v.rank_cap, step_work_floats<MC>(v.rank_cap), lut_rows * v.W, step_small_floats<MC>(),
                          kStepThreads / kWave, v.win_tiles, 0, VEC);
    if ((int)blockIdx.x >= n_items) return;
    const int item = xcd_item(blockIdx.x, n_items);
    const int tid = threadIdx.x;
    if (tid == 0) IPP_MARK(item, 0);

    ChainCols cc;
    DiagChain dc;
    int root, parent_id;
    const int n_cols = tree_chain<MC>(v, tv, item, root_ids, path_ids, cc, dc, root, parent_id);
    for (int k = tid; k < min(n_cols, v.rank_cap); k += kStepThreads) lds.rowp[k] = cc.row(k);  // (published by the prologue's barriers)
    const int new_id = new_ids ? uni(new_ids[item]) : -1;
    const bool expand = new_id >= 0 && new_id < tv.node_cap && !(flags & IPP_PREDICT_ONLY);
    const unsigned flags_eff = (flags | IPP_COV_ONLY | (expand ? 0u : (unsigned)IPP_PREDICT_ONLY)) & ~(unsigned)IPP_UPDATE_PREV;
    // (shifted by the new node's own t_lo once the header is known: in `mid` below / after the prologue)
    float* new_cols0 = expand ? tv.node_cov + (size_t)new_id * MC * tv.win_cells : nullptr;
    float* new_diag0 = expand ? tv.node_diag + (size_t)new_id * tv.win_cells : nullptr;
    int* new_meta = expand ? tv.node_meta + kNodeMeta * new_id : nullptr;

    // ---- phase A; under the footprint-dependent loads: block tables and the prior table
    auto mid = [&](const ItemHdr& hh) {
        if (tid == 0) { *lds.next_tile = 0; *lds.done_waves = 0; *lds.solve_flag = 0; lds.red[0] = 0.0; lds.red[1] = 0.0; }
        fill_block_tables<MC>(hh, lds.fb_yx, lds.fb_w);
        if (v.rect_meta)
            for (int k = tid; k < hh.rank; k += kStepThreads) lds.stage_rect(k, cc.rect(k));
        // (mask and the new node's diagonal per tile, from the parent state's diagonal read under the tile's stream, like
        // k_tree_gain: the pass over the span in front of the stream was 9 of the 36 us of this prologue)
        const float s3 = (float)(kSqrt3 * v.res) / hh.ls;
        for (int i = tid; i < lut_rows * v.W; i += kStepThreads) {
            const int dr = i / v.W, dc2 = i - dr * v.W;
            lds.lut[i] = matern_f(dr, dc2, s3, hh.sv);
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
    if (v.clip_cols) mark_inactive_columns(lds.span_s, lds.work, QS, h.rank, h.m, tid, kStepThreads);  // (published by the barrier below)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_s_dcache_inv();
    if (tid == 0) IPP_MARK(item, 1);
    if (tid < kWave) {
        int status;
        if constexpr (MC == 9) status = solve_wave_fast<MC>(v, h, item, flags_eff, lds.small, lds.work, 1, QS, lds.Ls, lds.ys, nullptr, status_out);
        else status = solve_wave<MC>(v, h, item, flags_eff, lds.small, lds.work, lds.Ls, lds.ys, status_out);
        if (tid == 0) IPP_MARK(item, 7);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (tid == 0) __hip_atomic_store(lds.solve_flag, status == IPP_STATUS_NOT_PD ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the new node's blocks, indexed with absolute cells from here on
    float* new_cols = expand ? new_cols0 - (size_t)h.t_lo * v.tile_cells : nullptr;
    float* new_diag = expand ? new_diag0 - (size_t)h.t_lo * v.tile_cells : nullptr;
    if (expand && tid == 0) { new_meta[2] = parent_id; new_meta[3] = root; }
    gain_tiles<MC, VEC, sf_pipe<MC, VEC>(), true, false, true, true, false, 0, RECT>(v, h, item, flags_eff, lut_rows, lds, qrows_w,
                                                          reward_out, &cc, new_cols, new_diag, new_meta, nullptr, &dc);
}

// ---- the same step as two launches (configs[4]-sized waves): in k_tree_step about 40 % of a workgroup's life is the
// latency-bound prologue, during which its registers and LDS hold no stream.  k_tree_prepare runs the prologue and the
// m x m algebra for all items at the prologue's own (higher) occupancy and leaves header | L^-1 | Q rows in the item
// scratch; k_tree_gain streams.  Same arithmetic as the fused kernel except that L^-1 is folded into Q before the
// stream (like k_prepare / k_gain_factor) instead of applied per tile: results agree to fp32 rounding, not bit for bit.
template <int MC>
__global__ __launch_bounds__(kPrepThreads, IPP_PREP_MINWAVES) void k_tree_prepare(
    View v, TreeView tv, const int* __restrict__ root_ids, const int* __restrict__ path_ids, const int* __restrict__ new_ids,
    int n_items, const double* __restrict__ action, const double* __restrict__ prev_action, unsigned flags,
    int* __restrict__ status_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tp[];
    if ((int)blockIdx.x >= n_items) return;
    const int item = xcd_item(blockIdx.x, n_items);
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    ChainCols cc;
    DiagChain dc;
    int root, parent_id;
    const int n_cols = tree_chain<MC>(v, tv, item, root_ids, path_ids, cc, dc, root, parent_id);
    const int new_id = new_ids ? uni(new_ids[item]) : -1;
    const bool expand = new_id >= 0 && new_id < tv.node_cap && !(flags & IPP_PREDICT_ONLY);
    const unsigned flags_eff = (flags | IPP_COV_ONLY | (expand ? 0u : (unsigned)IPP_PREDICT_ONLY)) & ~(unsigned)IPP_UPDATE_PREV;
    float* blk_out = v.q + (size_t)item * v.q_item;  // [L^-1 | y | pad | Q rows | zero rows]
    float* big = reinterpret_cast<float*>(smem_tp + ((prep_small_bytes<MC>() + 15) & ~(size_t)15));
    if constexpr (MC != 9) {
        // (m up to 25: the whole prologue by the workgroup, m x m algebra in LDS -- the register form of solve_wave_fast is written for MC = 9;
        // writes L^-1 | y and the Q rows of the chained state into the item's block itself)
        prepare_item_ex<MC, IPP_FACTOR, kPrepThreads, false, NoMidWork, true>(
            v, item, root_ids, nullptr, action, prev_action, nullptr, flags_eff, status_out, nullptr, nullptr, nullptr, smem_tp, big, 0, 1,
            blk_out + LQ, v.linv + (size_t)item * MC * MC, blk_out, v.yv + (size_t)item * MC, blk_out + MC * MC, nullptr, NoMidWork(), &cc,
            n_cols);
        return;
    }
    ItemHdr* hs = prepare_item_ex<MC, IPP_FACTOR, kPrepThreads, true, NoMidWork, true>(
        v, item, root_ids, nullptr, action, prev_action, nullptr, flags_eff, status_out, nullptr, nullptr, nullptr, smem_tp, big, 0, 1,
        nullptr, v.linv + (size_t)item * MC * MC, blk_out, v.yv + (size_t)item * MC, blk_out + MC * MC, nullptr, NoMidWork(), &cc,
        n_cols);
    if (threadIdx.x >= kWave) return;
    const ItemHdr h = uniform_hdr(*hs);
    if (h.m == 0) return;
    const PrepLds<MC> pl(smem_tp);
    float* Ls = reinterpret_cast<float*>(pl.L);  // (the L scratch is free: 90 doubles >= 81 + 9 floats)
    float* ys = Ls + MC * MC;
    if constexpr (MC == 9) {
        solve_wave_fast<MC>(v, h, item, flags_eff, smem_tp, big, (h.rank + 3) & ~3, 1, Ls, ys, blk_out + LQ, status_out);
        wave_lds_sync();
        const int lane = threadIdx.x;
        for (int i = lane; i < MC * MC; i += kWave) blk_out[i] = Ls[i];
        if (lane < MC) blk_out[MC * MC + lane] = ys[lane];
    }
}

template <int MC, int VEC, bool RECT = false>
__global__ __launch_bounds__(512, IPP_GF_MINWAVES) void k_tree_gain(
    View v, TreeView tv, const float* __restrict__ q_all, const int* __restrict__ root_ids, const int* __restrict__ path_ids,
    const int* __restrict__ new_ids, int n_items, unsigned flags, int lut_rows, float* __restrict__ reward_out) {
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tg[];
    const GainLds<MC> lds(smem_tg, v.rank_cap, 0, lut_rows * v.W, 0, blockDim.x / kWave, v.win_tiles, 0, VEC);
    if ((int)blockIdx.x >= n_items) return;
    const int item = xcd_item(blockIdx.x, n_items);
    const int tid = threadIdx.x, T = blockDim.x;
    __builtin_amdgcn_s_dcache_inv();  // (Q through the non-coherent scalar cache: nothing of an earlier launch may be served)
    const ItemHdr h = uniform_hdr(v.hdr[item]);
    const int r = h.rank;
    if (h.m == 0 || h.status == IPP_STATUS_NOT_PD) {
        if (tid == 0) reward_out[item] = (h.status == IPP_STATUS_NOT_PD) ? NAN : 0.f;
        return;
    }
    ChainCols cc;
    DiagChain dc;
    int root, parent_id;
    tree_chain<MC>(v, tv, item, root_ids, path_ids, cc, dc, root, parent_id);
    const int new_id = new_ids ? uni(new_ids[item]) : -1;
    const bool expand = new_id >= 0 && new_id < tv.node_cap && !(flags & IPP_PREDICT_ONLY);
    const unsigned flags_eff = (flags | IPP_COV_ONLY | (expand ? 0u : (unsigned)IPP_PREDICT_ONLY)) & ~(unsigned)IPP_UPDATE_PREV;
    float* new_cols0 = expand ? tv.node_cov + (size_t)new_id * MC * tv.win_cells : nullptr;
    float* new_diag0 = expand ? tv.node_diag + (size_t)new_id * tv.win_cells : nullptr;
    int* new_meta = expand ? tv.node_meta + kNodeMeta * new_id : nullptr;

    const float* __restrict__ blk = q_all + (size_t)item * v.q_item;  // [L^-1 | y | pad | Q rows | zero rows]
    for (int i = tid; i < LQ; i += T) lds.Ls[i] = blk[i];
    if (tid == 0) { *lds.next_tile = 0; *lds.done_waves = 0; lds.red[0] = 0.0; lds.red[1] = 0.0; }
    for (int k = tid; k < r; k += T) { lds.span_s[k] = cc.span(k); lds.rowp[k] = cc.row(k); }
    if (v.rect_meta)
        for (int k = tid; k < r; k += T) lds.stage_rect(k, cc.rect(k));
    if (v.clip_cols) {
        __syncthreads();
        mark_inactive_columns(lds.span_s, blk + LQ, (MC + 3) & ~3, r, h.m, tid, T);
    }
    fill_block_tables<MC>(h, lds.fb_yx, lds.fb_w);
    {
        const float s3 = (float)(kSqrt3 * v.res) / h.ls;
        for (int i = tid; i < lut_rows * v.W; i += T) {
            const int dr = i / v.W, dc2 = i - dr * v.W;
            lds.lut[i] = matern_f(dr, dc2, s3, h.sv);
        }
    }
    if (expand && tid == 0) { new_meta[2] = parent_id; new_meta[3] = root; }
    __syncthreads();
    float* new_cols = expand ? new_cols0 - (size_t)h.t_lo * v.tile_cells : nullptr;
    float* new_diag = expand ? new_diag0 - (size_t)h.t_lo * v.tile_cells : nullptr;
    // (mask and the new node's diagonal per tile, from the parent state's diagonal read under the tile's stream: no
    // phase-A pass over the span)
    gain_tiles<MC, VEC, gf_pipe<MC, VEC>(), false, false, true, false, false, 0, RECT>(v, h, item, flags_eff, lut_rows, lds, blk + LQ, reward_out, &cc, new_cols,
                                                       new_diag, new_meta, nullptr, &dc);
}

// diag of a node's state, assembled along its parent chain: out[c] = diagonal of the deepest node (from `node` upwards)
// whose span (and rectangle) covers c, else the root env's.
__global__ void k_tree_read_diag(View v, TreeView tv, int node, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= v.N) return;
    const int tile = c / v.tile_cells;
    int cur = node, root = tv.node_meta[kNodeMeta * node + 3];
    float val = 0.f;
    bool found = false;
    for (int hops = 0; hops <= kTreeDepth && cur >= 0; ++hops) {
        const int sp = tv.node_meta[kNodeMeta * cur + 1], lo = sp & 0xffff, hi = sp >> 16;
        if (tile >= lo && tile <= hi && rect_has((unsigned)tv.node_meta[kNodeMeta * cur + 4], c / v.W, c % v.W)) {
            val = tv.node_diag[(size_t)cur * tv.win_cells + (c - lo * v.tile_cells)];
            found = true;
            break;
        }
        cur = tv.node_meta[kNodeMeta * cur + 2];
    }
    out[c] = found ? val : v.diag[(size_t)root * v.Npad + c];
}

}  // namespace ipp
