// Tree-search predict step on COMPACT COLUMN PATCHES (the patch counterpart of k_tree.h; SURVEY 8(f) rank 1;
// planning/mcts_zero/mcts.py:166-265, planning/mcts_mission.py:228-230: every tree level is one covariance-only predict step
// from the parent node's state, whose result becomes the child's state).
//
// State of a node = the root env's columns + the <= MC columns of every node on the path (TreeView, k_tree.h); with
// View::patch every one of them is a patch of its rectangle: the root's in its env slot, a node's in
// node_cov[id][j][pstride], and the node's diagonal on its rectangle in node_diag[id][pstride] (elsewhere the diagonal is
// the deepest covering ancestor's, else the root env's).  The kernel is k_step_patch (k_step_patch.h) on that chain:
// one 2-wave workgroup per item, contributing columns compacted into LDS records, the row stream with DPP-broadcast
// coefficients -- with three differences: a column's patch address is a 64-bit offset from View::cov in units of 8 bytes
// (root slots and node storage are different regions of the arena), there is no observation / mean update (covariance
// only: the second wave starts streaming at once), and the results go to the NEW node's block; the root slot is never
// written.  Band-tile engines (scoring scratch, wide windows, odd grids) keep k_tree_step / k_tree_prepare + k_tree_gain.
#pragma once
#include "k_step_patch.h"
#include "k_tree.h"

namespace ipp {

constexpr long long kTreePatchGuard = 65536;  // bytes in front of View::cov that the shifted patch offsets may reach into (address math only)

// Io policy of the tree step for patch_units: the ROOT env's mean (rewards.py:11: the map's current mean), the PARENT state's variance
// (the deepest path node whose rectangle holds the cells -- column bounds are even, a lane's two cells are inside or outside together --
// else the root's), stored rows through a 64-bit offset from View::cov in 8-byte units (root slots and node blocks are different regions
// of the arena), results into the NEW node's block: its variance on its rectangle and its m columns.  Covariance only: no mean update.
struct TreeIo {
    static constexpr bool kMean = false;
    typedef float rowv __attribute__((ext_vector_type(2)));
    const char* base0;
    const float* mean_ro;
    const float* diag_root;
    const float* node_diag;
    float* new_cols;
    float* new_diag;
    unsigned prc[kTreeDepth];
    int pid[kTreeDepth];
    int m, pw, pstride;
    __device__ __forceinline__ rowv row_load(unsigned cofs8, unsigned voff) const {
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base0) + ((unsigned long long)cofs8 << 3), 0, 0x7ffffff0, 0x00020000);
        return __builtin_bit_cast(rowv, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, kPatchRowAux));
    }
    __device__ __forceinline__ float coef(const UnitLds& ul, int a, int l15) const { return ul.rec[(size_t)a * kPatchRec + l15]; }
    __device__ __forceinline__ void load_pre(int cell0, int flat, int rrow, int rcol, float (&md)[2][2]) const {
        load_vec<2>(mean_ro + cell0, md[0]);
        const float* src = diag_root + cell0;
#pragma unroll
        for (int d = 0; d < kTreeDepth; ++d) {
            const unsigned rc = prc[d];
            const int r0d = rc & 0xff, r1d = (rc >> 8) & 0xff, c0d = (rc >> 16) & 0xff, c1d = rc >> 24;
            if (rrow >= r0d && rrow <= r1d && rcol >= c0d && rcol <= c1d)
                src = node_diag + (size_t)pid[d] * pstride + (rrow - r0d) * pw + (rcol - c0d);
        }
        load_vec<2>(src, md[1]);
    }
    __device__ __forceinline__ void store(bool commit, bool lane_valid, int cell0, int flat, unsigned flat4, const float (&acc)[2][9],
                                          const float (&md)[2][2], const float (&dred)[2], const float (&dmean)[2]) const {
        if (!commit || !lane_valid) return;
        float outv[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) outv[c] = md[1][c] - dred[c];
        store_vec<2>(new_diag + flat, outv);
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (j < m) {
#pragma unroll
                for (int c = 0; c < 2; ++c) outv[c] = acc[c][j];
                store_stream<2>(new_cols + (size_t)j * pstride + flat, outv);
            }
    }
};

// The search driver's use of an item's reward (ipp_mcts_steps): the numerator of the tree edge that asked for the step,
// t_num[parent][k] = reward (cost + 1) (rewards.py:31 undone: the cost depends on the path that led to the node, the masked trace
// reduction does not), and a non-zero status into err[2] -- written by the kernel itself instead of by a launch behind it.
struct TreeEdgeOut {
    const int* parent; const int* k; const double* cost;  // [n_items] the edge (node, slot) and the cost the reward was divided by
    double* t_num; int* err; int kmax;                    // t_num == nullptr: off
    int item_base;  // this launch's items use the per-item scratch slots [item_base, item_base + n) of the engine (overflow records, byte
                    // counters): two searches stepped on two streams at once must not share them (ipp_mcts_tables.scratch_base)
};

template <int NW, int RJN = 0>  // (RJN: k_step_patch.h)
__global__ __launch_bounds__(64 * NW, kPatchMinW) void k_tree_patch(
    View v, TreeView tv, const int* __restrict__ root_ids, const int* __restrict__ path_ids, const int* __restrict__ new_ids,
    int n_items, const double* __restrict__ action, const double* __restrict__ prev_action, unsigned flags,
    int* __restrict__ status_out, float* __restrict__ reward_out, const int* __restrict__ n_dev, TreeEdgeOut eo) {
    // n_dev: the item count lives on the device (ipp_mcts_steps with n < 0: the search driver queues the levels of a
    // wave of simulations without reading their request counts back); the grid then has n_items >= *n_dev workgroups
    constexpr int MC = 9, VEC = 2, NT = kWave * NW, KP = kPatchKP;
    constexpr int RJ = RJN > 0 ? RJN : (kPatchMaxRank + NT - 1) / NT;
    static_assert(NW >= 2 && NW <= 4, "two to four waves per item");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_tp2[];
    const PatchLds lds(smem_tp2, v.pcap, v.plw * v.plw, NW, v.punits, v.rank_cap);
    if (n_dev) n_items = min(n_items, uni(*n_dev));
    if ((int)blockIdx.x >= n_items) return;
    const int item = xcd_item(blockIdx.x, n_items);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) IPP_MARK(item, 0);
    int* next_unit = lds.ctl; int* done_waves = lds.ctl + 1; int* solve_flag = lds.ctl + 2;
    int* wcnt = lds.ctl + 8;  // [RJ][NW]

    // ------------------------------------------------------------------ batch 1: root, path, inputs, the root's rectangles
    const int root0 = root_ids[item];
    const bool slots_ok = root0 >= 0 && root0 < v.cap;
    const int root = slots_ok ? root0 : 0;
    const double ax = action[3 * item + 0], ay = action[3 * item + 1], az = action[3 * item + 2];
    const double px = prev_action[3 * item + 0], py = prev_action[3 * item + 1], pz = prev_action[3 * item + 2];
    const int r_root = uni(v.rank[root]);
    const double sv_d = v.prior[2 * root + 0], ls_d = v.prior[2 * root + 1];
    int pid[kTreeDepth];
#pragma unroll
    for (int d = 0; d < kTreeDepth; ++d) pid[d] = uni(path_ids[(size_t)item * kTreeDepth + d]);
    const int new_id = new_ids ? uni(new_ids[item]) : -1;
    const int* __restrict__ rects = v.colrect + (size_t)root * v.rank_cap;
    unsigned rc_pre[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int k = tid + j * NT;
        rc_pre[j] = (k < v.rank_cap) ? (unsigned)rects[k] : 0u;
    }
    // ------------------------------------------------------------------ round 2: the path nodes' records (m, rectangle)
    int pm[kTreeDepth], poff[kTreeDepth];
    unsigned prc[kTreeDepth];
    int n_cols = r_root, depth = 0, parent_id = -1;
#pragma unroll
    for (int d = 0; d < kTreeDepth; ++d) {
        const int idc = min(max(pid[d], 0), tv.node_cap - 1);
        pm[d] = uni(tv.node_meta[kNodeMeta * idc]);
        prc[d] = (unsigned)uni(tv.node_meta[kNodeMeta * idc + 4]);
    }
#pragma unroll
    for (int d = 0; d < kTreeDepth; ++d) {
        const bool valid = pid[d] >= 0 && pid[d] < tv.node_cap;  // (valid ids first, -1 padded: TreeNodePool.path)
        poff[d] = valid ? n_cols : 0x7fffffff;
        if (!valid) { pm[d] = 0; prc[d] = 0x000000ffu; }  // (empty rectangle: r0 = 255 > r1 = 0)
        n_cols += pm[d];
        depth += valid ? 1 : 0;
        parent_id = valid ? pid[d] : parent_id;
    }

    // ------------------------------------------------------------------ header
    const bool expand = new_id >= 0 && new_id < tv.node_cap && !(flags & IPP_PREDICT_ONLY);
    const unsigned flags_eff = (flags | IPP_COV_ONLY | (expand ? 0u : (unsigned)IPP_PREDICT_ONLY)) & ~(unsigned)IPP_UPDATE_PREV;
    // (wave 0 evaluates the header, the other waves take it from LDS: k_step_patch.h)
    const int R = v.window_rows;
    const PrepLds<MC> pl(lds.small);
    ItemHdr h;
    if (__builtin_amdgcn_readfirstlane(wave) == 0) {
        h = make_item_header<MC, IPP_FACTOR>(v, root0, root0, slots_ok, ax, ay, az, px, py, pz, n_cols, sv_d, ls_d, flags_eff);
        const int r0 = max(0, h.yu - R), r1 = min(v.H - 1, h.yd + R);
        const int c0 = max(0, h.xl - R) & ~(VEC - 1), c1 = min(v.W - 1, min(v.W - 1, h.xr + R) | (VEC - 1));
        if (h.m > 0 && (r1 - r0 + 1 > v.ph || c1 - c0 + 1 > v.pw || n_cols > min(kPatchMaxRank, v.rank_cap))) {  // (the chain must fit the record lists)
            h.status = IPP_STATUS_BAD_FOOTPRINT; h.m = 0; h.f = 0; h.rows = 0; h.commit = 0;
        }
        if (tid == 0) {
            *pl.hs = h;
            *next_unit = 0; *done_waves = 0; *solve_flag = 0;
            lds.red[0] = 0.0; lds.red[1] = 0.0;
        }
    }
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(wave) != 0) h = *pl.hs;
    h = uniform_hdr(h);
    const int r0n = max(0, h.yu - R), r1n = min(v.H - 1, h.yd + R);
    const int c0n = max(0, h.xl - R) & ~(VEC - 1), c1n = min(v.W - 1, min(v.W - 1, h.xr + R) | (VEC - 1));
    const int hn = r1n - r0n + 1, wn = c1n - c0n + 1;
    if (tid < MC) { pl.zz[tid] = 0.0; pl.vv[tid] = 0.0; }  // (covariance only: no observation)
    const int m = h.m, f = h.f, r = h.rank;
    if (m == 0) {
        if (tid == 0) {
            v.hdr[item] = h;
            if (status_out) status_out[item] = h.status;
            reward_out[item] = 0.f;
            if (eo.t_num) {
                if (h.status != 0) eo.err[2] = h.status;
                eo.t_num[(size_t)eo.parent[item] * eo.kmax + eo.k[item]] = 0.0;
            }
        }
        return;
    }
    if (tid == 0) IPP_MARK(item, 3);
    const char* base0 = reinterpret_cast<const char*>(v.cov) - kTreePatchGuard;  // offsets of the records are relative to this
    const long long root_off = (long long)root * (long long)v.cov_slot * 4 + kTreePatchGuard;
    const long long node_off = (reinterpret_cast<const char*>(tv.node_cov) - reinterpret_cast<const char*>(v.cov)) + kTreePatchGuard;
    float* ovf = v.q + (size_t)(item + eo.item_base) * v.q_item;
    const int cap = v.pcap, pw = v.pw;

    // ------------------------------------------------------------------ the chain's columns of this thread: rectangle + byte offset
    // column k < r_root: root column k; else column k - poff[d] of path node d
    unsigned rcs[RJ];
    long long offs[RJ];
    bool con[RJ];
    unsigned long long bal[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int k = tid + j * NT;
        unsigned rc = rc_pre[j];
        long long off = root_off + (long long)k * v.pstride * 4;
#pragma unroll
        for (int d = 0; d < kTreeDepth; ++d)
            if (k >= poff[d]) {  // (poff increases along the path; invalid entries are INT_MAX)
                rc = prc[d];
                off = node_off + ((long long)pid[d] * MC + (k - poff[d])) * (long long)v.pstride * 4;
            }
        rcs[j] = rc;
        offs[j] = off;
        const int r0k = rc & 0xff, r1k = (rc >> 8) & 0xff, c0k = (rc >> 16) & 0xff, c1k = rc >> 24;
        con[j] = k < r && r0k <= h.yd && r1k >= h.yu && c0k <= h.xr && c1k >= h.xl;
        bal[j] = __ballot(con[j]);
        if (lane == 0) wcnt[j * NW + wave] = __popcll(bal[j]);
    }
    constexpr int TW = (NW > 2) ? 2 : 1;  // the wave that fills the small per-item tables (k_step_patch.h)
    const int ttid = tid - kWave * TW;
    if (ttid >= 0 && ttid < MC) {  // measurement blocks of the footprint as flat (cell, weight) tables
        const Block bb = block_of(min(ttid, m - 1), h.nx, h.rf, h.w, h.h);  // sensor_models.py:57-79
        for (int a = 0; a < 4; ++a) {
            const int aa = min(a, bb.count() - 1);
            const int ly = bb.y0 + blk_dy(aa, bb.bw), lx = bb.x0 + blk_dx(aa, bb.bw);
            lds.fb_yx[4 * ttid + a] = (h.yu + ly) | ((h.xl + lx) << 16);  // grid row | grid column << 16 (k_step_patch.h)
            lds.fb_w[4 * ttid + a] = (ttid < m && a < bb.count()) ? (float)bb.weight : 0.f;
            if (ttid < m) pl.bfi[4 * ttid + a] = bfi_pack(ly, lx, h.w);
        }
        if (ttid < m) { pl.bcnt[ttid] = bb.count(); pl.bwt[ttid] = bb.weight; }
    }
    __syncthreads();
    if (tid == 0) IPP_MARK(item, 4);
    int pos[RJ];
    int n_c = 0;
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const int c = wcnt[j * NW + w];
            before += (w < wave) ? c : 0;
            all += c;
        }
        pos[j] = n_c + before + __popcll(bal[j] & ((1ull << lane) - 1ull));
        n_c += all;
    }
    // the contributing columns change hands (k_step_patch.h): thread t takes the t-th of them, so the gather runs over n_c columns
    unsigned short* klist = lds.ridx;
    unsigned* rclist = reinterpret_cast<unsigned*>(pl.S);  // (S, L, Li: 2160 bytes in front of zz / vv)
    static_assert(4 * kPatchMaxRank <= 3 * MC * (MC + 1) * 8, "rectangle list does not fit the fp64 scratch");
#pragma unroll
    for (int j = 0; j < RJ; ++j)
        if (con[j]) { klist[pos[j]] = (unsigned short)(tid + j * NT); rclist[pos[j]] = rcs[j]; }
    __syncthreads();
    // (no barrier behind the reads: nothing below writes the two lists' areas before the barrier that ends the gather)

    // ------------------------------------------------------------------ gather HT for the contributing columns
    // (lanes <-> columns in rounds of 64; the waves of the item share a round, wave w takes the measurement blocks i = w (mod NW):
    // k_step_patch.h)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto mine = [&](int i) { return (i % NW) == wave_u; };
    auto col_off = [&](int k) {  // byte offset of chain column k from base0: root column k, or column k - poff[d] of path node d
        long long off = root_off + (long long)k * v.pstride * 4;
#pragma unroll
        for (int d = 0; d < kTreeDepth; ++d)
            if (k >= poff[d]) off = node_off + ((long long)pid[d] * MC + (k - poff[d])) * (long long)v.pstride * 4;
        return off;
    };
    auto gather_issue = [&](unsigned rc, long long off, bool on, float (&l)[MC][4]) {
        const unsigned r0k = rc & 0xff, r1k = (rc >> 8) & 0xff, c0k = (rc >> 16) & 0xff, c1k = rc >> 24;
        // patch_k[(fy - r0k) * pw + (fx - c0k)]; lanes without a column (and cells outside it) read the arena's first float
        const float* patch = reinterpret_cast<const float*>(base0 + (on ? off : kTreePatchGuard)) - ((int)r0k * pw + (int)c0k) * (on ? 1 : 0);
#pragma unroll
        for (int i = 0; i < MC; ++i) {
#pragma unroll
            for (int a = 0; a < 4; ++a) l[i][a] = 0.f;
            if (i < m && mine(i)) {  // wave-uniform
                const int cnt = uni(pl.bcnt[i]);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    if (a < cnt) {  // wave-uniform
                        const unsigned yx = (unsigned)uni(lds.fb_yx[4 * i + a]);
                        const unsigned fy = yx & 0xffffu, fx = yx >> 16;
                        const bool in = on && fy >= r0k && fy <= r1k && fx >= c0k && fx <= c1k;
                        const float val = in ? patch[(int)(fy * (unsigned)pw + fx)] : patch[on ? (int)(r0k * (unsigned)pw + c0k) : 0];
                        l[i][a] = in ? val : 0.f;
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto gather_store = [&](unsigned rc, long long off, bool on, int a_pos, const float (&l)[MC][4]) {
        if (!on) return;
        float* dst = (a_pos < cap) ? lds.rec + (size_t)a_pos * kPatchRec : ovf + (size_t)(a_pos - cap) * kPatchRec;
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            if (i < m && mine(i)) {
                const int cnt = uni(pl.bcnt[i]);
                float t = l[i][0];
                if (cnt > 1) t += l[i][1];
                if (cnt > 2) t += l[i][2] + l[i][3];
                dst[i] = -(t * lds.fb_w[4 * i]);
            }
        }
        if (wave_u != 0) return;
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (i >= m) dst[i] = 0.f;
        const int r0k = rc & 0xff, r1k = (rc >> 8) & 0xff, c0k = (rc >> 16) & 0xff, c1k = rc >> 24;
        const int shift = (r0n - r0k) * pw + (c0n - c0k);
        reinterpret_cast<float4*>(dst)[3] = make_float4(__uint_as_float((unsigned)((off + (long long)shift * 4) >> 3)),  // shifted patch: offset from base0 in 8-byte units
                                                        __int_as_float(r0k | (c0k << 16)), __int_as_float((r1k - r0k) | ((c1k - c0k) << 16)), 0.f);
    };
    {
        float l0[MC][4];
        const bool any0 = n_c > 0;  // (wave-uniform)
        const bool on0 = lane < n_c;
        const unsigned rc0 = on0 ? rclist[lane] : 0u;
        const long long off0 = col_off(on0 ? (int)klist[lane] : 0);
        if (any0) gather_issue(rc0, off0, on0, l0);
        {
            const float s3 = (float)(kSqrt3 * v.res) / h.ls;
            const int lw = v.plw;
            for (int i = tid; i < lw * lw; i += NT) {
                const int dr = div_small(i, lw), dc = i - dr * lw;
                lds.lut[i] = matern_f(dr, dc, s3, h.sv);
            }
            if (ttid >= 0 && ttid < f) { const int ky = div_small(ttid, h.w); pl.ktab[ttid] = matern_d(ky, ttid - ky * h.w, v.res, sv_d, ls_d); }
        }
        if (tid == 0) IPP_MARK(item, 5);
        if (any0) gather_store(rc0, off0, on0, lane, l0);
    }
#pragma unroll 1
    for (int t0 = kWave; t0 < n_c; t0 += kWave) {  // (wave-uniform; the lists stay in place until the barrier below)
        const int t = t0 + lane;
        const bool on = t < n_c;
        const unsigned rc = on ? rclist[t] : 0u;
        const long long off = col_off(on ? (int)klist[t] : 0);
        float l[MC][4];
        gather_issue(rc, off, on, l);
        gather_store(rc, off, on, t, l);
    }
    const int n_lds = min(n_c, cap), n_ovf = n_c - n_lds;
    if (n_ovf > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) IPP_MARK(item, 1);

    // ------------------------------------------------------------------ m x m algebra by wave 0; every other wave streams at once
    if (wave == 0) {
        const int status = solve_wave_fast<MC>(v, h, item, flags_eff, lds.small, lds.rec, 1, kPatchRec, lds.Ls, lds.ys, nullptr, status_out,
                                               nullptr, n_lds, n_ovf > 0 ? ovf : nullptr, n_ovf);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(solve_flag, status == IPP_STATUS_NOT_PD ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0 && eo.t_num && status != 0) eo.err[2] = status;
        if (lane == 0) IPP_MARK(item, 7);
    }
    const int n_fast = min(n_lds, 2 * kWave);
    unsigned mcofs[2], mlo[2], mex[2];
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const int a = p2 * kWave + lane;
        mcofs[p2] = 0u; mlo[p2] = 0x0000ffffu; mex[p2] = 0u;
        if (a < n_fast) {
            const float4 mt = *reinterpret_cast<const float4*>(lds.rec + (size_t)a * kPatchRec + 12);
            mcofs[p2] = __float_as_uint(mt.x); mlo[p2] = __float_as_uint(mt.y); mex[p2] = __float_as_uint(mt.z);
        }
    }

    // ------------------------------------------------------------------ units of the new node's patch (k_patch_units.h)
    const bool commit_u = h.commit != 0 && expand;
    const UnitGeo ug = unit_geometry(r0n, c0n, hn, wn);
    const int n_units = ug.n_units;
    TreeIo io;
    io.base0 = base0;
    io.mean_ro = v.mean + (size_t)root * v.Npad;
    io.diag_root = v.diag + (size_t)root * v.Npad;
    io.node_diag = tv.node_diag;
    io.new_cols = tv.node_cov + (size_t)max(new_id, 0) * MC * v.pstride;
    io.new_diag = tv.node_diag + (size_t)max(new_id, 0) * v.pstride;
#pragma unroll
    for (int d = 0; d < kTreeDepth; ++d) { io.prc[d] = prc[d]; io.pid[d] = pid[d]; }
    io.m = m; io.pw = pw; io.pstride = v.pstride;
    UnitArgs ua;
    ua.m = m; ua.rf1 = (h.rf == 1); ua.adaptive = (flags & IPP_ADAPTIVE) != 0; ua.commit_u = commit_u;
    ua.n_c = n_c; ua.n_fast = n_fast; ua.cap = cap; ua.ovf = ovf;
    ua.next_unit = next_unit; ua.solve_flag = solve_flag; ua.item = item;
    ua.ridx = lds.ridx + (size_t)wave * (v.rank_cap + KP);
    unsigned long long units = 0, needed = 0;
    bool dead = false;
    const UnitLds ul = {lds.rec, lds.Ls, lds.ys, lds.lut, lds.fb_yx, lds.fb_w};
    patch_units<KP>(v, ul, lds.unit_red, io, ua, ug, mcofs, mlo, mex, units, needed, dead);

    // ------------------------------------------------------------------ per-item results (last wave to arrive)
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(lds.red);
    if (lane == 0 && units) { atomicAdd(cnt, units); atomicAdd(cnt + 1, needed); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int arrived = 0;
    if (lane == 0) arrived = atomicAdd(done_waves, 1);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived == 0 && lane == 0) IPP_MARK(item, 6);
    if (arrived != NW - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    dead = __hip_atomic_load(solve_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 2;
    if (lane == 0) {
        IPP_MARK(item, 2);
        double tot = 0.0;
        for (int t = 0; t < n_units; ++t) tot += lds.unit_red[t];
        const float rew = dead ? NAN : (float)(tot / (pl.hs->cost_d + 1.0));  // rewards.py:31
        reward_out[item] = rew;
        if (eo.t_num) eo.t_num[(size_t)eo.parent[item] * eo.kmax + eo.k[item]] = (double)rew * (eo.cost[item] + 1.0);
        if (commit_u && !dead) {
            int* meta = tv.node_meta + kNodeMeta * new_id;
            meta[0] = m;
            meta[1] = pl.hs->t_lo | (pl.hs->t_hi << 16);
            meta[2] = parent_id;
            meta[3] = root;
            meta[4] = (int)rect_pack(r0n, r1n, c0n, c1n);
        }
        if (cnt[0]) {  // (per-item totals: k_step_patch.h)
            unsigned long long* ic = v.item_counts + 2 * (size_t)(item + eo.item_base);
            atomicAdd(ic, cnt[0]);      // (no return value: the wave does not wait for the round trip; nobody else adds to this line)
            atomicAdd(ic + 1, cnt[1]);
        }
    }
}

// diag of a node's state in the patch layout: the deepest node (from `node` upwards) whose rectangle holds the cell, else the root env's
__global__ void k_tree_read_diag_patch(View v, TreeView tv, int node, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= v.N) return;
    const int row = c / v.W, col = c - row * v.W;
    int cur = node;
    const int root = tv.node_meta[kNodeMeta * node + 3];
    float val = 0.f;
    bool found = false;
    for (int hops = 0; hops <= kTreeDepth && cur >= 0; ++hops) {
        const unsigned rc = (unsigned)tv.node_meta[kNodeMeta * cur + 4];
        if (rect_has(rc, row, col)) {
            val = tv.node_diag[(size_t)cur * v.pstride + (row - (int)(rc & 0xff)) * v.pw + (col - (int)((rc >> 16) & 0xff))];
            found = true;
            break;
        }
        cur = tv.node_meta[kNodeMeta * cur + 2];
    }
    out[c] = found ? val : v.diag[(size_t)root * v.Npad + c];
}

}  // namespace ipp
