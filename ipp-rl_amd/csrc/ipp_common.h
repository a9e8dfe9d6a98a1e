// Shared device-side definitions of the IPP step engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/ipp_engine.h"

#include "ipp_instrument.h"

namespace ipp {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kCountSlots = 64;    // byte-counter slots (View::counters)
#ifndef IPP_PREP_THREADS
#define IPP_PREP_THREADS 128
#endif
constexpr int kPrepThreads = IPP_PREP_THREADS;  // prologue workgroup
constexpr int kMaxTileThreads = 640;
constexpr int kBandRows = 20;      // rows of P per dense-downdate workgroup
constexpr double kSqrt3 = 1.7320508075688772;

// Per-item record written by the prologue kernel and consumed by the streaming kernels.
struct ItemHdr {
    int env, dst, rank, status;
    int xl, xr, yu, yd;
    int w, h, nx, ny;
    int rf, m, f, rows;      // rows = rank (factor) or f (dense): length of the streaming loop
    int fallback, commit, t_lo, t_hi;  // [t_lo, t_hi]: tiles that hold the column(s) this step appends
    float cost, sv, ls, nv;
    double cost_d, nv_d;
};

// The item header is the same for every lane, but it reaches the streaming kernels through vector loads (LDS or
// global), so the compiler keeps its ~30 words in VGPRs for the whole tile loop.  readfirstlane moves them to SGPRs.
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ float uni(float x) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))); }
__device__ __forceinline__ const float* uni_ptr(const float* p) {  // wave-uniform pointer -> SGPR pair
    const unsigned long long a = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return reinterpret_cast<const float*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double uni(double x) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ ItemHdr uniform_hdr(const ItemHdr& h) {
    ItemHdr o;
    o.env = uni(h.env); o.dst = uni(h.dst); o.rank = uni(h.rank); o.status = uni(h.status);
    o.xl = uni(h.xl); o.xr = uni(h.xr); o.yu = uni(h.yu); o.yd = uni(h.yd);
    o.w = uni(h.w); o.h = uni(h.h); o.nx = uni(h.nx); o.ny = uni(h.ny);
    o.rf = uni(h.rf); o.m = uni(h.m); o.f = uni(h.f); o.rows = uni(h.rows);
    o.fallback = uni(h.fallback); o.commit = uni(h.commit); o.t_lo = uni(h.t_lo); o.t_hi = uni(h.t_hi);
    o.cost = uni(h.cost); o.sv = uni(h.sv); o.ls = uni(h.ls); o.nv = uni(h.nv);
    o.cost_d = uni(h.cost_d); o.nv_d = uni(h.nv_d);
    return o;
}

// Where the columns of an item's factor state live when it is a TREE state (ipp_tree_step): the root env's slab
// followed by the column blocks of the nodes on the path from the root (each node holds the <= MC columns its own
// step appended, at stride Npad, plus their tile span).  Plain env steps read one slab and do not use this.
constexpr int kTreeDepth = 6;  // nodes on a path (episode_horizon of the tree searches is <= 5)
constexpr int kNodeMeta = 8;   // ints per node record (TreeView::node_meta)

// RECTANGLE of a stored column (View::rect_meta): a column written on rectangle tiles holds values on the grid rows
// [r0, r1] x columns [c0, c1] only (inclusive, the column bounds on whole VEC-cell groups) -- the other cells of its tile
// span are NOT written (no zeros stored) and every reader masks them.  Packed r0 | r1 << 8 | c0 << 16 | c1 << 24
// (grids of <= 256 x 256 cells); kRectFull = the whole tile span (columns written on band tiles).
constexpr unsigned kRectFull = 0xff00ff00u;
__host__ __device__ __forceinline__ unsigned rect_pack(int r0, int r1, int c0, int c1) {
    return (unsigned)r0 | ((unsigned)r1 << 8) | ((unsigned)c0 << 16) | ((unsigned)c1 << 24);
}
__host__ __device__ __forceinline__ bool rect_has(unsigned rc, int row, int col) {
    const int r0 = rc & 0xff, r1 = (rc >> 8) & 0xff, c0 = (rc >> 16) & 0xff, c1 = rc >> 24;
    return row >= r0 && row <= r1 && col >= c0 && col <= c1;
}

struct ChainCols {
    const float* root;        // root slab: columns 0 .. r_root-1
    const int* root_spans;    // tile spans of the root's columns
    const int* root_rects;    // their rectangles (View::colrect)
    int r_root, depth;
    // column block of path node j.  A node stores its columns on its own tile span only (stride nstride = win_tiles tiles
    // per column); the pointer is pre-shifted by -t_lo tiles so that it is indexed with the ABSOLUTE cell like the root's
    // rows.  Cells outside the span are never read (every consumer tests span(k) first).
    const float* node[kTreeDepth];
    int off[kTreeDepth];            // index of its first column in the chained state
    int nspan[kTreeDepth];          // its tile span (lo | hi << 16)
    unsigned nrect[kTreeDepth];     // its rectangle (rect_pack)
    size_t npad, nstride;
    __device__ __forceinline__ const float* row(int k) const {
        const float* b = root;
        int kk = k;
        size_t st = npad;
#pragma unroll
        for (int j = 0; j < kTreeDepth; ++j)
            if (j < depth && k >= off[j]) { b = node[j]; kk = k - off[j]; st = nstride; }
        return b + (size_t)kk * st;
    }
    __device__ __forceinline__ int span(int k) const {
        int s = (r_root > 0) ? root_spans[min(k, r_root - 1)] : 0;
#pragma unroll
        for (int j = 0; j < kTreeDepth; ++j)
            if (j < depth && k >= off[j]) s = nspan[j];
        return s;
    }
    __device__ __forceinline__ unsigned rect(int k) const {
        unsigned s = (r_root > 0) ? (unsigned)root_rects[min(k, r_root - 1)] : kRectFull;
#pragma unroll
        for (int j = 0; j < kTreeDepth; ++j)
            if (j < depth && k >= off[j]) s = nrect[j];
        return s;
    }
};

// Everything the kernels need, passed by value.
struct View {
    int W, H, N, Npad, T, n_tiles, vec, env_base;  // env_base: first env of a chunk when env_ids == NULL
    int mode, cap, rank_cap, max_batch;
    int window_rows, tile_cells;
    int tile_shift;  // log2(tile_cells) when it is a power of two, else -1 (the header's tile span then takes the division)
    const int* item_order;  // [n] dispatch order of the items of a launch (ipp_set_item_order), NULL: xcd_item
    int item_order_n;
    int clip_cols;  // windowed factor state: new columns are zero on the grid COLUMNS farther than window_rows from the footprint too
    int win_tiles;  // windowed factor state: most tiles [t_lo, t_hi] one step can touch (n_tiles when not windowed)
    int meas_cap, fp_cap, q_stride, q_rows;
    uint64_t q_item;  // floats of Q scratch per item: (q_rows + 2*kPipe pad rows) * q_stride
    double res, tanx, tany, rf_alt, coeff_a, coeff_b, sv0, ls0, vmax, amax, thr, kf;
    double ls_max;   // windowed factor state: largest length scale the window is good for (0: no limit)
    // state slabs
    float* mean;     // [cap][Npad]
    float* diag;     // [cap][Npad]
    float* gt;       // [2 cap][Npad] ground-truth planes: env e reads plane gt_slot[e] (e or cap + e); the OTHER one receives the next
                     // episode's field ahead of time (ipp_generate_grf_groups with gt_out == NULL) and a folded reset only flips the slot
    int* gt_slot;    // [2 cap]: [e] = the plane env e reads; [cap + e] = 1 while a field generated for e's NEXT episode waits in the alternate
                     // plane (set by the generator, taken by the flip: a flip that finds 0 would install a stale plane and poisons the env instead)
    double* prior;   // [cap][2]  (sigma^2, l)
    int* rank;       // [cap]
    int* colspan;    // factor: [cap][rank_cap]  lo_tile | hi_tile << 16 of every column of U
    int* colrect;    // factor: [cap][rank_cap]  rectangle of every column of U inside its tile span (rect_pack)
    int rect_meta;   // 1: steps on rectangle tiles write no zeros outside the rectangle; readers mask with colrect
    // PATCH layout of the windowed factor columns (k_step_patch.h): stored column k of an env is the compact patch of its
    // rectangle, ph x pw floats with the fixed row stride pw, at cov + env * cov_slot + k * pstride
    int patch;       // 1: patch layout (then rect_meta == 1 and colrect holds every column's rectangle)
    int pw, ph;      // patch width (cells, even) and height (rows)
    int pstride;     // floats per stored column: ph * pw rounded up to 16
    int plw;         // the prior table of the patch kernel is P0(|drow| < plw, |dcol| < plw)
    int pcap;        // column records per item kept in LDS (the rest in the item's global scratch block)
    int punits;      // most (64 lanes x 2 cells) units of one patch
    // split step (k_step_split.h): the items' blocks, written by the prologue kernel and read by the unit kernel
    float* blk;      // [max_batch][blk_stride], indexed by blk_pos0 + the workgroup index of the prologue launch (= dispatch position)
    int blk_stride;  // floats per block (SplitBlk::floats)
    int blk_pos0;    // first block of this launch (ipp_step_parts: the part's first position)
    // kCountSlots slots of 16 words (128 B apart): a workgroup adds its totals to slot (item % kCountSlots), word 0 =
    // streamed floats (SURVEY 8(d) count), word 8 = floats re-read for the mask.  One address for all workgroups cost
    // 5 % (one counter) / 19 % (two) of the fused step kernel: the waves' exits queued up behind same-address atomics.
    // (slot 0, words 1..7: debug phase timing)
    unsigned long long* counters;
    // patch kernels: per ITEM [max_batch][2] = (streamed floats, floats on the lanes inside the stored columns' rectangles), added by
    // the item's last wave with return-less atomics (an item index belongs to one workgroup per launch: no shared line, no wait)
    unsigned long long* item_counts;
    float* cov;      // factor: [cap][rank_cap][Npad]   dense: [cap][N][Npad]
    uint64_t cov_slot;  // floats per env slot
    // per-call scratch
    ItemHdr* hdr;    // [max_batch]
    float* linv;     // [max_batch][MC*MC]  upper-triangular L^-1 (or S^-1 on fallback), fp32 for the stream
    float* yv;       // [max_batch][MC]
    float* q;        // [max_batch][q_rows][q_stride]
    float* wc;       // dense: [max_batch][MC][Npad]
    double* partial; // [max_batch][n_tiles]
    double* dbg;     // [max_batch][2*MC*MC + 2*MC]   S, Linv, z, y in fp64 (tests)
    int dbg_capture; // the wave-level m x m algebra (solve_wave_fast: fused / patch / tree kernels) fills dbg only when set (ipp_debug_capture)
    double* grf_h;   // [H][W] circular-convolution kernel of the GRF
    double2* grf_cs; // [W] (cos, sin)(2 pi j / W)          (k_grf_dft.h)
    double* grf_g;   // [H/2+1][W] column-convolution kernels  (k_grf_dft.h)
    double* grf_hp;  // [NP][NP] Hartley matrix cos + sin of 2 pi j k / n, zero padded to NP = 16 ceil(n / 16)  (k_grf_hartley.h)
    double* grf_amp; // [NP][NP] spectral amplitude, zero padded  (k_grf_hartley.h)
    float* grf_raw;  // [max_batch][Npad] un-normalised field (ipp_reset)
    float* grf_raw2; // [max_batch][Npad] un-normalised field (ipp_generate_grf, may run on a side stream)
};

// The plane that holds env's ground truth / the one that is staged for its next episode.
__device__ __forceinline__ float* gt_plane(const View& v, int env) { return v.gt + (size_t)v.gt_slot[env] * v.Npad; }
__device__ __forceinline__ int gt_alt_slot(const View& v, int env) { const int s = v.gt_slot[env]; return s >= v.cap ? s - v.cap : s + v.cap; }

// Episode reset folded into a step launch (ipp_step_autoreset): item i resets its env after its step when
// src[i] >= 0, taking ground truth gt[src[i]] ([N] floats).  Mission.init_action by value.
struct AutoReset {
    const int* src;     // [n] or NULL (no resets in this launch)
    const float* gt;    // [..][N], or NULL: the new ground truth already sits in the env's alternate plane -- the reset flips gt_slot
    const double* prior; // [..][2] prior (sigma^2, l) of the new episode that takes ground truth k, or NULL: the config's (ipp_set_reset_prior)
    double* prev;       // [capacity][3] indexed by env id, or NULL
    double init[3];
};

// One wave resets env `env` (mapping/mappings.py:235-240,259-261: mean 0.5, P = prior; factor state: rank 0) and
// installs ground truth field k.  The caller guarantees that every earlier store / atomic to the env's planes
// has completed.
__device__ __forceinline__ void wave_reset_env(const View& v, const AutoReset& ar, int env, int k, int lane) {
    // (prior of the new episode like k_reset_small: shuffle_prior_cov scales, mappings.py:238-240; a length scale the column
    // window was not sized for poisons the env instead of losing accuracy silently)
    double sv_d = ar.prior ? ar.prior[2 * k + 0] : v.sv0, ls_d = ar.prior ? ar.prior[2 * k + 1] : v.ls0;
    if (v.ls_max > 0.0 && ls_d > v.ls_max * (1.0 + 1e-12)) sv_d = ls_d = NAN;
    float* mean = v.mean + (size_t)env * v.Npad;
    float* diag = v.diag + (size_t)env * v.Npad;
    if (!ar.gt && v.gt_slot[v.cap + env] == 0) sv_d = ls_d = NAN;  // nothing was staged for this env: no flip to a stale plane without a trace
    const float sv = (float)sv_d;
    const float m0 = isnan(sv_d) ? NAN : 0.5f;
    if (!ar.gt) {
        // the ground truth of the new episode was generated into the env's alternate plane: no 2 x 4 N bytes of copy (at 100x100 and
        // 2048 resets per step the copies were 164 MB of a step's traffic) -- mean and variance planes, then the flip
        const int alt = gt_alt_slot(v, env);
        float4* mean4 = reinterpret_cast<float4*>(mean);
        float4* diag4 = reinterpret_cast<float4*>(diag);
        const int n4 = v.N / 4, np4 = v.Npad / 4;
        if ((v.N & 3) == 0) {
            for (int c = lane; c < np4; c += kWave) {
                const bool valid = c < n4;
                mean4[c] = valid ? make_float4(m0, m0, m0, m0) : make_float4(0.f, 0.f, 0.f, 0.f);
                diag4[c] = valid ? make_float4(sv, sv, sv, sv) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            for (int c = lane; c < v.Npad; c += kWave) { mean[c] = c < v.N ? m0 : 0.f; diag[c] = c < v.N ? sv : 0.f; }
        }
        if (lane == 0) {
            v.gt_slot[env] = alt;
            v.gt_slot[v.cap + env] = 0;
            v.rank[env] = 0;
            v.prior[2 * env + 0] = sv_d;
            v.prior[2 * env + 1] = ls_d;
        }
        if (ar.prev && lane < 3) ar.prev[3 * env + lane] = ar.init[lane];
        return;
    }
    float* gt = gt_plane(v, env);
    const float* src = ar.gt + (size_t)k * v.N;
    if ((v.N & 3) == 0 && (reinterpret_cast<unsigned long long>(src) & 15ull) == 0ull) {
        // four cells per lane and instruction, the ground-truth loads of a pass all in flight before the first store: written one
        // cell per lane with the load inside the loop, the wave waited for ~40 dependent round trips (~80 us at 50x50) with
        // its workgroup's slot held -- resets folded into the step launch cost more than their own kernel
        const float4* src4 = reinterpret_cast<const float4*>(src);
        float4* mean4 = reinterpret_cast<float4*>(mean);
        float4* diag4 = reinterpret_cast<float4*>(diag);
        float4* gt4 = reinterpret_cast<float4*>(gt);
        const int n4 = v.N / 4, np4 = v.Npad / 4;
        constexpr int kU = 8;
        for (int c0 = lane; c0 < np4; c0 += kU * kWave) {
            float4 g[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int c = c0 + u * kWave;
                g[u] = (c < n4) ? src4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int c = c0 + u * kWave;
                if (c < np4) {
                    const bool valid = c < n4;
                    mean4[c] = valid ? make_float4(m0, m0, m0, m0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    diag4[c] = valid ? make_float4(sv, sv, sv, sv) : make_float4(0.f, 0.f, 0.f, 0.f);
                    gt4[c] = g[u];
                }
            }
        }
    } else {
        for (int c = lane; c < v.Npad; c += kWave) {
            const bool valid = c < v.N;
            mean[c] = valid ? m0 : 0.f;
            diag[c] = valid ? sv : 0.f;
            gt[c] = valid ? src[c] : 0.f;
        }
    }
    if (lane == 0) {
        v.rank[env] = 0;
        v.prior[2 * env + 0] = sv_d;
        v.prior[2 * env + 1] = ls_d;
    }
    if (ar.prev && lane < 3) ar.prev[3 * env + lane] = ar.init[lane];
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    return x;
}
__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    return x;
}
__device__ __forceinline__ float wave_min(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fminf(x, __shfl_xor(x, off, kWave));
    return x;
}
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, kWave));
    return x;
}

__device__ __forceinline__ double bcast_lane(double x, int src) {  // src: wave-uniform constant
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}

// ---- DPP row broadcasts: coefficients that are the same for every cell (a row of -HT, a row of L^-1, y) live in the LANES of one
// register (value j in lane j of every row of 16 lanes) and reach the FMAs through `row_newbcast`, instead of broadcast LDS reads
// (or scalar loads) in front of every use.
// acc += q[lane J of this lane's row of 16] * u: the row's -HT values live in the LANES of one register (lane l holds value
// l & 15, read from the LDS record with one ds_read_b32 per stored row while the row requests are in flight) and reach the FMAs
// through the DPP row broadcast -- no LDS read and no wait inside the FMA chain (three broadcast ds_read_b128 per row with a
// wait in front of their first use cost ~4 us per unit: the row loop was a chain of exposed LDS latencies)
template <int J>
__device__ __forceinline__ void fmac_bc(float& acc, float q, float u) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q), "v"(u), "n"(J));
}
template <int VEC, int MC>
__device__ __forceinline__ void fmac_row(float (&acc)[VEC][MC], float q, const float (&u)[VEC]) {
    static_assert(MC == 9, "unrolled for MC = 9");
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        fmac_bc<0>(acc[c][0], q, u[c]); fmac_bc<1>(acc[c][1], q, u[c]); fmac_bc<2>(acc[c][2], q, u[c]);
        fmac_bc<3>(acc[c][3], q, u[c]); fmac_bc<4>(acc[c][4], q, u[c]); fmac_bc<5>(acc[c][5], q, u[c]);
        fmac_bc<6>(acc[c][6], q, u[c]); fmac_bc<7>(acc[c][7], q, u[c]); fmac_bc<8>(acc[c][8], q, u[c]);
    }
}

// Wc[.][J] = sum_{b <= J} (Wc L)[.][b] * Linv[b][J]: row b of L^-1 sits in the lanes of lrow[b] (value J in lane J of every row of 16)
template <int J, int VEC, int MC>
__device__ __forceinline__ void linv_col(float (&acc)[VEC][MC], const float (&lrow)[MC]) {
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        float t = 0.f;
#pragma unroll
        for (int b = 0; b <= J; ++b) fmac_bc<J>(t, lrow[b], acc[c][b]);
        acc[c][J] = t;
    }
}
template <int MC>
__device__ __forceinline__ float dot_lanes(const float (&a)[MC], float yreg) {  // sum_j a[j] * y[j], y[j] in lane j of every row of 16
    static_assert(MC == 9, "unrolled for MC = 9");
    float d = 0.f;
    fmac_bc<0>(d, yreg, a[0]); fmac_bc<1>(d, yreg, a[1]); fmac_bc<2>(d, yreg, a[2]); fmac_bc<3>(d, yreg, a[3]); fmac_bc<4>(d, yreg, a[4]);
    fmac_bc<5>(d, yreg, a[5]); fmac_bc<6>(d, yreg, a[6]); fmac_bc<7>(d, yreg, a[7]); fmac_bc<8>(d, yreg, a[8]);
    return d;
}
// Sum over the wave without LDS permutes (six dependent ds_bpermute round trips per unit): quads, half rows and rows through
// DPP, the four row sums through v_readlane.  Every lane returns the total.
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_dpp(double x) {
    x += dpp_mov_f64<0xB1>(x);   // quad_perm [1, 0, 3, 2]
    x += dpp_mov_f64<0x4E>(x);   // quad_perm [2, 3, 0, 1]
    x += dpp_mov_f64<0x141>(x);  // row_half_mirror
    x += dpp_mov_f64<0x140>(x);  // row_mirror
    return (bcast_lane(x, 0) + bcast_lane(x, 16)) + (bcast_lane(x, 32) + bcast_lane(x, 48));
}

// compile-time loop (template arguments from loop counters)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Matern nu=3/2 prior between two cells (mapping/mappings.py:242-258 == analytic kernel).
__device__ __forceinline__ double matern_d(int dr, int dc, double res, double sv, double ls) {
    const double d = res * sqrt((double)(dr * dr + dc * dc));
    const double t = kSqrt3 * d / ls;
    return sv * (1.0 + t) * exp(-t);
}
__device__ __forceinline__ float matern_f(int dr, int dc, float res_s3_over_ls, float sv) {
    const float t = res_s3_over_ls * sqrtf((float)(dr * dr + dc * dc));
    return sv * (1.0f + t) * __expf(-t);
}

// Measurement block i of the footprint (sensors/models/sensor_models.py:57-79).
struct Block {
    int x0, y0, bw, bh;
    double weight;
    __device__ __forceinline__ int count() const { return bw * bh; }
};
// floor(n / d) for 0 <= n < 4096, 0 < d <= 64 without the ~20-instruction integer division: (n + 0.5) / d is at least
// 0.5 / d away from an integer, far more than the rounding of the reciprocal
__device__ __forceinline__ int div_small(int n, int d) { return (int)(((float)n + 0.5f) * __builtin_amdgcn_rcpf((float)d)); }
// cell a of a measurement block of width bw (1 or 2 cells: the resolution factor is 1 or 2): row and column inside the block
__device__ __forceinline__ int blk_dy(int a, int bw) { return bw == 2 ? a >> 1 : a; }
__device__ __forceinline__ int blk_dx(int a, int bw) { return bw == 2 ? a & 1 : 0; }
__device__ __forceinline__ Block block_of(int i, int nx, int rf, int w, int h) {
    Block b;
    const int by = div_small(i, nx), bx = i - by * nx;
    const int x1 = min(bx * rf + rf, w), y1 = min(by * rf + rf, h);
    b.x0 = min(bx * rf, x1);
    b.y0 = min(by * rf, y1);
    b.bw = x1 - b.x0;
    b.bh = y1 - b.y0;
    // (sensor_models.py:62-79: 1 / rf for clipped blocks, 1 / rf^2 for whole ones; rf is 1 or 2, both quotients are exact)
    const double inv = (rf == 2) ? 0.5 : (rf == 1 ? 1.0 : 1.0 / rf);
    b.weight = (b.bw * b.bh < rf * rf) ? inv : inv * inv;
    return b;
}

// XCD-aware block -> item map.  Blocks b, b+8, b+16.. run on one XCD (observed b % 8) and every XCD works through
// its own list independently, so item = b gives XCD x the items with index = x mod 8.  Any workload whose cost has
// a period sharing a factor with 8 then loads the XCDs unevenly (staggered 40-step episodes, rank ~ phase: the
// heaviest XCD started its last workgroup 60 us after the lightest one, 15 % of the step kernel).
// Here the item list is cut into 16 contiguous ranges and XCD x works through range x, then range 15 - x: a range
// covers many cycles of any short period, and a list sorted by cost (or any trend along the list) is balanced by
// the mirrored second range, which one contiguous range per XCD would not be (+55 % on the last XCD for a cost
// rising 0.3 -> 1.3 along the list).  Left over: cost periods of more than n/16 items (10 % at n = 4096, period 500).
__device__ __forceinline__ int xcd_item(int b, int n_items) {
    const int len = n_items >> 4, full = len << 4;  // 16 ranges of len items, then the ragged end
    if (b >= full) return b;
    const int xcd = b & 7, slot = b >> 3;            // slot in [0, 2 len)
    const bool second = slot >= len;
    return (second ? 15 - xcd : xcd) * len + (second ? slot - len : slot);
}
// (item, part) for kernels with several blocks per item: the parts of one item sit on consecutive slots of the same
// XCD and share its L2 (Q, Wc, header).  Grid = grid_for(n_items, parts).
__device__ __forceinline__ bool decode_block(int b, int n_items, int parts, int& item, int& part) {
    const int xcd = b & 7, slot = b >> 3;
    const int j = slot / parts;
    const int vb = j * 8 + xcd;  // the block index a one-block-per-item launch would have on this XCD
    part = slot - j * parts;
    if (vb >= n_items) { item = 0; return false; }
    item = xcd_item(vb, n_items);
    return true;
}
// Item of workgroup b: the caller's dispatch order when one is set for this launch size (heaviest items first: workgroups are
// dealt to the XCDs round-robin, so every XCD works through a descending list and the launch ends on the short items), else
// the XCD-balanced default.
__device__ __forceinline__ int launch_item(const View& v, int b, int n_items) {
    return (v.item_order && v.item_order_n == n_items) ? v.item_order[b] : xcd_item(b, n_items);
}
inline int grid_for(int n_items, int parts) { return ((n_items + 7) / 8) * 8 * parts; }

}  // namespace ipp
