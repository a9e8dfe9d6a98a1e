// Factor-state gain kernel: ONE workgroup per step item, wave-granular cell tiles.
//
//   Wc = P0[:,F] H_F^T L^-1  -  U (U[F,:]^T H_F^T L^-1)          (mapping/mappings.py:188, P = P0 - U U^T)
//
// The workgroup keeps L^-1, y, the tile spans of the stored columns and the prior table P0(|drow|, |dcol|) in LDS;
// every wave then walks over (64 x VEC)-cell tiles handed out by an LDS counter.  Per tile a wave
//   * compacts (ballot / popcount, in increasing k) the columns of U that are stored on the tile -- all of them
//     when window_rows == 0, else those whose footprint lies within window_rows grid rows (ipp_config),
//   * evaluates the prior term from the table (4 x VEC lookups in flight per measurement block),
//   * streams the stored rows in groups of KP (1 KiB per wave instruction, non-temporal) against Q rows that
//     arrive through the SCALAR cache: Q row k is wave-uniform, so it is read from the item's global scratch
//     block with s_load and used as an SGPR operand of the FMAs -- no VGPRs and no LDS for Q, which is what
//     leaves room for KP = 12 rows (12 KiB) in flight per wave at 4 waves / SIMD,
//   * runs the fused epilogue: masked trace reduction (reward), diag -= |Wc_i|^2, mean += Wc y, append the m new
//     rows of U on the tile (planning/common/rewards.py:8-31, mapping/mappings.py:190-197).
// No workgroup barrier inside the tile loop; tiles outside the span of the appended columns are skipped.
// HBM-bound: 4*MC FMAs per streamed float4; the prologue is paid once per item, not once per tile.
#pragma once
#include <type_traits>
#include "ipp_common.h"
#include "k_gain.h"

#ifndef IPP_GF_PIPE
#define IPP_GF_PIPE 12  // rows of U requested per group, stand-alone gain kernel
#endif
#ifndef IPP_SF_PIPE
#define IPP_SF_PIPE 10  // same, fused step kernel (A/B on MI355X: 6..12 within 3 %, 10 the most consistent)
#endif
// Rows requested per group by VEC: a row of a 64 * VEC-cell tile is 256 VEC bytes per wave, so the VEC = 2 kernels keep
// more of them in flight for the same bytes (A/B at 4096 envs of 50x50, VEC = 2: 10 / 16 / 20 rows 13.65 / 13.85 / 13.85 M)
#ifndef IPP_PIPE2
#define IPP_PIPE2 16
#endif
template <int MC, int VEC> constexpr int sf_pipe() { return (MC == 9 && VEC <= 2) ? IPP_PIPE2 : IPP_SF_PIPE; }
template <int MC, int VEC> constexpr int gf_pipe() { return (MC == 9 && VEC <= 2) ? IPP_PIPE2 : IPP_GF_PIPE; }
#ifndef IPP_GF_MINWAVES
#define IPP_GF_MINWAVES 4
#endif

namespace ipp {

// LDS layout shared by k_gain_factor and k_step_factor.
//   work: prior table (lut_floats) -- in the fused kernel first the prologue's HT staging rows, which are dead
//         once Q has been written to the item's global scratch block
//   small: the fused prologue's fp64 scratch (0 floats for the stand-alone gain kernel)
template <int MC>
struct GainLds {
    static constexpr int QS = (MC + 3) & ~3;
    static constexpr int LQ = (MC * MC + MC + 3) & ~3;
    float* Ls; float* ys; float* work; float* lut; unsigned char* small; double* red; int* next_tile; int* done_waves;
    int* solve_flag; int* span_s; unsigned* rect_s; int* fb_yx; float* fb_w; float* stage; unsigned short* ridx_all; unsigned char* mask4;
    double* tile_red;  // [win_tiles] masked trace reduction of every touched tile (summed in tile order at the end)
    const float** rowp;  // [chain_rows] tree steps: pointer to every column of the chained state (ChainCols::row, evaluated once)
    // rectangle of stored column k (rect_pack) -> the bytes of the per-lane test: a cell group at (row, col) lies inside
    // iff, as pairs of 16-bit lanes, min(pos - lo, ext) == pos - lo with pos = row | col << 16, lo = first row | first
    // column << 16, ext = (rows - 1) | (columns - 1) << 16 (rect_lo / rect_ext spread the bytes: one v_perm each) -- two
    // packed instructions and one compare per stored row and lane instead of four unpacked range tests
    __device__ __forceinline__ void stage_rect(int k, unsigned rc) const {
        const unsigned r0 = rc & 0xff, r1 = (rc >> 8) & 0xff, c0 = (rc >> 16) & 0xff, c1 = rc >> 24;
        rect_s[k] = (rc == kRectFull) ? 0xffff0000u : (r0 | (c0 << 8) | ((r1 - r0) << 16) | ((c1 - c0) << 24));
    }
    static __device__ __forceinline__ unsigned rect_lo(unsigned w) { return __builtin_amdgcn_perm(0u, w, 0x0c010c00u); }   // bytes [b0, 0, b1, 0]
    static __device__ __forceinline__ unsigned rect_ext(unsigned w) { return __builtin_amdgcn_perm(0u, w, 0x0c030c02u); }  // bytes [b2, 0, b3, 0]
    // vec: cells per lane of the kernel's tiles (the per-wave mean / diag parking area holds 2 * 64 * vec floats)
    __host__ __device__ static size_t bytes(int rank_cap, int work_floats, int lut_floats, int small_floats, int waves,
                                            int n_tiles, int mask_bytes = 0, int chain_rows = 0, int vec = 4) {
        size_t b = (size_t)(LQ + ((work_floats + 3) & ~3) + ((lut_floats + 3) & ~3) + ((small_floats + 3) & ~3)) * 4 + 16 * 8;
        b += (size_t)((rank_cap + 3) & ~3) * 8 + (size_t)8 * MC * 4 + (mask_bytes ? 0 : (size_t)waves * kWave * 2 * vec * 4) + (size_t)waves * (rank_cap + 8) * 2;
        b = ((b + 15) & ~(size_t)15) + (size_t)mask_bytes;
        b = ((b + 15) & ~(size_t)15) + (size_t)n_tiles * 8 + (size_t)chain_rows * 8;
        return (b + 15) & ~(size_t)15;
    }
    // work: HT staging rows of the fused prologue (0 floats for the stand-alone kernel); lut: prior table;
    // small: the fused prologue's fp64 scratch (0 floats for the stand-alone kernel)
    // mask_bytes > 0 (fused kernel): no mean / diag staging area, the env's mask bytes instead
    __device__ __forceinline__ GainLds() {}
    __device__ __forceinline__ GainLds(unsigned char* base, int rank_cap, int work_floats, int lut_floats, int small_floats, int waves,
                                       int n_tiles, int mask_bytes = 0, int vec = 4) {
        Ls = reinterpret_cast<float*>(base);
        ys = Ls + MC * MC;
        work = Ls + LQ;
        lut = work + ((work_floats + 3) & ~3);
        small = reinterpret_cast<unsigned char*>(lut + ((lut_floats + 3) & ~3));
        red = reinterpret_cast<double*>(small + (size_t)((small_floats + 3) & ~3) * 4);
        next_tile = reinterpret_cast<int*>(red + 15);  // red[8..15] are unused by the reduction (<= 8 waves)
        done_waves = next_tile + 1;
        solve_flag = reinterpret_cast<int*>(red + 14);  // fused kernel: 0 = L^-1 / y pending, 1 = ready, 2 = S not PD
        span_s = reinterpret_cast<int*>(red + 16);
        // [rank_cap] rectangles of the stored columns (View::rect_meta) as four bytes: first row, first column, rows - 1,
        // columns - 1 (stage_rect).  (One word per column: with two, the fused kernel's LDS passed 40 KiB and a CU held
        // three workgroups instead of four.)
        rect_s = reinterpret_cast<unsigned*>(span_s + ((rank_cap + 3) & ~3));
        fb_yx = span_s + 2 * ((rank_cap + 3) & ~3);        // [MC][4] footprint cell (row << 16 | col) of block b
        fb_w = reinterpret_cast<float*>(fb_yx + 4 * MC);   // [MC][4] weight of that cell (0 for padding)
        // [waves][2][64 lanes][vec]: mean / diag of the wave's current tile, parked here across the stream loop
        // (typed pointer arithmetic only: an integer round trip would turn the LDS pointer into a flat one)
        stage = fb_w + 4 * MC;
        ridx_all = reinterpret_cast<unsigned short*>(stage + (mask_bytes ? 0 : (size_t)waves * kWave * 2 * vec));
        // adaptive-mask bits of the whole env, one byte per VEC cells (fused kernel), behind the per-wave index lists
        mask4 = reinterpret_cast<unsigned char*>(ridx_all + (((size_t)waves * (rank_cap + 8) + 7) & ~(size_t)7));
        tile_red = reinterpret_cast<double*>(mask4 + (((size_t)mask_bytes + 15) & ~(size_t)15));
        rowp = reinterpret_cast<const float**>(tile_red + n_tiles);
    }
};

// Measurement blocks of the footprint (sensors/models/sensor_models.py:57-79) as flat (cell, weight) tables.
template <int MC>
__device__ __forceinline__ void fill_block_tables(const ItemHdr& h, int* fb_yx, float* fb_w) {
    const int tid = threadIdx.x;
    if (tid < MC) {
        const int m = h.m;
        const Block bb = block_of(min(tid, m - 1), h.nx, h.rf, h.w, h.h);
        for (int a = 0; a < 4; ++a) {
            const int aa = min(a, bb.count() - 1);
            fb_yx[4 * tid + a] = ((h.yu + bb.y0 + blk_dy(aa, bb.bw)) << 16) | (h.xl + bb.x0 + blk_dx(aa, bb.bw));
            fb_w[4 * tid + a] = (tid < m && a < bb.count()) ? (float)bb.weight : 0.f;
        }
    }
}

// Stored columns that cannot contribute to this step are taken out of the item's span table (an empty span: lo > hi):
// a column whose row of H U^T (rows: [k][stride], m values each) is exactly zero has an exactly zero Q row, so streaming
// it would add zeros.  With View::clip_cols that is every column whose own column range misses the footprint.
__device__ __forceinline__ void mark_inactive_columns(int* span_s, const float* rows, int stride, int r, int m, int tid, int T) {
    for (int k = tid; k < r; k += T) {
        bool any = false;
        for (int i = 0; i < m; ++i) any |= rows[(size_t)k * stride + i] != 0.f;
        if (!any) span_s[k] = 0xffff;  // lo = 0xffff, hi = 0: covers no tile
    }
}

// Pointer (indexed with the absolute cell) to the diagonal of the state (root env `root_diag` + path nodes) on `tile`.
struct DiagChain {
    const float* root_diag;
    const float* node[kTreeDepth];  // pre-shifted by -t_lo tiles like ChainCols::node
    int nspan[kTreeDepth];
    unsigned nrect[kTreeDepth];  // View::rect_meta: a node holds its diagonal on the rectangle of its step only
    int depth;
    __device__ __forceinline__ const float* source(int tile) const {
        const float* p = root_diag;
#pragma unroll
        for (int j = 0; j < kTreeDepth; ++j)
            if (j < depth && tile >= (nspan[j] & 0xffff) && tile <= (nspan[j] >> 16)) p = node[j];  // deeper nodes override
        return p;
    }
    // ... of the cell group at (row, col) on `tile` when the nodes' rectangles count (rectangle tiles with View::rect_meta)
    __device__ __forceinline__ const float* source(int tile, int row, int col) const {
        const float* p = root_diag;
#pragma unroll
        for (int j = 0; j < kTreeDepth; ++j)
            if (j < depth && tile >= (nspan[j] & 0xffff) && tile <= (nspan[j] >> 16) && rect_has(nrect[j], row, col)) p = node[j];
        return p;
    }
};

// Predicated row load of the stream: VEC floats at byte offset `off` of a wave-uniform row, or zeros where the lane's
// predicate is false -- as a BUFFER load whose offset is pushed out of range for the masked lanes (the hardware returns 0),
// instead of a global load under a per-row exec mask (hipcc wraps each of those in s_and_saveexec / s_cbranch_execz / s_or:
// ~10 scalar instructions and two branches per stored row).  `row` must be wave-uniform (SGPRs).
template <int VEC>
__device__ __forceinline__ auto row_load_masked(const float* row, unsigned row_bytes, unsigned off, bool ok) {
    typedef float rowv __attribute__((ext_vector_type(VEC)));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, (int)row_bytes, 0x00020000);
    const unsigned voff = ok ? off : 0xffffffffu;
    if constexpr (VEC == 2) return __builtin_bit_cast(rowv, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, 0, 2));  // (aux 2: nt)
    else if constexpr (VEC == 4) return __builtin_bit_cast(rowv, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 2));
    else return __builtin_bit_cast(rowv, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 2));
}

// Tile loop + per-item results; expects Ls / ys, the block tables, the prior table P0(|drow| < lut_rows, |dcol|)
// (row distances beyond it fall back to sqrt / exp), span_s[0..r)
// and the two counters (zeroed) in LDS, visible to the whole workgroup.
// qrows: the item's Q rows [k][QS] in global scratch, followed by >= 8 zero rows.  The wave-uniform row reads are
// emitted as scalar loads only if the compiler can prove the rows read-only (a `const float* __restrict__` kernel
// argument, stand-alone kernel) or is told so (QCONST, fused kernels).
// PRE (fused kernel): qrows holds the rows of -HT (HT = H_F U[F,:]^T, m values per column) instead of Q = -HT L^-1,
// the accumulators hold Wc L = P[:,F] H_F^T per cell, and the tile epilogue applies L^-1 (upper triangular, 45 FMAs
// per cell) once lds.solve_flag says wave 0 has finished the m x m algebra.  The stream therefore starts right
// after the gather instead of after S / Cholesky / L^-1 / Q.
// LMASK (fused kernel): the adaptive mask of the touched tiles [t_lo, t_hi] was put into lds.mask4 by phase A (one byte
// per VEC cells, indexed from tile t_lo; View::win_tiles bounds the tile count) and mean / diag are
// updated with no-return float atomics (one add per cell, bit-identical to load + add + store): the tile loop has no
// mean / diag loads, whose latency sat in front of every tile's stream.
// CHAIN (ipp_tree_step): the streamed columns come from a chained tree state (cc), the m new columns go to the new
// node's block (stride cc->nstride, pointer pre-shifted to absolute cells), diag_rw is the new node's diagonal on its
// span (pre-shifted too; LMASK: initialised by phase A as a copy of the parent state's; else written here from the
// parent state's diagonal, read through dch), and instead of the env's rank / spans the node's (m, span) record is
// written; nothing of the root env slot is modified.
// QCONST: qrows is read through the constant address space (scalar loads whatever the compiler can prove about
// aliasing); the caller guarantees the rows are complete and visible before the call and unchanged during it.
// RESET (ipp_step_autoreset): an item with ar->src[item] >= 0 resets its env once its step is complete: every wave
// waits for its own stores / atomics before it counts itself done, the last wave then rewrites the env's planes.
// NW > 0: that many waves of the workgroup run the tile loop of this item (a persistent variant of rounds 2-4, deleted: the consumer waves), else all.
template <int MC, int VEC, int KP, bool PRE, bool LMASK, bool CHAIN = false, bool QCONST = false, bool RESET = false, int NW = 0,
          bool RECT = false>
__device__ __forceinline__ void gain_tiles(const View& v, const ItemHdr& h, const int item, unsigned flags, int lut_rows,
                                           const GainLds<MC>& lds, const float* __restrict__ qrows,
                                           float* __restrict__ reward_out, const ChainCols* cc = nullptr,
                                           float* new_cols = nullptr, float* diag_rw = nullptr, int* node_meta = nullptr,
                                           const AutoReset* ar = nullptr, const DiagChain* dch = nullptr) {
    constexpr int kWaveTile = VEC * kWave;  // cells per wave tile
    constexpr int QS = (MC + 3) & ~3;
    const float* Ls = lds.Ls; const float* ys = lds.ys; const float* lut = lds.lut; double* red = lds.red;
    int* next_tile = lds.next_tile; int* done_waves = lds.done_waves; const int* span_s = lds.span_s;
    const int* fb_yx = lds.fb_yx; const float* fb_w = lds.fb_w;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave, nw = NW > 0 ? NW : T / kWave;
    const int m = h.m, r = h.rank;
    const float s3 = (float)(kSqrt3 * v.res) / h.ls;
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    cfloat_p qrows_c = (cfloat_p)(const void*)qrows;
    if (QCONST) asm volatile("" : "+s"(qrows_c));  // defined here: no load through it can move above the caller's barrier

    const float* cov_src = v.cov + (size_t)h.env * v.cov_slot;
    float* cov_dst = v.cov + (size_t)h.dst * v.cov_slot;
    unsigned short* ridx = lds.ridx_all + (size_t)wave * (v.rank_cap + 8);
    float* stage_w = lds.stage + (size_t)wave * kWave * 2 * VEC;  // (the callers carve the area with vec = VEC)
    const bool adaptive = (flags & IPP_ADAPTIVE) != 0;
    const size_t npad = (size_t)v.Npad;
    unsigned long long units = 0, extra = 0;
    bool solved = !PRE, dead = false;
#if IPP_PHASE_TIMING
    const unsigned long long loop0_ = wall_clock64();
#endif

    // Two-dimensional windows (View::clip_cols).  The window of a step is a band of whole grid rows, but a new column
    // decays with the distance to the footprint in BOTH directions (the criterion of window_rows is isotropic): the cells of
    // the band whose grid column is farther than window_rows from the footprint's columns get a ZERO in the new columns
    // (stored, so that every later reader sees zeros there) and their lanes request nothing.  The point is not the
    // bytes (halving them alone did not change the step time: the stream is bound by latency per (tile, stored row)
    // pair): a stored column whose own column range misses this step's FOOTPRINT has an exactly zero row of H U^T,
    // hence an exactly zero Q row, and is dropped from the stream altogether by the callers (mark_inactive_columns).
    const int clip_lo = v.clip_cols ? max(0, h.xl - v.window_rows) : 0;
    const int clip_hi = v.clip_cols ? min(v.W - 1, h.xr + v.window_rows) : v.W - 1;
    // RECTANGLE TILES (env steps on clipped windows): the work units are not the 64 VEC-cell tiles of the row band but
    // groups of 64 VEC cells of the rectangle rows [yu - R, yd + R] x columns [rect_lo, rect_hi] (the column range widened
    // to whole VEC-cell groups), row by row: a lane's VEC cells are still adjacent and aligned, a wave's request covers
    // ~5 grid rows of ~28 cells.  Half as many (tile, stored row) pairs at 50x50, a quarter at 100x100 -- that count, not
    // the bytes, is what the stream costs.  The cells of the band outside the rectangle only get the zeros of the new
    // columns, from a store-only pass over the band tiles (zero_fill below).  Storage and spans stay in band tiles.
    // Used where it wins (ipp_engine.hip): rectangles at most 0.4 grid rows wide (100x100: 32768 envs 18.6 -> 21.0 M env-steps/s), and steps
    // that store nothing (predict-only calls at 50x50: 21.6 -> 23.2 M).  Committed steps at 50x50 lose what the halved
    // pairs save to the store-only pass and the per-lane addressing (headline +-0, 32768 envs 25.1 -> 23.4 M): band tiles.
    // A compile-time variant (RECT; the host picks the kernel): with both tilings in one kernel the band path lost 6 %.
    static_assert(!RECT || !LMASK, "rectangle tiles need the per-tile mask");
    constexpr bool rect = RECT;  // (the host guarantees View::clip_cols and W % VEC == 0)
    const int rect_lo = clip_lo & ~(VEC - 1), rect_hi = min(v.W - 1, clip_hi | (VEC - 1));
    const int rect_row0 = max(0, h.yu - v.window_rows), rect_row1 = min(v.H - 1, h.yd + v.window_rows);
    const int rect_gpr = (rect_hi - rect_lo + 1) / VEC;                       // groups of VEC cells per rectangle row
    const int rect_groups = (rect_row1 - rect_row0 + 1) * rect_gpr;
    const int n_rect_tiles = rect ? (rect_groups + kWave - 1) / kWave : 0;
    // View::rect_meta: the appended columns are written on the rectangle only (no zeros stored outside it: their rectangle is
    // recorded in View::colrect / the node record) and every stored column is read under its rectangle; a tree node's diagonal
    // likewise lives on its rectangle only (DiagChain::source per cell group).  There is no store-only pass then.
    // (Compiled into the rectangle-tile variants only: the host never runs a band-tile variant on columns that carry a true
    // rectangle -- ipp_engine.hip, rect_meta rule -- and the extra live values cost the band variants a wave of occupancy.)
    const bool rm = RECT && v.rect_meta != 0;
    const unsigned* rect_s = lds.rect_s;
    const int n_band_tiles = (rect && rm) ? 0 : h.t_hi - h.t_lo + 1;
    constexpr int kTileShift = (kWaveTile == 64) ? 6 : (kWaveTile == 128) ? 7 : 8;
    static_assert(kWaveTile == (1 << kTileShift), "tile size must be a power of two");

    // tiles are handed out dynamically (LDS counter): a wave that finishes a short tile takes the next one, in address
    // order.  (Handing them out from the footprint's tile outwards -- longest first -- measured 3-9 % SLOWER on every
    // config: neighbouring tiles streamed at the same time share DRAM pages of the same stored rows.)
    for (;;) {
        int tidx = 0;
        if (lane == 0) tidx = atomicAdd(next_tile, 1);
        tidx = __builtin_amdgcn_readfirstlane(tidx);
        if (tidx >= n_rect_tiles + n_band_tiles) break;
        if (rect && tidx >= n_rect_tiles) {
            // ---- zero_fill: the cells of band tile (tidx - n_rect_tiles) outside the rectangle get zeros in the m new columns
            const int zt = h.t_lo + tidx - n_rect_tiles;
            const int zc = zt * kWaveTile + VEC * lane;
            const int zrow = min(zc, v.N - 1) / v.W, zcol = min(zc, v.N - 1) - zrow * v.W;
            const bool outside = zc < v.N && (zrow < rect_row0 || zrow > rect_row1 || zcol < rect_lo || zcol > rect_hi);  // (whole groups: W % VEC == 0)
            if (h.commit && outside) {
                float zv[VEC];
#pragma unroll
                for (int c = 0; c < VEC; ++c) zv[c] = 0.f;
                for (int j = 0; j < (rm ? 0 : m); ++j)
                    store_stream<VEC>((CHAIN ? new_cols + (size_t)j * cc->nstride : cov_dst + (size_t)(r + j) * npad) + zc, zv);
                if (CHAIN) {  // the new node's diagonal outside the rectangle is the parent state's
                    float dv[VEC];
                    load_vec<VEC>(dch->source(zt) + zc, dv);
                    store_vec<VEC>(diag_rw + zc, dv);
                }
            }
            const int zcells = __popcll(__ballot(outside)) * VEC;
            units += (unsigned long long)((h.commit && !rm) ? m : 0) * zcells;
            continue;
        }
        // band tile of this work unit (rect: of every lane's cells, and the range [bt_min, bt_max] the unit touches)
        const int g_lane = min(tidx * kWave + lane, rect_groups - 1);  // (rect only; lanes past the last group repeat it and are masked)
        const bool lane_valid = !rect || tidx * kWave + lane < rect_groups;
        int rrow = 0, rcol = 0;
        if (rect) { rrow = g_lane / rect_gpr; rcol = rect_lo + VEC * (g_lane - rrow * rect_gpr); rrow += rect_row0; }
        const int tile = rect ? 0 : h.t_lo + tidx;
        const int cell0 = rect ? rrow * v.W + rcol : tile * kWaveTile + VEC * lane;
        const int bt_lane = cell0 >> kTileShift;
        const int bt_min = rect ? __builtin_amdgcn_readfirstlane(bt_lane) : tile;
        const int bt_max = rect ? __builtin_amdgcn_readlane(bt_lane, kWave - 1) : tile;
        // mean / diag of the tile: requested now, parked in LDS after the base term (the loads have landed by then),
        // read back in the epilogue.  Kept in registers across the stream loop they were spilled to scratch, which
        // cost 10 % of the kernel (A/B with the loads ablated).
        float md_in[2][VEC];
        if (!LMASK) {
            load_vec<VEC>(v.mean + (size_t)h.env * npad + cell0, md_in[0]);
            load_vec<VEC>((CHAIN ? ((rect && v.rect_meta) ? dch->source(bt_lane, rrow, rcol) : dch->source(rect ? bt_lane : tile))
                                 : v.diag + (size_t)h.env * npad) + cell0, md_in[1]);  // (CHAIN: the parent state's)
        }

        // grid rows / columns of this unit (wave-uniform) and of this lane's VEC-cell group (rect_meta: W % VEC == 0, a group
        // lies in one grid row)
        const int urow0 = rect ? __builtin_amdgcn_readfirstlane(rrow) : (tile * kWaveTile) / v.W;
        const int urow1 = rect ? __builtin_amdgcn_readlane(rrow, kWave - 1) : min(tile * kWaveTile + kWaveTile - 1, v.N - 1) / v.W;
        const int ucol0 = rect ? (urow0 == urow1 ? __builtin_amdgcn_readfirstlane(rcol) : rect_lo) : 0;
        const int ucol1 = rect ? (urow0 == urow1 ? __builtin_amdgcn_readlane(rcol, kWave - 1) + VEC - 1 : rect_hi) : v.W - 1;
        const unsigned lpos = (unsigned)rrow | ((unsigned)rcol << 16);

        // ---- ordered compaction of the columns stored on this tile (wave-local, no barrier)
        int nact = 0;
        bool partial = false;  // (rect: some active column covers only a part of the unit: its loads are predicated per lane)
        bool span_part = false;  // (rm: ... already by its tile span -- columns written on band tiles; else the rectangle test suffices)
        for (int k0 = 0; k0 < r; k0 += kWave) {
            const int k = k0 + lane;
            bool on = false;
            bool part_k = false, part_sp = false;
            if (k < r) {
                const int sp = span_s[k];
                on = bt_max >= (sp & 0xffff) && bt_min <= (sp >> 16);
                part_k = on && !(bt_min >= (sp & 0xffff) && bt_max <= (sp >> 16));  // stored on a part of this unit's cells only
                if (rm) {
                    const unsigned w = rect_s[k];
                    const int r0 = w & 0xff, c0 = (w >> 8) & 0xff, r1 = r0 + (int)((w >> 16) & 0xff), c1 = c0 + (int)(w >> 24);
                    on = on && r1 >= urow0 && r0 <= urow1 && c1 >= ucol0 && c0 <= ucol1;
                    part_sp = part_k && w == 0xffff0000u;  // (a true rectangle lies inside its span: its test covers the span's)
                    part_k = on && (part_k || !(r0 <= urow0 && r1 >= urow1 && c0 <= ucol0 && c1 >= ucol1));
                }
            }
            partial |= __ballot(part_k) != 0ull;
            span_part |= __ballot(part_sp) != 0ull;  // (all lanes: wave-uniform)
            const unsigned long long mask = __ballot(on);
            if (on) ridx[nact + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)k;
            nact += __popcll(mask);
        }
        if (lane < 8) ridx[nact + lane] = (unsigned short)r;  // pipeline tail: zero Q row, any valid U row
        __builtin_amdgcn_wave_barrier();

        // ---- base term from the analytic prior: Wc0[i,:] = sum_b (sum_{f in block b} w_f P0[i, F_f]) L_inv[b,:]
        unsigned inmask = (1u << VEC) - 1u;  // cells of this lane inside the column range of the step
        float acc[VEC][MC];
        // MC = 9: rows of L^-1 and y in the lanes of ten registers (value j in lane j of every row of 16 lanes), see fmac_bc
        constexpr int kLR = (MC == 9) ? MC : 1;
        float lrow[kLR], yreg = 0.f;
        if constexpr (MC == 9 && !PRE) {
#pragma unroll
            for (int b = 0; b < MC; ++b) lrow[b] = Ls[b * MC + min(lane & 15, MC - 1)];
            yreg = ys[min(lane & 15, MC - 1)];
        }
#pragma unroll
        for (int c = 0; c < VEC; ++c)
#pragma unroll
            for (int j = 0; j < MC; ++j) acc[c][j] = 0.f;
        {
            int crow[VEC], ccol[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const int cell = min(cell0 + c, v.N - 1);
                crow[c] = rect ? rrow : cell / v.W;
                ccol[c] = rect ? rcol + c : cell - crow[c] * v.W;
            }
            if (rect) {
                inmask = lane_valid ? (1u << VEC) - 1u : 0u;
            } else if (v.clip_cols) {
                inmask = 0u;
#pragma unroll
                for (int c = 0; c < VEC; ++c) inmask |= ((ccol[c] >= clip_lo && ccol[c] <= clip_hi) ? 1u : 0u) << c;
            }
            // per block: 4 (padded, weight 0) footprint cells x VEC grid cells = 4*VEC independent table lookups in
            // flight, so the LDS latency is paid once per block instead of once per lookup
            // the table covers |drow| < lut_rows: decided per tile (wave-uniform) from the farthest tile / footprint rows
            const int trow0 = urow0, trow1 = urow1;
            const int dmax = max(max(abs(trow0 - h.yu), abs(trow0 - h.yd)), max(abs(trow1 - h.yu), abs(trow1 - h.yd)));
            // Rectangle tiles (MC = 9) run only on engines whose prior table covers every unit (ipp_engine.hip: rect_ok /
            // rect_tree require the complete table; a unit lies within window_rows of the footprint, |drow| <= R + 4):
            // the sqrt / exp variants are not compiled into those kernels -- 16 KB less code in kernels that are
            // larger than the instruction cache of a CU pair: fused step -4.4 % (A/B in profiles/r02_experiments.txt)
            constexpr bool kLutAll = RECT && MC == 9;
            const bool tile_lut = kLutAll || dmax < lut_rows;
            // rf = 1 (altitude <= rf_altitude): every measurement block is one cell, the other three table entries
            // carry weight 0: skip them (wave-uniform)
            // The variants are chosen ONCE per tile (both wave-uniform): decided per lookup, every one of the up to 36 VEC
            // lookup sites carried a branch around ~25 instructions of inlined sqrt / exp code -- 144 taken branches per
            // tile and a 9 000-instruction tile body (54 KB: the instruction cache of a CU pair is 64 KB).
            auto base_term = [&](auto lut_tag, auto nfc_tag) {
                constexpr bool TL = decltype(lut_tag)::value;
                constexpr int NFC = decltype(nfc_tag)::value;
                auto block_term = [&](int b, float (&cb)[VEC]) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) cb[c] = 0.f;
#pragma unroll
                    for (int a = 0; a < NFC; ++a) {
                        const int yx = fb_yx[4 * b + a];
                        const float wa = fb_w[4 * b + a];
                        const int fy = yx >> 16, fx = yx & 0xffff;
#pragma unroll
                        for (int c = 0; c < VEC; ++c) {
                            const int dr = abs(crow[c] - fy), dc = abs(ccol[c] - fx);
                            float p0;
                            // (24-bit multiply-add: v_mul_lo_u32 runs at quarter rate)
                            if constexpr (TL) p0 = lut[__umul24(dr, v.W) + dc];
                            else p0 = matern_f(dr, dc, s3, h.sv);
                            cb[c] = fmaf(wa, p0, cb[c]);
                        }
                    }
                };
                if (PRE) {
#pragma unroll
                    for (int b = 0; b < MC; ++b) {  // unrolled: acc[.][b] must be a static register index
                        if (b < m) {
                            float cb[VEC];
                            block_term(b, cb);
#pragma unroll
                            for (int c = 0; c < VEC; ++c) acc[c][b] = cb[c];
                        }
                    }
                } else if constexpr (MC == 9) {
                    // L^-1 is upper triangular (solve_wave_fast: column j of inv(C^T) has rows i <= j): block b feeds the
                    // columns j >= b only, 45 instead of 81 FMAs per cell; row b of L^-1 sits in the lanes of lrow[b] and reaches
                    // the FMAs through the DPP row broadcast (45 dependent broadcast LDS reads per tile before)
                    static_for<0, MC>([&](auto bc) {
                        constexpr int B = decltype(bc)::value;
                        if (B < m) {
                            float cb[VEC];
                            block_term(B, cb);
                            static_for<B, MC>([&](auto jc) {
                                constexpr int J = decltype(jc)::value;
#pragma unroll
                                for (int c = 0; c < VEC; ++c) fmac_bc<J>(acc[c][J], lrow[B], cb[c]);
                            });
                        }
                    });
                } else {
#pragma unroll
                    for (int b = 0; b < MC; ++b) {
                        if (b < m) {
                            float cb[VEC];
                            block_term(b, cb);
#pragma unroll
                            for (int j = b; j < MC; ++j) {
                                const float l = Ls[b * MC + j];
#pragma unroll
                                for (int c = 0; c < VEC; ++c) acc[c][j] = fmaf(cb[c], l, acc[c][j]);
                            }
                        }
                    }
                }
            };
            typedef std::integral_constant<int, 1> one_cell_t;
            typedef std::integral_constant<int, 4> four_cells_t;
            if (tile_lut) {
                if (h.rf == 1) base_term(std::true_type{}, one_cell_t{});
                else base_term(std::true_type{}, four_cells_t{});
            } else {  // (tall footprints / tables cut at 48 KiB only)
                if (h.rf == 1) base_term(std::false_type{}, one_cell_t{});
                else base_term(std::false_type{}, four_cells_t{});
            }
        }

        if (!LMASK) {
            float* st = stage_w + lane * VEC;
#pragma unroll
            for (int c = 0; c < VEC; ++c) { st[c] = md_in[0][c]; st[kWave * VEC + c] = md_in[1][c]; }
        }

        // ---- stream the stored rows: acc += row_k[cells] * Q[k,:]  (Q carries the sign of the downdate)
        if (nact > 0) {
            typedef float rowv __attribute__((ext_vector_type(VEC)));
            auto col_of = [&](int a) -> int { return __builtin_amdgcn_readfirstlane((int)ridx[min(a, nact + 7)]); };
            const int safe_k = col_of(0);  // nact > 0: the first column stored on this tile
            // Groups of KP rows, requested together and then consumed in order.  No row registers are carried
            // across the back edge: hipcc turns a carried (ping-pong) group into register copies at the loop end,
            // and each copy waits for its load, which empties the memory pipe once per iteration.  Overlap across
            // groups comes from the other waves of the CU.  Indices past nact hit the zero Q row.
            for (int a = 0; a < nact; a += KP) {
                rowv u[KP];
                int kk[KP];
                const float* rowk[KP];
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    kk[i] = col_of(a + i);
                    // padding entries (kk == r, zero Q row) read a column that IS stored on this tile: a column that is
                    // not stored here holds uninitialised memory, and NaN * 0 would poison the accumulators
                    const int ku = kk[i] < r ? kk[i] : safe_k;
                    // (CHAIN: the column's pointer from the LDS table; evaluating ChainCols::row here cost ~40 scalar
                    // instructions per column and spilled SGPRs: the tile loop was issue-bound, not memory-bound)
                    rowk[i] = CHAIN ? uni_ptr(lds.rowp[ku]) : cov_src + (size_t)ku * npad;
                }
                if ((rect || rm) && partial) {  // (units at the edge of a stored column's span / rectangle: per column, only the lanes it is stored on)
                    // The table words of all KP rows are read first (one LDS round trip for the group, not one in front of every
                    // request), and the three cases are separate loops (a wave-uniform choice per row cost a branch per row).
                    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
                    const unsigned row_bytes = (unsigned)v.Npad * 4u;  // (any bound >= the largest in-range offset + 16 that stays inside the row)
                    int kx[KP];
#pragma unroll
                    for (int i = 0; i < KP; ++i) kx[i] = kk[i] < r ? kk[i] : safe_k;
                    if (rm && !span_part) {  // rectangles only (every active column written on rectangle tiles, or covering the unit's tiles)
                        unsigned wrc[KP];
#pragma unroll
                        for (int i = 0; i < KP; ++i) wrc[i] = rect_s[kx[i]];
#pragma unroll
                        for (int i = 0; i < KP; ++i) {
                            const us2 d = __builtin_bit_cast(us2, lpos) - __builtin_bit_cast(us2, GainLds<MC>::rect_lo(wrc[i]));
                            const bool ok = inmask != 0u && __builtin_bit_cast(unsigned, __builtin_elementwise_min(d, __builtin_bit_cast(us2, GainLds<MC>::rect_ext(wrc[i])))) == __builtin_bit_cast(unsigned, d);
                            u[i] = row_load_masked<VEC>(rowk[i], row_bytes, (unsigned)cell0 * 4u, ok);
                        }
                    } else if (!rm) {  // tile spans only
                        int wsp[KP];
#pragma unroll
                        for (int i = 0; i < KP; ++i) wsp[i] = span_s[kx[i]];
#pragma unroll
                        for (int i = 0; i < KP; ++i) {
                            u[i] = row_load_masked<VEC>(rowk[i], row_bytes, (unsigned)cell0 * 4u, inmask != 0u && bt_lane >= (wsp[i] & 0xffff) && bt_lane <= (wsp[i] >> 16));
                        }
                    } else {  // both (columns written on band tiles beside columns written on rectangle tiles)
                        unsigned wrc[KP];
                        int wsp[KP];
#pragma unroll
                        for (int i = 0; i < KP; ++i) { wrc[i] = rect_s[kx[i]]; wsp[i] = span_s[kx[i]]; }
#pragma unroll
                        for (int i = 0; i < KP; ++i) {
                            const us2 d = __builtin_bit_cast(us2, lpos) - __builtin_bit_cast(us2, GainLds<MC>::rect_lo(wrc[i]));
                            const bool ok = inmask != 0u && bt_lane >= (wsp[i] & 0xffff) && bt_lane <= (wsp[i] >> 16) &&
                                            __builtin_bit_cast(unsigned, __builtin_elementwise_min(d, __builtin_bit_cast(us2, GainLds<MC>::rect_ext(wrc[i])))) == __builtin_bit_cast(unsigned, d);
                            u[i] = row_load_masked<VEC>(rowk[i], row_bytes, (unsigned)cell0 * 4u, ok);
                        }
                    }
                } else if (inmask) {  // (one exec-mask region for the whole group: clipped lanes request nothing)
#pragma unroll
                    for (int i = 0; i < KP; ++i) u[i] = __builtin_nontemporal_load(reinterpret_cast<const rowv*>(rowk[i] + cell0));
                } else {
#pragma unroll
                    for (int i = 0; i < KP; ++i) u[i] = (rowv)(0.f);
                }
                __builtin_amdgcn_sched_barrier(0);  // all KP requests leave before the first wait
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const float* __restrict__ qr = qrows + (size_t)kk[i] * QS;  // wave-uniform -> s_load
                    cfloat_p qc = qrows_c + (size_t)kk[i] * QS;
                    float qv[MC];
#pragma unroll
                    for (int j = 0; j < MC; ++j) qv[j] = QCONST ? qc[j] : qr[j];
#pragma unroll
                    for (int j = 0; j < MC; ++j)
#pragma unroll
                        for (int c = 0; c < VEC; ++c) acc[c][j] = fmaf(u[i][c], qv[j], acc[c][j]);
                }
            }
        }

        if (PRE) {
            // wait (first tile only) until wave 0 has published L^-1 and y, then Wc = (P[:,F] H_F^T) L^-1 in place:
            // column j needs the untransformed entries b <= j, so j runs downwards
            if (!solved) {
#if IPP_PHASE_TIMING
                const unsigned long long w0_ = wall_clock64();
#endif
                while (__hip_atomic_load(lds.solve_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(4);
                solved = true;
                dead = __hip_atomic_load(lds.solve_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 2;
#if IPP_PHASE_TIMING
                if (lane == 0) atomicAdd(&v.counters[6], wall_clock64() - w0_);
#endif
            }
            if constexpr (MC == 9) {
#pragma unroll
                for (int b = 0; b < MC; ++b) lrow[b] = Ls[b * MC + min(lane & 15, MC - 1)];
                yreg = ys[min(lane & 15, MC - 1)];
                linv_col<8>(acc, lrow); linv_col<7>(acc, lrow); linv_col<6>(acc, lrow); linv_col<5>(acc, lrow); linv_col<4>(acc, lrow);
                linv_col<3>(acc, lrow); linv_col<2>(acc, lrow); linv_col<1>(acc, lrow); linv_col<0>(acc, lrow);
            } else {
#pragma unroll
                for (int j = MC - 1; j >= 0; --j) {
                    float t[VEC];
#pragma unroll
                    for (int c = 0; c < VEC; ++c) t[c] = 0.f;
#pragma unroll
                    for (int b = 0; b <= j; ++b) {
                        const float l = Ls[b * MC + j];
#pragma unroll
                        for (int c = 0; c < VEC; ++c) t[c] = fmaf(acc[c][b], l, t[c]);
                    }
#pragma unroll
                    for (int c = 0; c < VEC; ++c) acc[c][j] = t[c];
                }
            }
        }
        const bool commit = h.commit && !dead;
        float mean_in[VEC], diag_in[VEC];
        unsigned mbits = 0xffu;
        if (LMASK) {
            mbits = lds.mask4[(tile - h.t_lo) * kWave + lane];
        } else {
            const float* st = stage_w + lane * VEC;
#pragma unroll
            for (int c = 0; c < VEC; ++c) { mean_in[c] = st[c]; diag_in[c] = st[kWave * VEC + c]; }
        }

        // ---- epilogue for this tile
        float dred[VEC], dmean[VEC];
        double part = 0.0;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const bool valid = (cell0 + c) < v.N && ((inmask >> c) & 1u) != 0u;  // (clipped cells: zeros in the new columns)
            float w2 = 0.f, dm = 0.f;
#pragma unroll
            for (int j = 0; j < MC; ++j) w2 = fmaf(acc[c][j], acc[c][j], w2);
            if constexpr (MC == 9) {
                dm = dot_lanes<MC>(acc[c], yreg);
            } else {
#pragma unroll
                for (int j = 0; j < MC; ++j) dm = fmaf(acc[c][j], ys[j], dm);
            }
            if (!valid) {
                w2 = 0.f; dm = 0.f;
#pragma unroll
                for (int j = 0; j < MC; ++j) acc[c][j] = 0.f;
            }
            dred[c] = w2;
            dmean[c] = dm;
            // rewards.py:11 mask from the pre-step mean and pre-step diag(P); rewards.py:23-30 trace reduction
            const bool in_mask = LMASK ? ((mbits >> c) & 1u) != 0u
                                       : (!adaptive || ((double)mean_in[c] + v.kf * (double)diag_in[c] >= v.thr));
            if (valid && in_mask) part += (double)w2;
        }
        part = wave_sum_dpp(part);
        if (lane == 0) lds.tile_red[tidx] = part;
        const int valid_cells = rect ? __popcll(__ballot(lane_valid)) * VEC : max(0, min(kWaveTile, v.N - tile * kWaveTile));
        // 4 N (r + m) + 16 N per committed step = (stored rows + m new rows + mean and diag read and
        // written) floats per touched cell.  LMASK: phase A also read mean / diag of this tile for the mask, which the
        // atomics then read again: those 2 extra floats per cell are traffic, not algorithm -- counted separately
        int in_cells = valid_cells;
        if (v.clip_cols && !rect) {  // stored rows, mean and diag are touched on the cells inside the column range only
            in_cells = 0;
#pragma unroll
            for (int c = 0; c < VEC; ++c) in_cells += __popcll(__ballot(((inmask >> c) & 1u) != 0u && (cell0 + c) < v.N));
        }
        units += (unsigned long long)(nact + (commit ? m + 4 : 2)) * in_cells + (unsigned long long)(commit ? m : 0) * (valid_cells - in_cells);
        if (LMASK && commit) extra += 2ull * in_cells;
        if (commit) {
            float outv[VEC];
            if (LMASK) {
                // in place (dst == env): diag -= |Wc_i|^2, mean += Wc_i y as read-modify-writes at L2.
                // TRANSPOSED first: a lane holds VEC consecutive cells, so "atomic add of component c" put 64 lanes at a
                // 16-byte stride over the whole 1-KiB tile, and every 64-byte segment was written VEC times: WRITE_SIZE
                // counted 4.0x the bytes for that pattern and it ran 4.3x slower than one atomic per consecutive cell
                // (tools/probes/write_probe.hip, profiles/r02_write_probe_calibration.txt).  Four cross-lane moves per
                // value (ds_bpermute, no LDS memory) make instruction c cover cells 64 c .. 64 c + 63 of the tile.
                float* dg = (CHAIN ? diag_rw : v.diag + (size_t)h.dst * npad) + (size_t)tile * kWaveTile;
                float* mu = v.mean + (size_t)h.dst * npad + (size_t)tile * kWaveTile;
                const bool no_atomics = false;
                const int comp = lane & (VEC - 1);
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    const int src = (kWave / VEC) * c + lane / VEC;  // lane that holds cell 64 c + lane of the tile
                    float d_t = 0.f, m_t = 0.f;
#pragma unroll
                    for (int q = 0; q < VEC; ++q) {
                        const float dq = __shfl(dred[q], src, kWave), mq = __shfl(dmean[q], src, kWave);
                        if (comp == q) { d_t = dq; m_t = mq; }
                    }
                    const int cell = tile * kWaveTile + kWave * c + lane;
                    if (cell < v.N && !no_atomics && (d_t != 0.f || m_t != 0.f)) {  // (clipped cells add nothing)
                        unsafeAtomicAdd(dg + kWave * c + lane, -d_t);
                        if (!(flags & IPP_COV_ONLY)) unsafeAtomicAdd(mu + kWave * c + lane, m_t);
                    }
                }
            } else if (lane_valid) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) outv[c] = diag_in[c] - dred[c];
                store_vec<VEC>((CHAIN ? diag_rw : v.diag + (size_t)h.dst * npad) + cell0, outv);
                if (!(flags & IPP_COV_ONLY)) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) outv[c] = mean_in[c] + dmean[c];
                    store_vec<VEC>(v.mean + (size_t)h.dst * npad + cell0, outv);
                }
            }
#pragma unroll
            for (int j = 0; j < MC; ++j)
                if (j < m && lane_valid) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) outv[c] = acc[c][j];
                    store_stream<VEC>((CHAIN ? new_cols + (size_t)j * cc->nstride : cov_dst + (size_t)(r + j) * npad) + cell0, outv);
                }
        }
        __builtin_amdgcn_wave_barrier();
    }

#if IPP_PHASE_TIMING
    if (lane == 0) atomicAdd(&v.counters[7], wall_clock64() - loop0_);
#endif
    // ------------------------------------------------------------------ per-item results
    // No closing barrier: a wave that has no tile left exits, freeing its slot; the last wave to arrive (LDS counter)
    // adds the per-tile sums in TILE order -- the tiles are handed out dynamically, so which wave reduced which tile
    // differs from run to run, but neither the per-tile values nor their order do: the reward is bit-reproducible by
    // construction -- and writes the item's reward, rank and the span of the appended columns.
    // byte counters: per workgroup in LDS, then one pair of global atomics by the last wave, spread over slots
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(red);  // red[0..1], zeroed with the tile counter
    if (lane == 0 && units) atomicAdd(cnt, units);
    if (lane == 0 && extra) atomicAdd(cnt + 1, extra);
    int reset_k = -1;
    if (RESET && ar->src) reset_k = __builtin_amdgcn_readfirstlane(ar->src[item]);
    // this wave's stores / atomics have landed (acknowledged by L2) before the last wave rewrites the env's planes: an
    // explicit wait, not an agent-scope release fence -- that one also writes the XCD's L2 back (buffer_wbl2), once per
    // wave of every resetting item: +13 us per fused step kernel (profiles/r02_experiments.txt)
    if (RESET && reset_k >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int arrived = 0;
    if (lane == 0) arrived = atomicAdd(done_waves, 1);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived == 0 && lane == 0) IPP_MARK(item, 6);  // first wave out
    if (arrived != nw - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (PRE) dead = __hip_atomic_load(lds.solve_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 2;  // (a wave without tiles never looked)
    const bool commit_item = h.commit && !dead;
    const unsigned new_rect = (rect && rm) ? rect_pack(rect_row0, rect_row1, rect_lo, rect_hi) : kRectFull;
    if (lane == 0) {
        IPP_MARK(item, 2);
        double tot = 0.0;
        for (int t = 0; t < (rect ? n_rect_tiles : n_band_tiles); ++t) tot += lds.tile_red[t];
        reward_out[item] = dead ? NAN : (float)(tot / (h.cost_d + 1.0));  // rewards.py:31
        if (commit_item && !CHAIN) v.rank[h.dst] = r + m;
        if (commit_item && CHAIN) { node_meta[0] = m; node_meta[1] = h.t_lo | (h.t_hi << 16); node_meta[4] = (int)new_rect; }
        unsigned long long* slot = v.counters + (size_t)(item & (kCountSlots - 1)) * 16;
        if (cnt[0]) atomicAdd(slot, cnt[0]);
        if (cnt[1]) atomicAdd(slot + 8, cnt[1]);
    }
    if (commit_item && !CHAIN && lane < m) {
        v.colspan[(size_t)h.dst * v.rank_cap + r + lane] = h.t_lo | (h.t_hi << 16);
        v.colrect[(size_t)h.dst * v.rank_cap + r + lane] = (int)new_rect;
    }
    if (RESET && reset_k >= 0) wave_reset_env(v, *ar, h.dst, reset_k, lane);  // (after the rank store above, same lane 0)
}

// Stand-alone gain kernel (after k_prepare): stages L^-1 | y, the spans and the prior table, then gain_tiles.
// q_all == v.q, passed separately so that it is a read-only kernel argument (scalar loads of the Q rows).
template <int MC, int VEC, bool RECT = false>
__global__ __launch_bounds__(512, IPP_GF_MINWAVES) void k_gain_factor(View v, const float* __restrict__ q_all, int n_items,
                                                                   unsigned flags, int lut_rows,
                                                                   float* __restrict__ reward_out) {
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gf[];
    const GainLds<MC> lds(smem_gf, v.rank_cap, 0, lut_rows * v.W, 0, blockDim.x / kWave, v.win_tiles, 0, VEC);
    if ((int)blockIdx.x >= n_items) return;
    const int item = launch_item(v, blockIdx.x, n_items);
    const int tid = threadIdx.x, T = blockDim.x;
    __builtin_amdgcn_s_dcache_inv();  // (Q through the non-coherent scalar cache: nothing of an earlier launch may be served)
    const ItemHdr h = uniform_hdr(v.hdr[item]);
    const int r = h.rank;
    if (h.m == 0 || h.status == IPP_STATUS_NOT_PD) {
        if (tid == 0) reward_out[item] = (h.status == IPP_STATUS_NOT_PD) ? NAN : 0.f;
        return;
    }
    const float* __restrict__ blk = q_all + (size_t)item * v.q_item;  // [L^-1 | y | pad | Q rows | zero rows]
    for (int i = tid; i < LQ; i += T) lds.Ls[i] = blk[i];
    if (tid == 0) { *lds.next_tile = 0; *lds.done_waves = 0; lds.red[0] = 0.0; lds.red[1] = 0.0; }
    for (int k = tid; k < r; k += T) lds.span_s[k] = v.colspan[(size_t)h.env * v.rank_cap + k];
    if (v.rect_meta)
        for (int k = tid; k < r; k += T) lds.stage_rect(k, (unsigned)v.colrect[(size_t)h.env * v.rank_cap + k]);
    if (v.clip_cols) {
        __syncthreads();
        mark_inactive_columns(lds.span_s, blk + LQ, (MC + 3) & ~3, r, h.m, tid, T);
    }
    fill_block_tables<MC>(h, lds.fb_yx, lds.fb_w);
    {
        const float s3 = (float)(kSqrt3 * v.res) / h.ls;
        for (int i = tid; i < lut_rows * v.W; i += T) {
            const int dr = i / v.W, dc = i - dr * v.W;
            lds.lut[i] = matern_f(dr, dc, s3, h.sv);
        }
    }
    __syncthreads();
    gain_tiles<MC, VEC, gf_pipe<MC, VEC>(), false, false, false, false, false, 0, RECT>(v, h, item, flags, lut_rows, lds, blk + LQ, reward_out);
}

}  // namespace ipp
