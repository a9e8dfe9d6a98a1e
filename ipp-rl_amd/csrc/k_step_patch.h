// Factor-state env step on COMPACT COLUMN PATCHES.
//
// Storage (View::patch).  A column of U appended by a step is non-zero only on the rectangle of grid rows / columns within
// window_rows of that step's footprint (two-dimensional windows).  Here the column IS that rectangle:
// column k of an env is a patch of ph x pw floats, row-major with the FIXED row stride pw,
//     U_k[row][col] = patch_k[(row - r0_k) * pw + (col - c0_k)],      (r0_k, r1_k, c0_k, c1_k) = View::colrect[env][k],
// patch_k = cov + env * cov_slot + k * pstride.  Because every patch has the same row stride, the cells of the NEW step's
// rectangle, enumerated in its own patch order (flat = prow * pw + pcol), sit at flat + shift_k in stored column k with the
// wave-uniform shift_k = (r0_new - r0_k) * pw + (c0_new - c0_k): a wave's request for a stored row is a run of consecutive bytes
// of that column (lanes outside the column's rectangle masked), whatever the grid width, and a column slot is 2.6 KB instead
// of Npad floats.  The appended columns are written the same way: m coalesced row writes per unit.
//
// k_step_patch<NW, KP, MINW, SPLIT = false>: the FUSED step, one workgroup of NW waves per item (default 3 waves at MINW = 6 waves per
// SIMD: 80 VGPRs, 8 workgroups = 2048 item slots per CU x 256).  Per item:
//   prologue   inputs + the rectangles of all stored columns in one batch of loads; wave 0 evaluates the fp64 header and hands it over
//              through LDS; the CONTRIBUTING columns (rectangle meets the footprint: the others have an exactly zero row of H U^T)
//              are compacted in increasing k and change hands (thread t takes the t-th), so the gather of HT = H_F U[F,:]^T runs over
//              n_c columns; per contributing column one 64-byte RECORD [-HT(0..11) | byte offset of its shifted patch | rectangle as
//              two packed 16-bit pairs | k] in LDS (beyond View::pcap: in the item's global scratch block); the prior table
//              P0(|drow| < plw, |dcol| < plw) is built under the gather's round trip
//   algebra    wave 0: S, Cholesky, L^-1, y in registers (solve_wave_fast, fp64); wave 1: the observation (observe_wave); the other
//              wave(s) stream from the start
//   units      k_patch_units.h (shared with k_tree_patch and the split step), drawn from an LDS ticket
//   results    last wave: reward = sum of the units' reductions IN UNIT ORDER / (cost + 1), rank, rectangles, the scheduled reset
// k_step_patch<NW, KP, MINW, SPLIT = true>: the PROLOGUE KERNEL of the split step (k_step_split.h): the same code up to the m x m
// algebra; instead of running the units it leaves header, tables, L^-1 | y and the records in the item's block of
// View::blk for the unit-parallel kernel k_step_units.
// Arithmetic per cell (prior term, order of the stored rows, L^-1 in the epilogue, reward sums in unit order) is the one of
// gain_tiles<PRE, RECT> (k_gain_factor.h); the m x m algebra and the observation are the shared device functions of k_prepare.h.
// mapping/mappings.py:178-197, planning/common/rewards.py:8-31.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include "ipp_common.h"
#include "k_gain.h"
#include "k_gain_factor.h"
#include "k_prepare.h"

namespace ipp {

constexpr int kPatchRec = 16;  // floats per column record
constexpr int kPatchKP = 8;    // stored rows requested per group (8 rows in flight per wave and the unit body fit 80 VGPRs: 6 waves per SIMD)
constexpr int kPatchRowAux = 2;                      // cache policy bits of the row requests (2: nt)
constexpr int kPatchStoreAux = IPP_NT_STORES ? 2 : 0;  // ... of the new rows' stores (1 sc0, 2 nt, 16 sc1)
constexpr int kPatchMinW = 6;  // waves per SIMD the register allocation of the streaming kernels aims at
constexpr int kPatchWavesDefault = 3;  // waves per item of the fused kernel.  THREE at six waves per SIMD = 8 workgroups per CU = 2048 item slots: one wave
                                       // for the m x m algebra, one for the observation, one that streams from the start (profiles/r04_experiments.txt 7)
constexpr int kPatchWavesPerCu = 4 * kPatchMinW;
constexpr int kPatchCtl = 32;  // control words
constexpr int kPatchMaxRank = 512;  // largest rank_cap of a patch engine (every thread tests kPatchMaxRank / threads rectangles)

// Geometry of the patches for a config (host + device).
struct PatchGeo {
    int pw, ph, pstride, plw, punits;
};
inline PatchGeo patch_geometry(int W, int H, int R) {
    PatchGeo g;
    // widest rectangle: 2 R + 5 columns (the widest unclipped footprint of m <= 9 blocks is 5 cells) with both ends moved out
    // to even columns; tallest: 2 R + 6 rows (a 6-row footprint only exists clipped at the border: fewer rows then)
    // (row strides of 28 / 32 floats -- 128-byte rows -- measured slower than the dense 26: profiles/r04_experiments.txt 9)
    g.pw = std::min((W + 1) & ~1, ((2 * R + 5 + 2) / 2) * 2);
    g.ph = std::min(H, 2 * R + 6);
    g.pstride = (g.pw * g.ph + 15) & ~15;
    g.plw = std::min(std::max(W, H), R + 7);
    g.punits = (g.ph * g.pw + 2 * kWave - 1) / (2 * kWave);
    return g;
}

struct PatchLds {
    float* rec; float* Ls; float* ys; float* lut; unsigned char* small; double* red; int* ctl; int* fb_yx; float* fb_w;
    unsigned short* ridx; double* unit_red;
    static constexpr int LQ = 96;  // L^-1 (81) | y (9) | pad
    __host__ __device__ static size_t small_bytes() { return (prep_small_bytes<9>() + 15) & ~(size_t)15; }
    __host__ __device__ static size_t bytes(int cap, int lut_floats, int waves, int units, int rank_cap) {
        size_t b = (size_t)cap * kPatchRec * 4 + LQ * 4 + (size_t)((lut_floats + 3) & ~3) * 4 + small_bytes();
        b += 4 * 8 + kPatchCtl * 4 + 2 * 4 * 9 * 4;
        b += (((size_t)waves * (rank_cap + kPatchKP) * 2) + 15) & ~(size_t)15;
        b += (size_t)units * 8;
        return (b + 15) & ~(size_t)15;
    }
    __device__ __forceinline__ PatchLds(unsigned char* base, int cap, int lut_floats, int waves, int units, int rank_cap) {
        rec = reinterpret_cast<float*>(base);
        Ls = rec + (size_t)cap * kPatchRec;
        ys = Ls + 81;
        lut = Ls + LQ;
        small = reinterpret_cast<unsigned char*>(lut + ((lut_floats + 3) & ~3));
        red = reinterpret_cast<double*>(small + small_bytes());
        ctl = reinterpret_cast<int*>(red + 4);
        fb_yx = ctl + kPatchCtl;
        fb_w = reinterpret_cast<float*>(fb_yx + 4 * 9);
        ridx = reinterpret_cast<unsigned short*>(fb_w + 4 * 9);
        unit_red = reinterpret_cast<double*>(ridx + ((((size_t)waves * (rank_cap + kPatchKP)) + 7) & ~(size_t)7));
    }
};

// The item's BLOCK of the split step (View::blk, one per dispatch position of a launch; k_step_split.h): written by the prologue
// kernel (k_step_patch<1, ., ., true>), read by the unit-parallel kernel.  Offsets in floats from the block's start; every part
// starts on a 128-byte line.
struct SplitBlk {
    static constexpr int kHdr = 0;      // 32 ints: the words below
    static constexpr int kSync = 32;    // 64 words = 32 x 8 bytes: [0] arrival counter of the units, [1 + u] reduction of unit u (fp64; <= 31 units)
    static constexpr int kTab = 96;     // fb_yx [36] | fb_w [36] | L^-1 [81] y [9] pad [6] | prior table [lutf4]: copied to LDS by every unit
    static constexpr int kTabFixed = 72 + PatchLds::LQ;
    // header words
    enum { ITEM = 0, ENV, M, NC, NUNITS, RANK, BITS, RECT, TSPAN, RESET, COST_LO, COST_HI, HDR_WORDS };
    enum { B_RF1 = 1, B_COMMIT = 2, B_DEAD = 4 };
    __host__ __device__ static int lutf4(int plw) { return (plw * plw + 3) & ~3; }
    __host__ __device__ static int rec_off(int plw) { return (kTab + kTabFixed + lutf4(plw) + 31) & ~31; }
    __host__ __device__ static size_t floats(int plw, int rank_cap) { return ((size_t)rec_off(plw) + (size_t)rank_cap * kPatchRec + 31) & ~(size_t)31; }
};

}  // namespace ipp
#include "k_patch_units.h"
namespace ipp {

// Io policy of the env step: pre-step mean / variance from the env's planes, stored rows through the item's single buffer resource
// (patch offset = scalar offset), -HT of a record from the workgroup's LDS staging, results into the env's planes and the next m
// patches of its slot.
struct StepIo {
    static constexpr bool kMean = true;
    typedef float rowv __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t row_rs;
    float* mean_rw;
    float* diag_rw;
    bool cov_only;
    bool wt_planes;  // mean / variance stores write-through (split step, envs reset by the launch)
    int m, row0_bytes, pstride_bytes;
    __device__ __forceinline__ rowv row_load(unsigned cofs, unsigned voff) const {
        // one resource for the whole item, the patch offset as the request's scalar offset (a resource per row was four
        // scalar instructions per row)
        return __builtin_bit_cast(rowv, __builtin_amdgcn_raw_buffer_load_b64(row_rs, voff, (int)cofs, kPatchRowAux));
    }
    __device__ __forceinline__ float coef(const UnitLds& ul, int a, int l15) const { return ul.rec[(size_t)a * kPatchRec + l15]; }
    __device__ __forceinline__ void load_pre(int cell0, int flat, int rrow, int rcol, float (&md)[2][2]) const {
        load_vec<2>(mean_rw + cell0, md[0]);
        load_vec<2>(diag_rw + cell0, md[1]);
    }
    __device__ __forceinline__ void store(bool commit, bool lane_valid, int cell0, int flat, unsigned flat4, const float (&acc)[2][9],
                                          const float (&md)[2][2], const float (&dred)[2], const float (&dmean)[2]) const {
        if (!commit) return;
        if (lane_valid) {
            float od[2], om[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) { od[c] = md[1][c] - dred[c]; om[c] = md[0][c] + dmean[c]; }
            if (!wt_planes) {
                store_vec<2>(diag_rw + cell0, od);
                if (!cov_only) store_vec<2>(mean_rw + cell0, om);
            } else {
                // split step, an env that this launch resets: 8-byte agent-scope stores = write-through (sc1), in memory before this
                // wave's arrival whatever XCD it runs on (k_step_split.h)
                typedef unsigned long long u64;
                __hip_atomic_store(reinterpret_cast<u64*>(diag_rw + cell0), ((u64)__float_as_uint(od[1]) << 32) | __float_as_uint(od[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!cov_only)
                    __hip_atomic_store(reinterpret_cast<u64*>(mean_rw + cell0), ((u64)__float_as_uint(om[1]) << 32) | __float_as_uint(om[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // the m new rows (buffer stores through the item's resource: the row as scalar offset, lanes outside the rectangle out of
        // range -- no 64-bit address per lane and row).  (The loop stays in this function: handed on to a helper by reference, `acc`
        // is not promoted to registers -- 86 scratch instructions in the unit body, 20 % of the step)
        typedef decltype(__builtin_amdgcn_raw_buffer_load_b64(row_rs, 0, 0, 0)) raw2;
#pragma unroll
        for (int j = 0; j < 9; ++j)
            if (j < m) {
                rowv t;
#pragma unroll
                for (int c = 0; c < 2; ++c) t[c] = acc[c][j];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(raw2, t), row_rs, lane_valid ? flat4 : 0xffffffffu, row0_bytes + j * pstride_bytes, kPatchStoreAux);
            }
    }
};

template <bool ONE>
__device__ __forceinline__ void patch_sync() {
    if (ONE) wave_lds_sync(); else __syncthreads();
}

// (KPN = 4 rows per request group: 12 workgroups of two waves per CU, 5 % faster for launches of 32768 items and 27 % slower for 4096
// -- two-wave engines take it for large launches only, ipp_info.patch_big_min_items)
// RJN: rounds of NW x 64 threads over the columns' rectangles (tests, compaction); 0 = enough for kPatchMaxRank.  An engine whose rank_cap
// fits fewer rounds runs the instantiation without the empty ones (configs[2]: 1 of 3, the headline: 2 of 3 -- they are predicated
// instructions otherwise, 2-3 % of an item's at configs[2], whose step is bound by the vector units).
template <int NW, int KPN = kPatchKP, int MINW = kPatchMinW, bool SPLIT = false, int RJN = 0>
__global__ __launch_bounds__(64 * NW, MINW) void k_step_patch(
    View v, const int* __restrict__ env_ids, int n_items, const double* __restrict__ action,
    const double* __restrict__ prev_action, const float* __restrict__ meas_noise, unsigned flags,
    int* __restrict__ status_out, float* __restrict__ reward_out, AutoReset ar) {
    constexpr int MC = 9, VEC = 2, NT = kWave * NW, KP = KPN;
    static_assert(KPN <= kPatchKP, "the row lists are padded for kPatchKP entries");
    constexpr int RJ = RJN > 0 ? RJN : (kPatchMaxRank + NT - 1) / NT;  // rectangles per thread, loaded with the inputs
    constexpr int OW = (NW > 1) ? 1 : 0;                   // the wave that evaluates the observation
    constexpr int TW = (NW > 2) ? 2 : OW;                  // the wave that fills the small per-item tables (block cells, fp64 prior of the footprint)
    constexpr bool ONE = (NW == 1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sp[];
    const PatchLds lds(smem_sp, v.pcap, SPLIT ? 0 : v.plw * v.plw, SPLIT ? 1 : NW, v.punits, v.rank_cap);  // (split step: no prior table, one list area)
    if ((int)blockIdx.x >= n_items) return;
    const int item = launch_item(v, blockIdx.x, n_items);
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    if (tid == 0) IPP_MARK(item, 0);
    IPP_WT_DECL;
    int* next_unit = lds.ctl; int* done_waves = lds.ctl + 1; int* solve_flag = lds.ctl + 2; int* obs_flag = lds.ctl + 3;
    int* wcnt = lds.ctl + 8;  // [RJ][NW] contributing columns found by wave w among its columns j
    // split step: the item's block, indexed by the DISPATCH POSITION of this workgroup (the unit kernel maps its workgroups to
    // positions the same way: no lookup of the item order in front of its first load)
    float* blk = SPLIT ? v.blk + (size_t)(v.blk_pos0 + (int)blockIdx.x) * v.blk_stride : nullptr;
    int* bh = reinterpret_cast<int*>(blk);

    // ------------------------------------------------------------------ batch 1: everything that does not depend on the footprint
    const int env0 = env_ids ? env_ids[item] : item + v.env_base;
    const bool slots_ok = env0 >= 0 && env0 < v.cap;
    const int envc = slots_ok ? env0 : 0;
    const double ax = action[3 * item + 0], ay = action[3 * item + 1], az = action[3 * item + 2];
    const double px = prev_action[3 * item + 0], py = prev_action[3 * item + 1], pz = prev_action[3 * item + 2];
    const int rank_ld = v.rank[envc];
    const double sv_d = v.prior[2 * envc + 0], ls_d = v.prior[2 * envc + 1];
    const int otid = tid - kWave * OW;
    const float eps_ld = (meas_noise && otid >= 0 && otid < MC) ? meas_noise[(size_t)item * MC + otid] : 0.f;
    const int* __restrict__ rects = v.colrect + (size_t)envc * v.rank_cap;
    unsigned rc_pre[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int k = tid + j * NT;
        rc_pre[j] = (k < v.rank_cap) ? (unsigned)rects[k] : 0u;
    }

    // ------------------------------------------------------------------ header (every thread; uniform)
    // Wave 0 evaluates the header (fp64 divisions, exp, square roots: ~600 vector instructions) and hands it to the other
    // waves through LDS; they arrive at the barrier with their batch-1 loads in flight.
    const int R = v.window_rows;
    const PrepLds<MC> pl(lds.small);
    ItemHdr h;
    if (ONE || __builtin_amdgcn_readfirstlane(wave) == 0) {
        h = make_item_header<MC, IPP_FACTOR>(v, env0, env0, slots_ok, ax, ay, az, px, py, pz, rank_ld, sv_d, ls_d, flags);
        // rectangle of this step = its patch: rows / columns within R of the footprint, the column range widened to even columns
        const int r0 = max(0, h.yu - R), r1 = min(v.H - 1, h.yd + R);
        const int c0 = max(0, h.xl - R) & ~(VEC - 1), c1 = min(v.W - 1, min(v.W - 1, h.xr + R) | (VEC - 1));
        if (h.m > 0 && (r1 - r0 + 1 > v.ph || c1 - c0 + 1 > v.pw)) {  // (cannot happen for footprints of <= MC blocks: patch_geometry)
            h.status = IPP_STATUS_BAD_FOOTPRINT; h.m = 0; h.f = 0; h.rows = 0; h.commit = 0;
        }
        if (tid == 0) {
            *pl.hs = h;
            *next_unit = 0; *done_waves = 0; *solve_flag = 0; *obs_flag = 0;
            lds.red[0] = 0.0; lds.red[1] = 0.0;
        }
    }
    if (!ONE) {
        __syncthreads();
        if (__builtin_amdgcn_readfirstlane(wave) != 0) h = *pl.hs;
    }
    h = uniform_hdr(h);
    IPP_EXIT_POINT(1);
    const int r0n = max(0, h.yu - R), r1n = min(v.H - 1, h.yd + R);
    const int c0n = max(0, h.xl - R) & ~(VEC - 1), c1n = min(v.W - 1, min(v.W - 1, h.xr + R) | (VEC - 1));
    const int hn = r1n - r0n + 1, wn = c1n - c0n + 1;
    const int m = h.m, f = h.f, r = h.rank;
    if (m == 0) {  // bad footprint: no step, but a scheduled reset still happens
        if (tid == 0) {
            v.hdr[item] = h;
            if (status_out) status_out[item] = h.status;
            reward_out[item] = 0.f;
            if (SPLIT) { bh[SplitBlk::ITEM] = item; bh[SplitBlk::NUNITS] = 0; }  // (no unit runs)
        }
        patch_sync<ONE>();  // (every thread holds its copy of prev_action)
        if ((flags & IPP_UPDATE_PREV) && tid == 0) {
            double* pw = const_cast<double*>(prev_action);
            pw[3 * item + 0] = ax; pw[3 * item + 1] = ay; pw[3 * item + 2] = az;
        }
        if (ar.src && tid < kWave) {
            const int k = __builtin_amdgcn_readfirstlane(ar.src[item]);
            if (k >= 0 && h.env >= 0 && h.env < v.cap) wave_reset_env(v, ar, h.env, k, tid);
        }
        return;
    }
    if (tid == 0) IPP_MARK(item, 3);
    const bool cov_only = (flags & IPP_COV_ONLY) != 0;
    const float* mean_env = v.mean + (size_t)h.env * v.Npad;
    const float* gt_env = gt_plane(v, h.env);
    float* slot = v.cov + (size_t)h.env * v.cov_slot;
    // records that do not fit the LDS staging: the item's global scratch block (split step: the record area of the item's block, which
    // receives EVERY record -- record a at blk_rec + a * 16 -- the first pcap of them are staged in LDS as well, for the m x m algebra)
    float* blk_rec = SPLIT ? blk + SplitBlk::rec_off(v.plw) : nullptr;
    const int cap = v.pcap, pw = v.pw;
    float* ovf = SPLIT ? blk_rec + (size_t)cap * kPatchRec : v.q + (size_t)item * v.q_item;

    // ------------------------------------------------------------------ batch 2: inputs of the observation (lanes of wave OW)
    ObsRegs oregs;
    {
        const int ly = div_small(max(otid, 0), h.w), lx = max(otid, 0) - ly * h.w;
        oregs.gt = (!cov_only && otid >= 0 && otid < f) ? gt_env[(h.yu + ly) * v.W + h.xl + lx] : 0.f;  // simulations/__init__.py:24-25
        const Block mob = block_of(min(max(otid, 0), m - 1), h.nx, h.rf, h.w, h.h);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int aa = min(a, mob.count() - 1);
            const int cell = (h.yu + mob.y0 + blk_dy(aa, mob.bw)) * v.W + h.xl + mob.x0 + blk_dx(aa, mob.bw);
            oregs.mean[a] = (!cov_only && otid >= 0 && otid < m && a < mob.count()) ? mean_env[cell] : 0.f;  // H x, mappings.py:195
        }
        oregs.eps = eps_ld;
    }

    // ------------------------------------------------------------------ columns that reach the footprint, in increasing k
    // (a column whose rectangle misses the footprint has an exactly zero row of H U^T: it cannot change this step)
    bool con[RJ];
    unsigned long long bal[RJ];
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int k = tid + j * NT;
        const unsigned rc = rc_pre[j];
        const int r0k = rc & 0xff, r1k = (rc >> 8) & 0xff, c0k = (rc >> 16) & 0xff, c1k = rc >> 24;
        con[j] = k < r && r0k <= h.yd && r1k >= h.yu && c0k <= h.xr && c1k >= h.xl;
        bal[j] = __ballot(con[j]);
        if (!ONE && lane == 0) wcnt[j * NW + wave] = __popcll(bal[j]);
    }
    const int ttid = tid - kWave * TW;
    if (ttid >= 0 && ttid < MC) {  // measurement blocks of the footprint as flat (cell, weight) tables + the tables of the m x m algebra
        const Block bb = block_of(min(ttid, m - 1), h.nx, h.rf, h.w, h.h);  // sensor_models.py:57-79
        for (int a = 0; a < 4; ++a) {
            const int aa = min(a, bb.count() - 1);
            const int ly = bb.y0 + blk_dy(aa, bb.bw), lx = bb.x0 + blk_dx(aa, bb.bw);
            lds.fb_yx[4 * ttid + a] = (h.yu + ly) | ((h.xl + lx) << 16);  // grid row | grid column << 16 (the packing of the rectangle tests)
            lds.fb_w[4 * ttid + a] = (ttid < m && a < bb.count()) ? (float)bb.weight : 0.f;
            if (ttid < m) pl.bfi[4 * ttid + a] = bfi_pack(ly, lx, h.w);
        }
        if (ttid < m) { pl.bcnt[ttid] = bb.count(); pl.bwt[ttid] = bb.weight; }
    }
    patch_sync<ONE>();
    if (tid == 0) IPP_MARK(item, 4);
    IPP_EXIT_POINT(2);
    int pos[RJ];
    int n_c = 0;
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        int before = 0, all = 0;
        if (ONE) {
            all = __popcll(bal[j]);
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const int c = wcnt[j * NW + w];
                before += (w < wave) ? c : 0;
                all += c;
            }
        }
        pos[j] = n_c + before + __popcll(bal[j] & ((1ull << lane) - 1ull));
        n_c += all;
    }
    // The contributing columns change hands: thread t takes the t-th of them (column index and rectangle through LDS: the
    // index list in the row-list area of the unit loop, the rectangles in the fp64 scratch of the m x m algebra, both idle
    // until the barrier below), so that the gather runs over n_c columns instead of all r of them -- with one column in
    // three contributing it was 3.8 rounds of 9 .. 36 requests per item, 15 % of the kernel's vector instructions.
    unsigned short* klist = lds.ridx;
    unsigned* rclist = reinterpret_cast<unsigned*>(pl.S);  // (S .. ktab: 2952 bytes)
    static_assert(4 * kPatchMaxRank <= (3 * MC * (MC + 1) + 3 * MC + 2 * 4 * MC) * 8, "rectangle list does not fit the fp64 scratch");
#pragma unroll
    for (int j = 0; j < RJ; ++j)
        if (con[j]) { klist[pos[j]] = (unsigned short)(tid + j * NT); rclist[pos[j]] = rc_pre[j]; }
    patch_sync<ONE>();
    // (no barrier behind the reads: nothing below writes the two lists' areas before the barrier that ends the gather -- the
    // records go to lds.rec, the prior tables to lds.lut and pl.ktab, which starts behind the rectangle list)
    static_assert(4 * kPatchMaxRank <= (3 * MC * (MC + 1) + 3 * MC + 4 * MC) * 8, "rectangle list reaches pl.ktab");

    // ------------------------------------------------------------------ gather HT = H_F U[F,:]^T for the contributing columns
    // Lanes <-> columns (rounds of 64), the footprint's blocks and cells are wave-uniform loop counters, every request is an
    // unconditional buffer load whose offset is pushed out of range where the column is not stored.  The waves of the item SHARE
    // a round: wave w takes the measurement blocks i = w (mod NW) of all its columns and writes those entries of the records -- with
    // the usual 30-60 contributing columns one wave issued all 9 .. 36 requests per column while the others waited at the barrier.
    const int wave_u = ONE ? 0 : __builtin_amdgcn_readfirstlane(wave);
    auto mine = [&](int i) { return ONE || (i % NW) == wave_u; };
    const auto slot_rs = __builtin_amdgcn_make_buffer_rsrc(slot, 0, 0x7ffffff0, 0x00020000);
    const auto row_rs = __builtin_amdgcn_make_buffer_rsrc(slot - v.pstride, 0, 0x7ffffff0, 0x00020000);  // (records hold offsets from here)
    // (cells and weights of the measurement blocks from the LDS tables filled in front of the barrier above: evaluating
    // block_of per request site -- divisions by nx, bw -- was 5000 instructions of a 17000-instruction kernel)
    auto gather_issue = [&](unsigned rc, int k, bool on, float (&l)[MC][4]) {
        // rectangle as (first row | first column << 16) and (rows - 1 | columns - 1 << 16): a cell (row | column << 16) lies inside
        // iff min(cell - first, extent) == cell - first in both halves (two packed instructions instead of four compares); a lane
        // without a contributing column gets a rectangle nothing lies in
        const unsigned first = on ? (rc & 0x00ff00ffu) : 0xffffffffu;
        const unsigned extent = on ? (((rc >> 8) & 0x00ff00ffu) - (rc & 0x00ff00ffu)) : 0u;
        const int base4 = (k * v.pstride - (int)(rc & 0xff) * pw - (int)((rc >> 16) & 0xff)) * 4;  // patch_k[(fy - r0k) * pw + (fx - c0k)] = slot[base + fy * pw + fx]
        typedef unsigned short us2g __attribute__((ext_vector_type(2)));
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
                        const int cell4 = (int)((yx & 0xffffu) * (unsigned)pw + (yx >> 16)) * 4;  // (scalar)
                        const us2g d = __builtin_bit_cast(us2g, yx) - __builtin_bit_cast(us2g, first);
                        const bool in = __builtin_bit_cast(unsigned, __builtin_elementwise_min(d, __builtin_bit_cast(us2g, extent))) == __builtin_bit_cast(unsigned, d);
                        const unsigned voff = in ? (unsigned)(base4 + cell4) : 0xffffffffu;
                        l[i][a] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(slot_rs, voff, 0, 0));
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto gather_store = [&](unsigned rc, int k, bool on, int a_pos, const float (&l)[MC][4]) {
        if (!on) return;
        float* dst = (a_pos < cap) ? lds.rec + (size_t)a_pos * kPatchRec : nullptr;
        float* gdst = (SPLIT || a_pos >= cap) ? (SPLIT ? blk_rec + (size_t)a_pos * kPatchRec : ovf + (size_t)(a_pos - cap) * kPatchRec) : nullptr;
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            if (i < m && mine(i)) {
                const int cnt = uni(pl.bcnt[i]);
                float t = l[i][0];
                if (cnt > 1) t += l[i][1];
                if (cnt > 2) t += l[i][2] + l[i][3];
                const float val = -(t * lds.fb_w[4 * i]);  // block weight; sign of the downdate folded in
                if (dst) dst[i] = val;
                if (gdst) gdst[i] = val;
            }
        }
        if (wave_u != 0) return;
        // wave 0: the entries no block fills and the column's address / rectangle
#pragma unroll
        for (int i = 0; i < 12; ++i)
            if (i >= m) {
                if (dst) dst[i] = 0.f;
                if (gdst) gdst[i] = 0.f;
            }
        const int r0k = rc & 0xff, r1k = (rc >> 8) & 0xff, c0k = (rc >> 16) & 0xff, c1k = rc >> 24;
        const int shift = (r0n - r0k) * pw + (c0n - c0k);
        const float4 meta = make_float4(__int_as_float((k * v.pstride + shift + v.pstride) * 4),  // byte offset of the shifted patch from one patch in front of the slot (>= 0: the scalar offset of the row requests)
                                        __int_as_float(r0k | (c0k << 16)),                           // rectangle: first row | first column << 16
                                        __int_as_float((r1k - r0k) | ((c1k - c0k) << 16)),          //            rows - 1 | columns - 1 << 16
                                        __int_as_float(k));
        if (dst) reinterpret_cast<float4*>(dst)[3] = meta;
        if (gdst) reinterpret_cast<float4*>(gdst)[3] = meta;
    };
    {
        float l0[MC][4];
        const bool any0 = n_c > 0;  // (wave-uniform)
        const bool on0 = lane < n_c;
        const int k0 = on0 ? (int)klist[lane] : 0;
        const unsigned rc0 = on0 ? rclist[lane] : 0u;
        if (any0) gather_issue(rc0, k0, on0, l0);
        // ---- under the round trip: prior table, footprint tables of the m x m algebra, the padding record
        {
            const float s3 = (float)(kSqrt3 * v.res) / h.ls;
            const int lw = v.plw;
            float* lut_dst = SPLIT ? blk + SplitBlk::kTab + SplitBlk::kTabFixed : lds.lut;  // (split step: straight into the item's block)
            for (int i = tid; i < lw * lw; i += NT) {
                const int dr = div_small(i, lw), dc = i - dr * lw;
                lut_dst[i] = matern_f(dr, dc, s3, h.sv);
            }
            if (ttid >= 0 && ttid < f) { const int ky = div_small(ttid, h.w); pl.ktab[ttid] = matern_d(ky, ttid - ky * h.w, v.res, sv_d, ls_d); }
        }
        if (tid == 0) IPP_MARK(item, 5);
        if (any0) gather_store(rc0, k0, on0, lane, l0);
    }
#pragma unroll 1
    for (int t0 = kWave; t0 < n_c; t0 += kWave) {  // (wave-uniform; the lists stay in place until the barrier below)
        const int t = t0 + lane;
        const bool on = t < n_c;
        const int kk = on ? (int)klist[t] : 0;
        const unsigned rc = on ? rclist[t] : 0u;
        float l[MC][4];
        gather_issue(rc, kk, on, l);
        gather_store(rc, kk, on, t, l);
    }
    if ((flags & IPP_UPDATE_PREV) && tid == 0) {
        // (every thread took its copy of prev_action in batch 1, in front of the barrier above)
        double* pwr = const_cast<double*>(prev_action);
        pwr[3 * item + 0] = ax; pwr[3 * item + 1] = ay; pwr[3 * item + 2] = az;
    }
    const int n_lds = min(n_c, cap), n_ovf = n_c - n_lds;
    if (n_ovf > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // overflow records are read back by the other waves
    patch_sync<ONE>();
    if (tid == 0) IPP_MARK(item, 1);
    IPP_WT(6);
    IPP_EXIT_POINT(3);

    // ------------------------------------------------------------------ m x m algebra (wave 0) / observation (wave OW)
    float* linv_dst = SPLIT ? blk + SplitBlk::kTab + 72 : lds.Ls;  // (split step: L^-1 | y straight into the item's block)
    int status_w0 = 0;  // (wave 0: the status of the m x m algebra)
    if (ONE) {
        observe_wave<MC>(v, h, flags, lds.small, oregs);
        status_w0 = solve_wave_fast<MC>(v, h, item, flags, lds.small, lds.rec, 1, kPatchRec, linv_dst, linv_dst + 81, nullptr, status_out,
                                        nullptr, n_lds, n_ovf > 0 ? ovf : nullptr, n_ovf);
        wave_lds_sync();
        if (lane == 0) *solve_flag = (status_w0 == IPP_STATUS_NOT_PD) ? 2 : 1;
        wave_lds_sync();
        if (lane == 0) IPP_MARK(item, 7);
    } else if (wave == 0) {
        status_w0 = solve_wave_fast<MC>(v, h, item, flags, lds.small, lds.rec, 1, kPatchRec, linv_dst, linv_dst + 81, nullptr, status_out,
                                        obs_flag, n_lds, n_ovf > 0 ? ovf : nullptr, n_ovf);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(solve_flag, status_w0 == IPP_STATUS_NOT_PD ? 2 : 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (lane == 0) IPP_MARK(item, 7);
    } else if (wave == OW) {
        observe_wave<MC>(v, h, flags, lds.small, oregs);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(obs_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    IPP_WT(7);
    IPP_EXIT_POINT(4);
    const UnitGeo ug = unit_geometry(r0n, c0n, hn, wn);
    const int n_units = ug.n_units;

    if constexpr (SPLIT) {
        // ---------------------------------------------------------------- split step: the rest of the item's block (wave 0), and out
        if (!ONE && __builtin_amdgcn_readfirstlane(wave) != 0) return;
        const bool dead_p = status_w0 == IPP_STATUS_NOT_PD;
        if (lane < 4 * MC) {  // footprint tables of the units' prior term
            bh[SplitBlk::kTab + lane] = lds.fb_yx[lane];
            blk[SplitBlk::kTab + 36 + lane] = lds.fb_w[lane];
        }
        if (lane == 0) {
            bh[SplitBlk::ITEM] = item; bh[SplitBlk::ENV] = h.env; bh[SplitBlk::M] = m; bh[SplitBlk::NC] = n_c;
            bh[SplitBlk::NUNITS] = n_units; bh[SplitBlk::RANK] = r;
            bh[SplitBlk::BITS] = ((h.rf == 1) ? SplitBlk::B_RF1 : 0) | ((h.commit != 0) ? SplitBlk::B_COMMIT : 0) | (dead_p ? SplitBlk::B_DEAD : 0);
            bh[SplitBlk::RECT] = (int)rect_pack(r0n, r1n, c0n, c1n);
            bh[SplitBlk::TSPAN] = pl.hs->t_lo | (pl.hs->t_hi << 16);
            bh[SplitBlk::RESET] = ar.src ? ar.src[item] : -1;
            const double cost_d = pl.hs->cost_d;
            bh[SplitBlk::COST_LO] = __double2loint(cost_d); bh[SplitBlk::COST_HI] = __double2hiint(cost_d);
            bh[SplitBlk::kSync] = 0;  // arrival counter of the units
            IPP_MARK(item, 2);
        }
        return;
    } else {
    // rectangles and patch offsets of the first 128 records, record a in lane a & 63 of set a >> 6 (read by the unit loop through
    // v_readlane: no LDS round trip per stored row)
    const int n_fast = min(n_lds, 2 * kWave);
    unsigned mcofs[2], mlo[2], mex[2];
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const int a = p2 * kWave + lane;
        mcofs[p2] = 0u; mlo[p2] = 0x0000ffffu; mex[p2] = 0u;  // (empty rectangle)
        if (a < n_fast) {
            const float4 mt = *reinterpret_cast<const float4*>(lds.rec + (size_t)a * kPatchRec + 12);
            mcofs[p2] = __float_as_uint(mt.x); mlo[p2] = __float_as_uint(mt.y); mex[p2] = __float_as_uint(mt.z);
        }
    }

    // ------------------------------------------------------------------ units of the new patch (k_patch_units.h)
    const bool commit_u = h.commit != 0;
    const int env_u = h.env;
    StepIo io;
    io.row_rs = row_rs;
    io.mean_rw = v.mean + (size_t)env_u * v.Npad;
    io.diag_rw = v.diag + (size_t)env_u * v.Npad;
    io.cov_only = cov_only;
    io.wt_planes = false;
    io.m = m;
    io.row0_bytes = (r + 1) * v.pstride * 4;  // first new row, from one patch in front of the slot (row_rs)
    io.pstride_bytes = v.pstride * 4;
    UnitArgs ua;
    ua.m = m; ua.rf1 = (h.rf == 1); ua.adaptive = (flags & IPP_ADAPTIVE) != 0; ua.commit_u = commit_u;
    ua.n_c = n_c; ua.n_fast = n_fast; ua.cap = cap; ua.ovf = ovf;
    ua.next_unit = next_unit; ua.solve_flag = solve_flag; ua.item = item;
    ua.ridx = lds.ridx + (size_t)wave * (v.rank_cap + KP);
    const UnitLds ul = {lds.rec, lds.Ls, lds.ys, lds.lut, lds.fb_yx, lds.fb_w};
    unsigned long long units = 0, needed = 0;
    bool dead = false;
    patch_units<KP>(v, ul, lds.unit_red, io, ua, ug, mcofs, mlo, mex, units, needed, dead);
    IPP_EXIT_POINT(5);
    IPP_WT_RESET;
    IPP_WT_COUNT(11, 1);

    // ------------------------------------------------------------------ per-item results (last wave to arrive)
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(lds.red);
    if (lane == 0 && units) { atomicAdd(cnt, units); atomicAdd(cnt + 1, needed); }
    int reset_k = -1;
    if (ar.src) reset_k = __builtin_amdgcn_readfirstlane(ar.src[item]);
    // this wave's stores have landed before the last wave rewrites the env's planes (explicit wait, not an agent-scope
    // release fence: that one also writes the XCD's L2 back, once per wave of every resetting item)
    if (reset_k >= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int arrived = 0;
    if (lane == 0) arrived = atomicAdd(done_waves, 1);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived == 0 && lane == 0) IPP_MARK(item, 6);
    if (arrived != NW - 1) { IPP_WT_FLUSH(lane); return; }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    dead = __hip_atomic_load(solve_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 2;  // (a wave without units never looked)
    const bool commit_item = commit_u && !dead;
    const double cost_d = pl.hs->cost_d;  // (the header's other words are re-read from LDS: kept in SGPRs they stay live across the whole kernel)
    const int t_span = pl.hs->t_lo | (pl.hs->t_hi << 16);
    if (lane == 0) {
        IPP_MARK(item, 2);
        double tot = 0.0;
        for (int t = 0; t < n_units; ++t) tot += lds.unit_red[t];  // unit order: bit-reproducible whatever wave took which unit
        reward_out[item] = dead ? NAN : (float)(tot / (cost_d + 1.0));  // rewards.py:31
        if (commit_item) v.rank[env_u] = r + m;
        if (cnt[0]) {  // (per-item totals: this workgroup is the only one of the launch that owns the item index -- no contention)
            unsigned long long* ic = v.item_counts + 2 * (size_t)item;
            atomicAdd(ic, cnt[0]);      // (no return value: the wave does not wait for the round trip; nobody else adds to this line)
            atomicAdd(ic + 1, cnt[1]);
        }
    }
    if (commit_item && lane < m) {
        v.colspan[(size_t)env_u * v.rank_cap + r + lane] = t_span;
        v.colrect[(size_t)env_u * v.rank_cap + r + lane] = (int)rect_pack(r0n, r1n, c0n, c1n);
    }
    if (reset_k >= 0) wave_reset_env(v, ar, env_u, reset_k, lane);  // (after the rank store above, same lane 0)
    IPP_WT(8);
    IPP_WT_FLUSH(lane);
    }
}

// Patch-layout factor state -> dense P = P0 - U U^T (ipp_read_cov_dense: tests, np.diag(state), feature planes).
__global__ __launch_bounds__(256) void k_read_cov_patch(View v, int env, float* __restrict__ out) {
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= v.N || j >= v.N) return;
    const double sv = v.prior[2 * env + 0], ls = v.prior[2 * env + 1];
    const int ri = i / v.W, ci = i - ri * v.W, rj = j / v.W, cj = j - rj * v.W;
    double acc = matern_d(ri - rj, ci - cj, v.res, sv, ls);
    const float* U = v.cov + (size_t)env * v.cov_slot;
    const int r = v.rank[env];
    for (int k = 0; k < r; ++k) {
        const unsigned rc = (unsigned)v.colrect[(size_t)env * v.rank_cap + k];
        if (!rect_has(rc, ri, ci) || !rect_has(rc, rj, cj)) continue;
        const int r0 = rc & 0xff, c0 = (rc >> 16) & 0xff;
        const float* p = U + (size_t)k * v.pstride;
        acc -= (double)p[(ri - r0) * v.pw + (ci - c0)] * (double)p[(rj - r0) * v.pw + (cj - c0)];
    }
    out[(size_t)i * v.N + j] = (float)acc;
}

}  // namespace ipp
