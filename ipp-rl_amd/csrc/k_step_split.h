// The SPLIT env step on compact column patches: prologue kernel + unit-parallel kernel.
//
// In the fused kernel (k_step_patch.h) an item holds one of 2048 workgroup slots for ~32 us of which ~13 us are its prologue -- a chain
// of dependent round trips (inputs -> header -> contributing columns -> gather -> m x m algebra) during which its three 80-register
// waves stream nothing (profiles/r04_experiments.txt 13, 16).  Here the two phases are two launches on the same stream:
//
//   P  k_step_patch<NW, KP, MINW, SPLIT = true>  ITEM-parallel, NW = 1 .. 3 waves per item (default 3: 14 us per item instead of 20-22 with
//      one): the prologue of the fused kernel, word for word (same template, same device functions: header, compaction, gather,
//      observation, S / Cholesky / L^-1 / y), which leaves the item's BLOCK in View::blk (SplitBlk, k_step_patch.h): header words,
//      footprint tables, L^-1 | y, the prior table, one 64-byte record per contributing column.  Latency-bound, ~30 MB per launch.
//   U  k_step_units                              UNIT-parallel, one wave per (item, unit): grid = 8 ceil(n / 8) x punits workgroups of 64
//      threads, workgroup b -> dispatch position and unit through the XCD-aware map below (the units of an item sit on consecutive
//      slots of one XCD, heaviest items first); copies the block's tables to LDS, reads the records' offset / rectangle words into the
//      two register pages and runs patch_unit (k_patch_units.h) -- every slot streams from its first microsecond, and the altitude-14
//      items that end a fused launch spread over as many CUs as they have units.
//
// Results are bit-identical to the fused kernel: a unit's arithmetic is the same body; the reward is summed IN UNIT ORDER by the last
// unit of the item to arrive (per-item arrival counter), which also writes rank / rectangles and runs the scheduled reset.
// Hand-off between the units of an item (cdna_hip_programming.md Guideline 16): the unit's fp64 reduction is ONE 8-byte agent-scope
// (write-through) store, every wave drains its stores (s_waitcnt vmcnt(0)) in front of its arrival (agent-scope atomic add), the last
// arriver reads the reductions with agent-scope loads.  No fence: nothing else travels between the units -- except, for an item
// whose env is reset by this launch, the units' mean / variance stores, which are then write-through as well so that the reset's
// plain stores to the same lines are the last word whatever XCDs the units ran on.
// MEASURED (profiles/r05_experiments.txt 2): U alone streams at 0.40 of 8 TB/s (fused kernel: 0.30 as one launch) with a slot duty of
// 0.88, but the step loses on every schedule (42.5 against 55.7 M env-steps/s on two groups): P is a 28-38 us latency chain per launch
// in each group's dependency chain, and it cannot be placed while the other group's U holds every wave slot.  Selected by
// IPP_SPLIT=<min items> (ipp_info.patch_split_min_items); never by default.
// mapping/mappings.py:178-197, planning/common/rewards.py:8-31.
#pragma once
#include "k_step_patch.h"

namespace ipp {

constexpr int kSplitMinWP = 6;  // waves per SIMD the prologue kernel aims at (three waves per item: 8 items per CU, 20 KB of LDS each)
constexpr int kSplitMinWU = 6;  // ... the unit kernel (it needs 64 VGPRs: the hardware then runs 8 waves per SIMD)

// LDS of a unit wave: the block's tables (fb_yx | fb_w | L^-1 | y | prior table, in the block's order) and the list area of the
// records beyond the register pages.
struct SplitLds {
    float* tab; unsigned short* ridx;
    __host__ __device__ static size_t bytes(int plw, int rank_cap) {
        return (((size_t)(SplitBlk::kTabFixed + SplitBlk::lutf4(plw)) * 4 + 15) & ~(size_t)15) + ((((size_t)(rank_cap + kPatchKP) * 2) + 15) & ~(size_t)15);
    }
    __device__ __forceinline__ SplitLds(unsigned char* base, int plw) {
        tab = reinterpret_cast<float*>(base);
        ridx = reinterpret_cast<unsigned short*>(base + (((size_t)(SplitBlk::kTabFixed + SplitBlk::lutf4(plw)) * 4 + 15) & ~(size_t)15));
    }
};

// Io policy of a unit of the split step: StepIo with the records' -HT values read from the item's block (global, L2-resident: written
// by the prologue kernel one launch earlier); for items whose env is reset by this launch StepIo::wt_planes makes the plane stores write-through.
struct SplitIo : StepIo {
    __amdgpu_buffer_rsrc_t rec_rs;
    __device__ __forceinline__ float coef(const UnitLds&, int a, int l15) const {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rec_rs, (unsigned)l15 * 4u, a * (kPatchRec * 4), 0));
    }
};

// Workgroup b of the unit kernel -> (dispatch position, unit): workgroups b, b + 8, b + 16 .. run on one XCD (observed; speed only),
// so XCD x works through the positions x, x + 8, .. with the units of a position on consecutive slots: they share the XCD's L2 for
// the item's block, the lines of the stored rows that two neighbouring units both touch, and the mean / variance planes.
__device__ __forceinline__ bool split_decode(int b, int n_pos, int punits, int& pos, int& unit) {
    const int xcd = b & 7, slot = b >> 3;
    const int j = slot / punits;
    unit = slot - j * punits;
    pos = j * 8 + xcd;
    return pos < n_pos;
}
inline int split_grid(int n_pos, int punits) { return ((n_pos + 7) / 8) * 8 * punits; }

template <int KP = kPatchKP, int MINW = kSplitMinWU>
__global__ __launch_bounds__(64, MINW) void k_step_units(View v, int n_pos, unsigned flags, float* __restrict__ reward_out, AutoReset ar) {
    constexpr int MC = 9;
    typedef unsigned long long u64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_su[];
    const SplitLds sl(smem_su, v.plw);
    const int lane = threadIdx.x;
    int pos, u;
    if (!split_decode((int)blockIdx.x, n_pos, v.punits, pos, u)) return;
    float* blk = v.blk + (size_t)(v.blk_pos0 + pos) * v.blk_stride;
    const int* bh = reinterpret_cast<const int*>(blk);
    // ---- header: one 128-byte line, word l in lane l; then everything the unit needs in ONE batch of loads
    const int hw = bh[lane & 31];
    const int n_units = __builtin_amdgcn_readlane(hw, SplitBlk::NUNITS);
    if (u >= n_units) return;
    IPP_UNIT_TRACE(v.blk_pos0 + pos, 0, u, 1, wall_clock64());
    const int item = __builtin_amdgcn_readlane(hw, SplitBlk::ITEM), env = __builtin_amdgcn_readlane(hw, SplitBlk::ENV);
    const int m = __builtin_amdgcn_readlane(hw, SplitBlk::M), n_c = __builtin_amdgcn_readlane(hw, SplitBlk::NC);
    const int r = __builtin_amdgcn_readlane(hw, SplitBlk::RANK), bits = __builtin_amdgcn_readlane(hw, SplitBlk::BITS);
    const unsigned rect = (unsigned)__builtin_amdgcn_readlane(hw, SplitBlk::RECT);
    const int reset_k = __builtin_amdgcn_readlane(hw, SplitBlk::RESET);
    const float* blk_rec = blk + SplitBlk::rec_off(v.plw);
    {   // tables -> LDS (16 bytes per lane and request)
        const int n4 = (SplitBlk::kTabFixed + SplitBlk::lutf4(v.plw)) / 4;
        const float4* src = reinterpret_cast<const float4*>(blk + SplitBlk::kTab);
        float4* dst = reinterpret_cast<float4*>(sl.tab);
        for (int i = lane; i < n4; i += kWave) dst[i] = src[i];
    }
    // offset / rectangle words of the first 128 records, record a in lane a & 63 of page a >> 6
    const int n_fast = min(n_c, 2 * kWave);
    unsigned mcofs[2], mlo[2], mex[2];
#pragma unroll
    for (int p2 = 0; p2 < 2; ++p2) {
        const int a = p2 * kWave + lane;
        mcofs[p2] = 0u; mlo[p2] = 0x0000ffffu; mex[p2] = 0u;  // (empty rectangle)
        if (a < n_fast) {
            const float4 mt = *reinterpret_cast<const float4*>(blk_rec + (size_t)a * kPatchRec + 12);
            mcofs[p2] = __float_as_uint(mt.x); mlo[p2] = __float_as_uint(mt.y); mex[p2] = __float_as_uint(mt.z);
        }
    }
    const int r0n = rect & 0xff, r1n = (rect >> 8) & 0xff, c0n = (rect >> 16) & 0xff, c1n = rect >> 24;
    const UnitGeo ug = unit_geometry(r0n, c0n, r1n - r0n + 1, c1n - c0n + 1);
    float* slot = v.cov + (size_t)env * v.cov_slot;
    SplitIo io;
    io.row_rs = __builtin_amdgcn_make_buffer_rsrc(slot - v.pstride, 0, 0x7ffffff0, 0x00020000);  // (records hold offsets from here)
    io.rec_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(blk_rec), 0, 0x7ffffff0, 0x00020000);
    io.mean_rw = v.mean + (size_t)env * v.Npad;
    io.diag_rw = v.diag + (size_t)env * v.Npad;
    io.cov_only = (flags & IPP_COV_ONLY) != 0;
    io.m = m;
    io.row0_bytes = (r + 1) * v.pstride * 4;
    io.pstride_bytes = v.pstride * 4;
    io.wt_planes = reset_k >= 0;
    UnitArgs ua;
    ua.m = m; ua.rf1 = (bits & SplitBlk::B_RF1) != 0; ua.adaptive = (flags & IPP_ADAPTIVE) != 0; ua.commit_u = (bits & SplitBlk::B_COMMIT) != 0;
    ua.n_c = n_c; ua.n_fast = n_fast; ua.cap = 0; ua.ovf = blk_rec;
    ua.next_unit = nullptr; ua.solve_flag = nullptr; ua.item = item;
    ua.ridx = sl.ridx;
    const UnitLds ul = {nullptr, sl.tab + 72, sl.tab + 72 + 81, sl.tab + SplitBlk::kTabFixed, reinterpret_cast<const int*>(sl.tab), sl.tab + 36};
    wave_lds_sync();  // the tables are in LDS
    IPP_UNIT_TRACE(v.blk_pos0 + pos, 0, u, 2, wall_clock64());
    unsigned long long units = 0, needed = 0;
    bool solved = true, dead = (bits & SplitBlk::B_DEAD) != 0;
    double part;
    patch_unit<KP>(v, ul, io, ua, ug, u, mcofs, mlo, mex, units, needed, solved, dead, part);

    // ---- publish this unit, arrive; the item's last unit finishes the item
    u64* sync = reinterpret_cast<u64*>(blk + SplitBlk::kSync);
    if (lane == 0) {
        __hip_atomic_store(sync + 1 + u, (u64)__double_as_longlong(part), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        u64* ic = v.item_counts + 2 * (size_t)item;  // (per-item totals of the byte counters: return-less adds)
        atomicAdd(ic, units);
        atomicAdd(ic + 1, needed);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY store of this wave has left before it arrives
    int arrived = 0;
    if (lane == 0) arrived = __hip_atomic_fetch_add(reinterpret_cast<int*>(sync), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    IPP_UNIT_TRACE(v.blk_pos0 + pos, 0, u, 3, wall_clock64());
    IPP_UNIT_TRACE(v.blk_pos0 + pos, 0, u, 0, (unsigned long long)item);
    if (arrived != n_units - 1) return;
    const bool commit_item = ua.commit_u && !dead;
    double pu = 0.0;
    if (lane < n_units) pu = __longlong_as_double((long long)__hip_atomic_load(sync + 1 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    double tot = 0.0;
    for (int t = 0; t < n_units; ++t) tot += bcast_lane(pu, t);  // unit order: bit-identical to the fused kernel's sum
    if (lane == 0) {
        const double cost_d = __hiloint2double(__builtin_amdgcn_readlane(hw, SplitBlk::COST_HI), __builtin_amdgcn_readlane(hw, SplitBlk::COST_LO));
        reward_out[item] = dead ? NAN : (float)(tot / (cost_d + 1.0));  // rewards.py:31
        if (commit_item) v.rank[env] = r + m;
    }
    if (commit_item && lane < m) {
        v.colspan[(size_t)env * v.rank_cap + r + lane] = __builtin_amdgcn_readlane(hw, SplitBlk::TSPAN);
        v.colrect[(size_t)env * v.rank_cap + r + lane] = (int)rect;
    }
    if (reset_k >= 0 && ar.src) wave_reset_env(v, ar, env, reset_k, lane);  // (after the rank store above, same lane 0)
}

}  // namespace ipp
