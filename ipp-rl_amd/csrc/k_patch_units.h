// The unit loop of the patch kernels -- ONE body for the env step (k_step_patch.h) and the tree-search predict step
// (k_tree_patch.h): Wc = (P[:,F] H_F^T) L^-1 on the cells of the new patch, the masked trace reduction, and the writes of the step.
// mapping/mappings.py:185-197 (Wc, P' = P - Wc Wc^T, x' = x + Wc L^-T v), planning/common/rewards.py:8-31 (mask, reward).
//
// What differs between the two kernels sits behind the `Io` policy: where the pre-step mean / variance of a cell come from, how a
// stored row is addressed (one buffer resource per item + scalar offset, or a 64-bit offset from View::cov), and where the results go.
//
// Round 4 (profiles/r04_experiments.txt): the kernels are bound by the instructions of an item, so the loop sheds instructions:
//   * UNITS OVER THE VALID CELLS.  A unit is 64 lanes x 2 consecutive cells of the new rectangle enumerated row-major over ITS OWN width
//     wn (index -> (prow, pcol) through a per-item reciprocal), not over the fixed patch stride pw: rectangles clipped at the grid border
//     (mean 20 of 26 columns at 50x50) took units full of masked lanes -- 4.6 units per item on average, now 3.8.  The address of a cell
//     in a (shifted) stored patch is still prow * pw + pcol, so a wave's request for a stored row is ~5 runs of wn floats instead of one
//     run of 128; the padding columns of a patch are neither read nor written any more.
//   * NO PER-UNIT COMPACTION.  The records that meet a unit's rows are a 64-bit mask per page of 64 records (their rectangles sit in the
//     lanes of two registers: two compares and a ballot), walked with scalar bit instructions -- the record index of a row is born in an
//     SGPR.  Before: a compaction loop with LDS list writes per unit, and per row an LDS read of the list, a v_readlane and its hazard nops.
//     Order of accumulation unchanged (increasing record index): results are bit-identical to the round-3 kernels.
//   * records beyond the two register pages (more than 128 contributing columns, or beyond the LDS staging) keep the list-based path.
#pragma once
#include <type_traits>
#include "ipp_common.h"
#include "k_gain.h"

#ifndef IPP_UNIT_OLDMASK
#define IPP_UNIT_OLDMASK 0   // A/B: lane validity through the request offset and a second mask for the in-rectangle count
#endif
#ifndef IPP_UNIT_PWSTRIDE
#define IPP_UNIT_PWSTRIDE 0  // A/B: enumerate the units over the patch stride pw instead of the rectangle's own width
#endif
#ifndef IPP_UNIT_PIPE
#define IPP_UNIT_PIPE 0      // 1: whole request groups software-pipelined (2 KP rows in flight per wave)
#endif
#ifndef IPP_UNIT_STAGED
#define IPP_UNIT_STAGED 0    // 1: the bookkeeping of a request group stage by stage over its rows (independent instructions per stage)
#endif
#ifndef IPP_UNIT_SKIP_SURPLUS
#define IPP_UNIT_SKIP_SURPLUS 0  // 1: no FMAs for the masked surplus rows of remainder groups (-0.44 M of 31.9 M vector instructions, no time: profiles/r04_experiments.txt 19)
#endif
#ifndef IPP_UNIT_MDFIRST
#define IPP_UNIT_MDFIRST 0   // 1: mean / variance of a unit's cells requested at the start of the unit
#endif
#ifndef IPP_UNIT_NOCOUNT
#define IPP_UNIT_NOCOUNT 0   // A/B: no in-rectangle count (roofline.necessary_bytes reads 0)
#endif

namespace ipp {

constexpr int kUnitDivShift = 18;  // idx / wn == (idx * wdiv) >> 18, wdiv = ceil(2^18 / wn) (exact for every index of a patch: patch_units_exact)

// Units of a rectangle of hn rows x wn columns (wn even).
__host__ __device__ inline int patch_unit_count(int hn, int wn) { return (hn * wn + 2 * kWave - 1) / (2 * kWave); }
// every even width up to pw divides exactly through its reciprocal for every cell index a unit can hold
inline bool patch_units_exact(int pw, int ph) {
    for (int wn = 2; wn <= pw; wn += 2) {
        const unsigned wdiv = ((1u << kUnitDivShift) + wn - 1) / wn;
        for (int idx = 0; idx < ph * wn + 2 * kWave; ++idx)
            if ((int)(((unsigned)idx * wdiv) >> kUnitDivShift) != idx / wn) return false;
    }
    return true;
}

struct UnitGeo {
    int r0n, c0n, hn, wn;  // rectangle of the new patch: first grid row / column, rows, columns (even)
    int n_units;
    unsigned wdiv;
    int wreal;             // columns of the rectangle (== wn unless IPP_UNIT_PWSTRIDE)
};
__device__ __forceinline__ UnitGeo unit_geometry(int r0n, int c0n, int hn, int wn, int pw) {
    UnitGeo g;
    g.wreal = wn;
#if IPP_UNIT_PWSTRIDE
    wn = pw;  // A/B: units over the patch stride (round 3): lanes in the padding columns are masked
#endif
    g.r0n = r0n; g.c0n = c0n; g.hn = hn; g.wn = wn;
    g.n_units = patch_unit_count(hn, wn);
    g.wdiv = ((1u << kUnitDivShift) + (unsigned)wn - 1u) / (unsigned)max(wn, 1);
    return g;
}

// Everything of the item the loop needs besides the Io policy.
struct UnitArgs {
    int m;              // measurements of the step
    bool rf1;           // resolution factor 1: one footprint cell per block
    bool adaptive;      // masked reward (rewards.py:8-12)
    bool commit_u;      // the step writes its results (unless the solve reports a non-PD S)
    int n_c;            // contributing columns = records
    int n_fast;         // records whose offset / rectangle sit in the register pages (<= 128, all of them in the LDS staging)
    int cap;            // records in the LDS staging (the rest in `ovf`)
    const float* ovf;   // global block of the records beyond the LDS staging
    int* next_unit;     // LDS ticket counter
    int* solve_flag;    // LDS: 0 pending, 1 L^-1 / y ready, 2 S not positive definite
    unsigned short* ridx;  // this wave's list area (records beyond the register pages)
    int item;           // (timing builds: the per-unit trace)
};

// The loop.  mcofs / mlo / mex: offset word, (first row | first column << 16) and (rows - 1 | columns - 1 << 16) of record a in lane
// a & 63 of page a >> 6.  Returns the floats this wave streamed (SURVEY 8(d) count) in `units` and the floats of the lanes that really
// were inside a stored column's rectangle in `needed`; `dead`: S was not positive definite.
// (instruction-count build, tools/valu_sections.py: section k of the unit loop is skipped when ipp_debug_capture(1 + k) is set)
#if defined(IPP_EXIT_POINTS) && IPP_EXIT_POINTS
#define IPP_UNIT_SKIP(k) (v.dbg_capture == (k) + 1)
#else
#define IPP_UNIT_SKIP(k) false
#endif

template <int KP, class Io>
__device__ __forceinline__ void patch_units(const View& v, const PatchLds& lds, Io& io, const UnitArgs& ua, const UnitGeo& g,
                                            const unsigned (&mcofs)[2], const unsigned (&mlo)[2], const unsigned (&mex)[2],
                                            unsigned long long& units, unsigned long long& needed, bool& dead) {
    constexpr int MC = 9, VEC = 2;
    typedef float rowv __attribute__((ext_vector_type(VEC)));
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & (kWave - 1);
    const float* Ls = lds.Ls; const float* ys = lds.ys; const float* lut = lds.lut;
    const int* fb_yx = lds.fb_yx; const float* fb_w = lds.fb_w;
    const int m = ua.m, pw = v.pw, lw = v.plw, cap = ua.cap, n_c = ua.n_c, n_fast = ua.n_fast;
    const float* ovf = ua.ovf;
    unsigned short* ridx = ua.ridx;
    bool solved = false;
    units = 0; needed = 0; dead = false;
    int tslot = 0;
    IPP_WT_DECL;  // (phase clocks of the timing build: 0 setup + masks, 1 prior term, 2 stream, 3 mean / diag + solve wait, 4 L^-1 + epilogue, 5 stores)

    for (;;) {
        int u = 0;
        if (lane == 0) u = atomicAdd(ua.next_unit, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= g.n_units) break;
        IPP_UNIT_TRACE(ua.item, (int)(threadIdx.x >> 6), tslot, 1, wall_clock64());
#if defined(IPP_ISSUE_TEST) && IPP_ISSUE_TEST  // (tools/skip_timing.py: extra independent vector instructions at the top of a unit, where few registers are live)
        if (v.dbg_capture >= 11) {
            float d0 = (float)lane, d1 = 1.f, d2 = 2.f, d3 = 3.f;
            for (int q = 0; q < 32 * (v.dbg_capture - 10); ++q)
                asm volatile("v_fmac_f32 %0, %4, %4\n v_fmac_f32 %1, %4, %4\n v_fmac_f32 %2, %4, %4\n v_fmac_f32 %3, %4, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(0.f));
            if (d0 + d1 + d2 + d3 == 12345.678f) units += 1;
        }
#endif
        const int idx = 2 * (u * kWave + lane);
        const int prow = (int)(((unsigned)idx * g.wdiv) >> kUnitDivShift), pcol = idx - prow * g.wn;
#if IPP_UNIT_PWSTRIDE
        const bool lane_valid = prow < g.hn && pcol < g.wreal;
        const int rrow = g.r0n + min(prow, g.hn - 1), rcol = g.c0n + min(pcol, g.wreal - VEC);
#else
        const bool lane_valid = prow < g.hn;
        const int rrow = g.r0n + min(prow, g.hn - 1), rcol = g.c0n + pcol;
#endif
        const int cell0 = rrow * v.W + rcol;  // (clamped for the masked lanes: any valid address)
        const int flat = min(prow, g.hn - 1) * pw + pcol;  // the lane's cells in the patch storage (fixed row stride pw)
        const int urow0 = g.r0n + (int)(((unsigned)(u * 2 * kWave) * g.wdiv) >> kUnitDivShift);
        const int urow1 = g.r0n + min(g.hn - 1, (int)(((unsigned)(u * 2 * kWave + 2 * kWave - 1) * g.wdiv) >> kUnitDivShift));
        // (a lane without cells carries a position that no rectangle holds: the rectangle test masks it with the rest)
#if IPP_UNIT_OLDMASK
        const unsigned lpos = (unsigned)rrow | ((unsigned)rcol << 16);
        const unsigned flat4 = lane_valid ? (unsigned)flat * 4u : 0xffffffffu;
#else
        const unsigned lpos = lane_valid ? ((unsigned)rrow | ((unsigned)rcol << 16)) : 0xffffffffu;
        const unsigned flat4 = (unsigned)flat * 4u;  // byte offset of the lane's cells in a (shifted) patch
#endif

#if IPP_UNIT_MDFIRST
        // pre-step mean and variance of the unit's cells, requested FIRST: their round trip (2-3 us under load) runs under the row
        // stream instead of in front of the epilogue (four registers held across the stream: no spills at 79 VGPRs)
        float md_in[2][VEC];
        io.load_pre(cell0, flat, rrow, rcol, md_in);
#endif
        // ---- records whose rectangle meets the rows of this unit: one mask per register page (lane a <-> record a; an empty page
        // entry holds first row 0xffff and never matches)
        unsigned long long pmask[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int r0k = (int)(mlo[p] & 0xffffu), r1k = r0k + (int)(mex[p] & 0xffffu);
            pmask[p] = __ballot((IPP_PATCH_ABLATE & 64) ? (p * kWave + lane < n_fast) : (r1k >= urow0 && r0k <= urow1));
        }
        int nact = __popcll(pmask[0]) + __popcll(pmask[1]);
        // ... and the rare records beyond the pages: ordered list in LDS, as in round 3
        int nslow = 0, first_slow = 0;
        if (n_c > n_fast) {
            for (int a0 = n_fast & ~(kWave - 1); a0 < n_c; a0 += kWave) {
                const int a = a0 + lane;
                bool on = false;
                if (a >= n_fast && a < n_c) {
                    const float* rp = (a < cap) ? (const float*)(lds.rec + (size_t)a * kPatchRec) : (const float*)(ovf + (size_t)(a - cap) * kPatchRec);
                    const unsigned lo = __float_as_uint(rp[13]), ex = __float_as_uint(rp[14]);
                    const int r0k = lo & 0xffff, r1k = r0k + (int)(ex & 0xffff);
                    on = r1k >= urow0 && r0k <= urow1;
                }
                const unsigned long long mask = __ballot(on);
                if (on) ridx[nslow + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)a;
                if (nslow == 0 && mask) first_slow = a0 + (int)__builtin_ctzll(mask);
                nslow += __popcll(mask);
            }
            // group tail: entries past nslow repeat the first listed record with their requests masked off (0 * finite = 0)
            if (nslow > 0 && lane < KP) ridx[nslow + lane] = (unsigned short)first_slow;
            __builtin_amdgcn_wave_barrier();
            nact += nslow;
        }

        IPP_WT(0);
        // ---- base term from the analytic prior: acc[.][b] = sum_{f in block b} w_f P0[cell, F_f]  (Wc L, L^-1 in the epilogue)
        float acc[VEC][MC];
#pragma unroll
        for (int c = 0; c < VEC; ++c)
#pragma unroll
            for (int j = 0; j < MC; ++j) acc[c][j] = 0.f;
        {
            typedef float __attribute__((address_space(3))) lds_float;
            const unsigned lut_b = (unsigned)(size_t)(const lds_float*)lut, lw4 = 4u * (unsigned)lw, rcol4 = 4u * (unsigned)rcol;
            auto base_term = [&](auto nfc_tag) {
                constexpr int NFC = decltype(nfc_tag)::value;
#pragma unroll
                for (int b = 0; b < MC; ++b) {
                    if (b < m) {
                        float cb[VEC];
#pragma unroll
                        for (int c = 0; c < VEC; ++c) cb[c] = 0.f;
#pragma unroll
                        for (int a = 0; a < NFC; ++a) {
                            const int yx = fb_yx[4 * b + a];
                            const float wa = fb_w[4 * b + a];
                            // lut[|rrow - fy| * lw + |rcol + c - fx|] with the LDS byte address out of two v_sad_u32
                            // (|a - b| + c) and one multiply-add (the abs / multiply / shift form was 17 instructions per cell
                            // pair, a tenth of the kernel)
                            const unsigned fy = (unsigned)(yx & 0xffff), fx4 = (unsigned)(yx >> 16) * 4u;
                            const unsigned row_b = __umul24(__usad((unsigned)rrow, fy, 0u), lw4) + lut_b;
#pragma unroll
                            for (int c = 0; c < VEC; ++c) {
                                const unsigned addr = __usad(rcol4 + 4u * c, fx4, row_b);
                                cb[c] = fmaf(wa, *reinterpret_cast<const lds_float*>((size_t)addr), cb[c]);
                            }
                        }
#pragma unroll
                        for (int c = 0; c < VEC; ++c) acc[c][b] = cb[c];
                    }
                }
            };
            if ((IPP_PATCH_ABLATE & 4) || IPP_UNIT_SKIP(6)) { acc[0][0] = (float)rrow; acc[1][0] = (float)rcol; }
            else if (ua.rf1) base_term(std::integral_constant<int, 1>{});
            else base_term(std::integral_constant<int, 4>{});
        }

        IPP_WT(1);
        // ---- stream the stored rows: acc += patch_k[flat + shift_k] * (-HT[k,:])
        // N rows per request group (all N requests leave before the first wait): whole groups of KP without a row-exists test, the
        // remainder of a page in a group of 2, 4 or KP rows whose surplus rows repeat the group's first record with the request masked
        // off.  The row's -HT values live in the lanes of ONE register (value l & 15 in lane l) and reach the 18 FMAs through
        // v_fmac_f32_dpp row_newbcast.
        int in_rect = 0;  // lanes inside the stored columns' rectangles, summed over the rows of the unit
        // nreal < N (remainder groups): the surplus rows (requests masked off) skip their FMAs too -- a wave-uniform branch per row instead
        // of 18 vector instructions on zeros (a third of the row requests of the headline workload are such rows: 3 M of 32 M
        // vector instructions per launch, profiles/r04_valu_sections.txt)
        auto fma_rows = [&](auto n_tag, const rowv* uu, const float* qr, int nreal = 1 << 30) {
            constexpr int N = decltype(n_tag)::value;
#pragma unroll
            for (int i = 0; i < N; ++i) {
#if IPP_UNIT_SKIP_SURPLUS
                if (i >= nreal) continue;  // (wave-uniform)
#endif
                if (IPP_PATCH_ABLATE & 16) { acc[0][0] = fmaf(uu[i][0], qr[i], acc[0][0]); acc[1][0] = fmaf(uu[i][1], qr[i], acc[1][0]); continue; }
                const float ur[VEC] = {uu[i][0], uu[i][1]};
                fmac_row<VEC, MC>(acc, qr[i], ur);
            }
        };
        // FAST: records of one register page, taken off its mask in increasing order -- index, patch offset and rectangle are scalars
        // (s_ff1 / v_readlane with a scalar lane select), -HT from the LDS record
        auto fast_group = [&](unsigned long long& mk, int page, int nreal, auto n_tag, auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            constexpr int N = decltype(n_tag)::value;
            const unsigned pc = page ? mcofs[1] : mcofs[0], pl_ = page ? mlo[1] : mlo[0], pe = page ? mex[1] : mex[0];
            rowv uu[N];
            float qr[N];
            int e0 = 0;
#if IPP_UNIT_STAGED
            // stage by stage over the N rows of the group instead of row by row: every stage is N independent instructions, so a wave
            // that has the SIMD to itself (the tail of a launch: the heaviest items) does not wait on the chain
            // s_ff1 -> v_readlane -> v_pk_sub -> v_pk_min -> v_cmp -> v_cndmask -> buffer_load of ONE row at a time
            int es[N];
            unsigned cofs_[N], lo_[N], ex_[N];
            bool okb[N];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool real = FULL || i < nreal;
                int e = e0;
                if (real) { e = (int)__builtin_ctzll(mk); asm("s_bitset0_b64 %0, %1" : "+s"(mk) : "s"(e)); }
                if (i == 0) e0 = e;
                es[i] = e;
            }
#pragma unroll
            for (int i = 0; i < N; ++i) {
                cofs_[i] = (unsigned)__builtin_amdgcn_readlane((int)pc, es[i]);
                lo_[i] = (unsigned)__builtin_amdgcn_readlane((int)pl_, es[i]);
                ex_[i] = (unsigned)__builtin_amdgcn_readlane((int)pe, es[i]);
            }
            us2 dd[N];
#pragma unroll
            for (int i = 0; i < N; ++i) dd[i] = __builtin_bit_cast(us2, lpos) - __builtin_bit_cast(us2, lo_[i]);
            us2 mm[N];
#pragma unroll
            for (int i = 0; i < N; ++i) mm[i] = __builtin_elementwise_min(dd[i], __builtin_bit_cast(us2, ex_[i]));
#pragma unroll
            for (int i = 0; i < N; ++i) okb[i] = __builtin_bit_cast(unsigned, mm[i]) == __builtin_bit_cast(unsigned, dd[i]);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool real = FULL || i < nreal;
                const unsigned f4 = real ? flat4 : 0xffffffffu;
#if !IPP_UNIT_NOCOUNT
                if (real) in_rect += __popcll(__ballot(okb[i]));
#endif
                uu[i] = io.row_load(cofs_[i], okb[i] ? f4 : 0xffffffffu);
            }
#pragma unroll
            for (int i = 0; i < N; ++i) qr[i] = lds.rec[(size_t)(page * kWave + es[i]) * kPatchRec + (lane & 15)];
            __builtin_amdgcn_sched_barrier(0);
            fma_rows(n_tag, uu, qr);
            return;
#endif
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const bool real = FULL || i < nreal;  // (wave-uniform)
                int e = e0;
                if (real) {
                    e = (int)__builtin_ctzll(mk);
                    asm("s_bitset0_b64 %0, %1" : "+s"(mk) : "s"(e));  // (mk &= mk - 1 is three scalar instructions)
                }
                if (i == 0) e0 = e;
                const unsigned cofs = (unsigned)__builtin_amdgcn_readlane((int)pc, e);
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)pl_, e);
                const unsigned ex = (unsigned)__builtin_amdgcn_readlane((int)pe, e);
                const us2 d = __builtin_bit_cast(us2, lpos) - __builtin_bit_cast(us2, lo);
                // inside the column's rectangle: one compare whose result is the lane mask of the request AND the count of the lanes
                // that fetch; a surplus row of a remainder group (wave-uniform) requests nothing
                const bool ok = __builtin_bit_cast(unsigned, __builtin_elementwise_min(d, __builtin_bit_cast(us2, ex))) == __builtin_bit_cast(unsigned, d);
                const unsigned f4 = (FULL || real) ? flat4 : 0xffffffffu;
#if IPP_UNIT_OLDMASK
                in_rect += __popcll(__ballot(ok && lane_valid && (FULL || real)));
#elif !IPP_UNIT_NOCOUNT
                if (FULL || real) in_rect += __popcll(__ballot(ok));
#endif
                if (IPP_PATCH_ABLATE & 1) uu[i] = (rowv)(__uint_as_float(cofs) * 1e-30f + (ok ? 1.f : 0.f));
                else uu[i] = io.row_load(cofs, ok ? f4 : 0xffffffffu);
                qr[i] = lds.rec[(size_t)(page * kWave + e) * kPatchRec + (lane & 15)];  // (read while the requests are in flight)
            }
            __builtin_amdgcn_sched_barrier(0);  // all N requests leave before the first wait
            if (FULL) fma_rows(n_tag, uu, qr); else fma_rows(n_tag, uu, qr, nreal);
        };
        // (IPP_UNIT_PIPE) whole groups software-pipelined: the requests of group g + 1 leave before the FMAs of group g, so a wave keeps
        // 2 KP rows in flight and the round trip of a group hides behind the arithmetic of its predecessor
        auto issue_full = [&](unsigned long long& mk, int page, rowv (&uu)[KP], float (&qr)[KP]) {
            const unsigned pc = page ? mcofs[1] : mcofs[0], pl_ = page ? mlo[1] : mlo[0], pe = page ? mex[1] : mex[0];
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int e = (int)__builtin_ctzll(mk);
                asm("s_bitset0_b64 %0, %1" : "+s"(mk) : "s"(e));
                const unsigned cofs = (unsigned)__builtin_amdgcn_readlane((int)pc, e);
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)pl_, e);
                const unsigned ex = (unsigned)__builtin_amdgcn_readlane((int)pe, e);
                const us2 d = __builtin_bit_cast(us2, lpos) - __builtin_bit_cast(us2, lo);
                const bool ok = __builtin_bit_cast(unsigned, __builtin_elementwise_min(d, __builtin_bit_cast(us2, ex))) == __builtin_bit_cast(unsigned, d);
#if !IPP_UNIT_NOCOUNT
                in_rect += __popcll(__ballot(ok));
#endif
                uu[i] = io.row_load(cofs, ok ? flat4 : 0xffffffffu);
                qr[i] = lds.rec[(size_t)(page * kWave + e) * kPatchRec + (lane & 15)];
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll 1
        for (int page = 0; page < 2; ++page) {
            if (IPP_UNIT_SKIP(7)) break;
            unsigned long long mk = page ? pmask[1] : pmask[0];
            int left = __popcll(mk);
            typedef std::integral_constant<int, KP> n_kp;
#if IPP_UNIT_PIPE
            if (left >= KP) {
                rowv ua[KP], ub[KP];
                float qa[KP], qb[KP];
                issue_full(mk, page, ua, qa); left -= KP;
                for (;;) {
                    if (left < KP) { fma_rows(n_kp{}, ua, qa); break; }
                    issue_full(mk, page, ub, qb); left -= KP;
                    fma_rows(n_kp{}, ua, qa);
                    if (left < KP) { fma_rows(n_kp{}, ub, qb); break; }
                    issue_full(mk, page, ua, qa); left -= KP;
                    fma_rows(n_kp{}, ub, qb);
                }
            }
#else
            for (; left >= KP; left -= KP) fast_group(mk, page, KP, n_kp{}, std::true_type{});
#endif
            if (left > 0) {
                if (KP > 8 && left > 8) fast_group(mk, page, left, n_kp{}, std::false_type{});
                else if (KP > 4 && left > 4) fast_group(mk, page, left, std::integral_constant<int, (KP < 8 ? KP : 8)>{}, std::false_type{});
                else if (left > 2) fast_group(mk, page, left, std::integral_constant<int, 4>{}, std::false_type{});
                else fast_group(mk, page, left, std::integral_constant<int, 2>{}, std::false_type{});
            }
        }
        // SLOW: records from the list (generic pointers: LDS staging or the global overflow block)
        for (int a0 = 0; a0 < nslow; a0 += KP) {
            const int ev = ridx[a0 + min(lane, KP - 1)];
            rowv uu[KP];
            float qr[KP];
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                const int e = __builtin_amdgcn_readlane(ev, i);
                const float* rp = (e < cap) ? (const float*)(lds.rec + (size_t)e * kPatchRec) : (const float*)(ovf + (size_t)(e - cap) * kPatchRec);
                const float4 mt = *reinterpret_cast<const float4*>(rp + 12);
                const unsigned cofs = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(mt.x));
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(mt.y));
                const unsigned ex = (unsigned)__builtin_amdgcn_readfirstlane(__float_as_int(mt.z));
                const us2 d = __builtin_bit_cast(us2, lpos) - __builtin_bit_cast(us2, lo);
                const bool ok = (bool)((int)(a0 + i < nslow) &
                                (int)(__builtin_bit_cast(unsigned, __builtin_elementwise_min(d, __builtin_bit_cast(us2, ex))) == __builtin_bit_cast(unsigned, d)));
                in_rect += __popcll(__ballot(ok));
                uu[i] = io.row_load(cofs, ok ? flat4 : 0xffffffffu);
                qr[i] = rp[lane & 15];
            }
            __builtin_amdgcn_sched_barrier(0);
            fma_rows(std::integral_constant<int, KP>{}, uu, qr);
        }

        IPP_UNIT_TRACE(ua.item, (int)(threadIdx.x >> 6), tslot, 0, ((unsigned long long)u << 32) | (unsigned)nact);
        IPP_UNIT_TRACE(ua.item, (int)(threadIdx.x >> 6), tslot, 2, wall_clock64());
        IPP_WT(2);
        IPP_WT_COUNT(9, (nact + KP - 1) / KP);
        IPP_WT_COUNT(10, 1);
#if !IPP_UNIT_MDFIRST
        // pre-step mean and variance of the unit's cells (read behind the row stream: held across it, the four values were spilled
        // to scratch, per unit and wave; the L^-1 FMAs below cover the round trip)
        float md_in[2][VEC];
        io.load_pre(cell0, flat, rrow, rcol, md_in);
#endif
        // ---- wait (first unit only) for L^-1 and y, then Wc = (P[:,F] H_F^T) L^-1 in place (column j needs the entries b <= j)
        if (!solved) {
            while (__hip_atomic_load(ua.solve_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(4);
            solved = true;
            dead = __hip_atomic_load(ua.solve_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 2;
        }
        IPP_WT(3);
        // L^-1 row b in the lanes of lrow[b], y in the lanes of yreg: ten LDS reads in flight together, the 90 + 18 FMAs take their
        // coefficients through the DPP row broadcast (45 + 9 dependent broadcast reads before)
        float lrow[MC], yreg = 0.f;
#pragma unroll
        for (int b = 0; b < MC; ++b) lrow[b] = Ls[b * MC + min(lane & 15, MC - 1)];
        if (Io::kMean) yreg = ys[min(lane & 15, MC - 1)];
        if (!(IPP_PATCH_ABLATE & 8) && !IPP_UNIT_SKIP(8)) {
            linv_col<8>(acc, lrow); linv_col<7>(acc, lrow); linv_col<6>(acc, lrow); linv_col<5>(acc, lrow); linv_col<4>(acc, lrow);
            linv_col<3>(acc, lrow); linv_col<2>(acc, lrow); linv_col<1>(acc, lrow); linv_col<0>(acc, lrow);
        }
        const bool commit = ua.commit_u && !dead;

        // ---- epilogue: masked trace reduction, diag -= |Wc_i|^2, mean += Wc_i y, append the m new rows
        float dred[VEC], dmean[VEC];
        double part = 0.0;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            float w2 = 0.f, dm = 0.f;
#pragma unroll
            for (int j = 0; j < MC; ++j) w2 = fmaf(acc[c][j], acc[c][j], w2);
            if (Io::kMean) dm = dot_lanes<MC>(acc[c], yreg);
            if (!lane_valid) { w2 = 0.f; dm = 0.f; }
            dred[c] = w2;
            dmean[c] = dm;
            // rewards.py:11 mask from the pre-step mean and pre-step diag(P); rewards.py:23-30 trace reduction
            const bool in_mask = !ua.adaptive || ((double)md_in[0][c] + v.kf * (double)md_in[1][c] >= v.thr);
            if (lane_valid && in_mask) part += (double)w2;
        }
        part = wave_sum_dpp(part);
        if (lane == 0) lds.unit_red[u] = part;
        const int in_cells = __popcll(__ballot(lane_valid)) * VEC;
        // (stored rows + m new rows + mean and diag read and written) floats per touched cell; `needed`: the stored rows
        // counted only on the lanes inside each column's own rectangle (the others are masked requests: never fetched)
        const int fixed = commit ? m + 4 : 2;
        units += (unsigned long long)(nact + fixed) * in_cells;
        needed += (unsigned long long)in_rect * VEC + (unsigned long long)fixed * in_cells;
        IPP_WT(4);
        if (!IPP_UNIT_SKIP(9)) io.store(commit, lane_valid, cell0, flat, flat4, acc, md_in, dred, dmean);
        __builtin_amdgcn_wave_barrier();
        IPP_WT(5);
        IPP_UNIT_TRACE(ua.item, (int)(threadIdx.x >> 6), tslot, 3, wall_clock64());
        ++tslot;
    }
    IPP_WT_FLUSH(lane);
}

}  // namespace ipp
