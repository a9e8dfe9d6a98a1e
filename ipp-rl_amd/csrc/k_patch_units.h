// The unit body of the patch kernels -- ONE body for the fused env step (k_step_patch.h), the tree-search predict step
// (k_tree_patch.h) and the unit-parallel kernel of the split env step (k_step_split.h): Wc = (P[:,F] H_F^T) L^-1 on the cells of the new
// patch, the masked trace reduction, and the writes of the step.
// mapping/mappings.py:185-197 (Wc, P' = P - Wc Wc^T, x' = x + Wc L^-T v), planning/common/rewards.py:8-31 (mask, reward).
//
// What differs between the kernels sits behind the `Io` policy: where the pre-step mean / variance of a cell come from, how a stored
// row is addressed (one buffer resource per item + scalar offset, or a 64-bit offset from View::cov), where a column record's -HT
// values are read (the workgroup's LDS staging or the item's global block) and where the results go.
//
//   * A UNIT is 64 lanes x 2 consecutive cells of the new rectangle enumerated row-major over ITS OWN width wn (index -> (prow, pcol)
//     through a per-item reciprocal), not over the fixed patch stride pw: 3.8 units per item at the headline instead of 4.6.  The
//     address of a cell in a (shifted) stored patch is still prow * pw + pcol.
//   * The records that meet a unit's rows are a 64-bit mask per page of 64 records (their rectangles sit in the lanes of two registers:
//     two compares and a ballot), walked with s_ff1 / s_bitset0 -- the record index of a row is born in an SGPR.  Order of
//     accumulation: increasing record index, whatever kernel runs the unit: results are bit-identical between the three.
//   * Request groups of KP rows (all KP requests leave before the first wait); the remainder of a page in a group of 2, 4 or KP rows
//     whose surplus rows repeat the group's first record with the request masked off.  A row's -HT values live in the lanes of ONE
//     register (value l & 15 in lane l) and reach the 18 FMAs through v_fmac_f32_dpp row_newbcast.
//   * records beyond the two register pages (more than 128 contributing columns, or beyond the LDS staging) take a list-based path.
#pragma once
#include <type_traits>
#include "ipp_common.h"
#include "k_gain.h"

namespace ipp {

constexpr int kUnitDivShift = 18;  // idx / wn == (idx * wdiv) >> 18, wdiv = ceil(2^18 / wn) (exact for every index of a patch: patch_units_exact, checked per engine)

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
};
__device__ __forceinline__ UnitGeo unit_geometry(int r0n, int c0n, int hn, int wn) {
    UnitGeo g;
    g.r0n = r0n; g.c0n = c0n; g.hn = hn; g.wn = wn;
    g.n_units = patch_unit_count(hn, wn);
    g.wdiv = ((1u << kUnitDivShift) + (unsigned)wn - 1u) / (unsigned)max(wn, 1);
    return g;
}

// Everything of the item a unit needs besides the Io policy.
struct UnitArgs {
    int m;              // measurements of the step
    bool rf1;           // resolution factor 1: one footprint cell per block
    bool adaptive;      // masked reward (rewards.py:8-12)
    bool commit_u;      // the step writes its results (unless the solve reports a non-PD S)
    int n_c;            // contributing columns = records
    int n_fast;         // records whose offset / rectangle sit in the register pages (<= 128)
    int cap;            // records in the LDS staging (the rest in `ovf`)
    const float* ovf;   // global block of the records beyond the LDS staging
    int* next_unit;     // LDS ticket counter (patch_units)
    int* solve_flag;    // LDS: 0 pending, 1 L^-1 / y ready, 2 S not positive definite; NULL: L^-1 / y are ready, `dead` is given
    unsigned short* ridx;  // this wave's list area (records beyond the register pages)
    int item;           // (timing builds: the per-unit trace)
};

// LDS tables a unit reads (the fused kernels carve them out of PatchLds, the split kernel out of its own small block).
struct UnitLds {
    const float* rec;    // column records staged in LDS ([cap][kPatchRec]; Io::coef decides whether they are read from here)
    const float* Ls;     // L^-1 [9][9]
    const float* ys;     // y [9]
    const float* lut;    // P0(|drow|, |dcol|), plw x plw
    const int* fb_yx;    // [9][4] footprint cells of the measurement blocks: grid row | grid column << 16
    const float* fb_w;   // [9][4] their weights
};

// ONE unit u of the new patch.  mcofs / mlo / mex: offset word, (first row | first column << 16) and (rows - 1 | columns - 1 << 16)
// of record a in lane a & 63 of page a >> 6.  Adds the floats this unit streamed (SURVEY 8(d) count) to `units` and the floats of the
// lanes that really were inside a stored column's rectangle to `needed`; `part`: the unit's masked trace reduction (fp64, every lane);
// solved / dead: state of the wait for the m x m algebra (first unit of a wave only).
template <int KP, class Io>
__device__ __forceinline__ void patch_unit(const View& v, const UnitLds& ul, Io& io, const UnitArgs& ua, const UnitGeo& g, const int u,
                                           const unsigned (&mcofs)[2], const unsigned (&mlo)[2], const unsigned (&mex)[2],
                                           unsigned long long& units, unsigned long long& needed, bool& solved, bool& dead, double& part) {
    constexpr int MC = 9, VEC = 2;
    typedef float rowv __attribute__((ext_vector_type(VEC)));
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & (kWave - 1);
    const float* Ls = ul.Ls; const float* ys = ul.ys; const float* lut = ul.lut;
    const int* fb_yx = ul.fb_yx; const float* fb_w = ul.fb_w;
    const int m = ua.m, pw = v.pw, lw = v.plw, cap = ua.cap, n_c = ua.n_c, n_fast = ua.n_fast;
    const float* ovf = ua.ovf;
    unsigned short* ridx = ua.ridx;
    IPP_WT_DECL;  // (phase clocks of the timing build: 0 setup + masks, 1 prior term, 2 stream, 3 mean / diag + solve wait, 4 L^-1 + epilogue, 5 stores)

    const int idx = 2 * (u * kWave + lane);
    const int prow = (int)(((unsigned)idx * g.wdiv) >> kUnitDivShift), pcol = idx - prow * g.wn;
    const bool lane_valid = prow < g.hn;
    const int rrow = g.r0n + min(prow, g.hn - 1), rcol = g.c0n + pcol;
    const int cell0 = rrow * v.W + rcol;  // (clamped for the masked lanes: any valid address)
    const int flat = min(prow, g.hn - 1) * pw + pcol;  // the lane's cells in the patch storage (fixed row stride pw)
    const int urow0 = g.r0n + (int)(((unsigned)(u * 2 * kWave) * g.wdiv) >> kUnitDivShift);
    const int urow1 = g.r0n + min(g.hn - 1, (int)(((unsigned)(u * 2 * kWave + 2 * kWave - 1) * g.wdiv) >> kUnitDivShift));
    // (a lane without cells carries a position that no rectangle holds: the rectangle test masks it with the rest)
    const unsigned lpos = lane_valid ? ((unsigned)rrow | ((unsigned)rcol << 16)) : 0xffffffffu;
    const unsigned flat4 = (unsigned)flat * 4u;  // byte offset of the lane's cells in a (shifted) patch

    // ---- records whose rectangle meets the rows of this unit: one mask per register page (lane a <-> record a; an empty page
    // entry holds first row 0xffff and never matches)
    unsigned long long pmask[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r0k = (int)(mlo[p] & 0xffffu), r1k = r0k + (int)(mex[p] & 0xffffu);
        pmask[p] = __ballot(r1k >= urow0 && r0k <= urow1);
    }
    int nact = __popcll(pmask[0]) + __popcll(pmask[1]);
    // ... and the rare records beyond the pages: ordered list in LDS
    int nslow = 0, first_slow = 0;
    if (n_c > n_fast) {
        for (int a0 = n_fast & ~(kWave - 1); a0 < n_c; a0 += kWave) {
            const int a = a0 + lane;
            bool on = false;
            if (a >= n_fast && a < n_c) {
                const float* rp = (a < cap) ? (const float*)(ul.rec + (size_t)a * kPatchRec) : (const float*)(ovf + (size_t)(a - cap) * kPatchRec);
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
        if (IPP_UNIT_SKIP(6)) { acc[0][0] = (float)rrow; acc[1][0] = (float)rcol; }
        else if (ua.rf1) base_term(std::integral_constant<int, 1>{});
        else base_term(std::integral_constant<int, 4>{});
    }

    IPP_WT(1);
    // ---- stream the stored rows: acc += patch_k[flat + shift_k] * (-HT[k,:])
    int in_rect = 0;  // lanes inside the stored columns' rectangles, summed over the rows of the unit
    auto fma_rows = [&](auto n_tag, const rowv* uu, const float* qr) {
        constexpr int N = decltype(n_tag)::value;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const float ur[VEC] = {uu[i][0], uu[i][1]};
            fmac_row<VEC, MC>(acc, qr[i], ur);
        }
    };
    // FAST: records of one register page, taken off its mask in increasing order -- index, patch offset and rectangle are scalars
    // (s_ff1 / v_readlane with a scalar lane select), -HT through the Io policy (LDS record or the item's global block)
    auto fast_group = [&](unsigned long long& mk, int page, int nreal, auto n_tag, auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int N = decltype(n_tag)::value;
        const unsigned pc = page ? mcofs[1] : mcofs[0], pl_ = page ? mlo[1] : mlo[0], pe = page ? mex[1] : mex[0];
        rowv uu[N];
        float qr[N];
        int e0 = 0;
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
            if (FULL || real) in_rect += __popcll(__ballot(ok));
            uu[i] = io.row_load(cofs, ok ? f4 : 0xffffffffu);
            qr[i] = io.coef(ul, page * kWave + e, lane & 15);  // (read while the requests are in flight)
        }
        __builtin_amdgcn_sched_barrier(0);  // all N requests leave before the first wait
        fma_rows(n_tag, uu, qr);
    };
#pragma unroll 1
    for (int page = 0; page < 2; ++page) {
        if (IPP_UNIT_SKIP(7)) break;
        unsigned long long mk = page ? pmask[1] : pmask[0];
        int left = __popcll(mk);
        typedef std::integral_constant<int, KP> n_kp;
        for (; left >= KP; left -= KP) fast_group(mk, page, KP, n_kp{}, std::true_type{});
        if (left > 0) {
            if (KP > 8 && left > 8) fast_group(mk, page, left, n_kp{}, std::false_type{});
            else if (KP > 4 && left > 4) fast_group(mk, page, left, std::integral_constant<int, (KP < 8 ? KP : 8)>{}, std::false_type{});
            else if (left > 2) fast_group(mk, page, left, std::integral_constant<int, 4>{}, std::false_type{});
            else fast_group(mk, page, left, std::integral_constant<int, 2>{}, std::false_type{});
        }
    }
    // SLOW: records from the list (generic pointers: LDS staging or the global block)
    for (int a0 = 0; a0 < nslow; a0 += KP) {
        const int ev = ridx[a0 + min(lane, KP - 1)];
        rowv uu[KP];
        float qr[KP];
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const int e = __builtin_amdgcn_readlane(ev, i);
            const float* rp = (e < cap) ? (const float*)(ul.rec + (size_t)e * kPatchRec) : (const float*)(ovf + (size_t)(e - cap) * kPatchRec);
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

    IPP_WT(2);
    IPP_WT_COUNT(9, (nact + KP - 1) / KP);
    IPP_WT_COUNT(10, 1);
    // pre-step mean and variance of the unit's cells (read behind the row stream: held across it, the four values were spilled
    // to scratch, per unit and wave; the L^-1 FMAs below cover the round trip)
    float md_in[2][VEC];
    io.load_pre(cell0, flat, rrow, rcol, md_in);
    // ---- wait (first unit of a wave in the fused kernels only) for L^-1 and y, then Wc = (P[:,F] H_F^T) L^-1 in place (column j
    // needs the entries b <= j)
    if (!solved && ua.solve_flag) {
        while (__hip_atomic_load(ua.solve_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(4);
        dead = __hip_atomic_load(ua.solve_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 2;
    }
    solved = true;
    IPP_WT(3);
    // L^-1 row b in the lanes of lrow[b], y in the lanes of yreg: ten LDS reads in flight together, the 90 + 18 FMAs take their
    // coefficients through the DPP row broadcast (45 + 9 dependent broadcast reads before)
    float lrow[MC], yreg = 0.f;
#pragma unroll
    for (int b = 0; b < MC; ++b) lrow[b] = Ls[b * MC + min(lane & 15, MC - 1)];
    if (Io::kMean) yreg = ys[min(lane & 15, MC - 1)];
    if (!IPP_UNIT_SKIP(8)) {
        linv_col<8>(acc, lrow); linv_col<7>(acc, lrow); linv_col<6>(acc, lrow); linv_col<5>(acc, lrow); linv_col<4>(acc, lrow);
        linv_col<3>(acc, lrow); linv_col<2>(acc, lrow); linv_col<1>(acc, lrow); linv_col<0>(acc, lrow);
    }
    const bool commit = ua.commit_u && !dead;

    // ---- epilogue: masked trace reduction, diag -= |Wc_i|^2, mean += Wc_i y, append the m new rows
    float dred[VEC], dmean[VEC];
    part = 0.0;
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
    IPP_WT_FLUSH(lane);
}

// The unit loop of the fused kernels: the waves of an item's workgroup draw units from an LDS ticket; unit u's reduction goes to
// unit_red[u] (the item's last wave sums them in unit order: bit-reproducible whatever wave took which unit).
template <int KP, class Io>
__device__ __forceinline__ void patch_units(const View& v, const UnitLds& ul, double* unit_red, Io& io, const UnitArgs& ua, const UnitGeo& g,
                                            const unsigned (&mcofs)[2], const unsigned (&mlo)[2], const unsigned (&mex)[2],
                                            unsigned long long& units, unsigned long long& needed, bool& dead) {
    const int lane = threadIdx.x & (kWave - 1);
    bool solved = false;
    units = 0; needed = 0; dead = false;
    int tslot = 0;
    for (;;) {
        int u = 0;
        if (lane == 0) u = atomicAdd(ua.next_unit, 1);
        u = __builtin_amdgcn_readfirstlane(u);
        if (u >= g.n_units) break;
        IPP_UNIT_TRACE(ua.item, (int)(threadIdx.x >> 6), tslot, 1, wall_clock64());
        double part;
        patch_unit<KP>(v, ul, io, ua, g, u, mcofs, mlo, mex, units, needed, solved, dead, part);
        if (lane == 0) unit_red[u] = part;
        IPP_UNIT_TRACE(ua.item, (int)(threadIdx.x >> 6), tslot, 0, ((unsigned long long)u << 32));
        IPP_UNIT_TRACE(ua.item, (int)(threadIdx.x >> 6), tslot, 3, wall_clock64());
        ++tslot;
    }
}

}  // namespace ipp
