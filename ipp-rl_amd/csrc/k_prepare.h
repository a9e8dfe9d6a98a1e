// Prologue kernel: one workgroup per step item.
//   footprint / resolution factor / noise / cost            (sensors/cameras.py:34-75,122-125,
//                                                             sensors/models/sensor_models.py:27-36,
//                                                             planning/common/actions.py:8-41)
//   observation z = clip(area_downsample(gt[F]) + nv*eps)    (simulations/simulations.py:26-34,
//                                                             simulations/sensor_manipulations.py:7-57)
//   S = H P_FF H^T + R, Cholesky, L^-1, y = L^-T (z - H x)   (mapping/mappings.py:178-197)
//   Q (stored with its sign folded in) so that the streaming kernel gets Wc = base + rows * Q (mappings.py:188)
// The m x m algebra is fp64 (cond(S) ~ 400 on a first visit); outputs for the stream are fp32.
#pragma once
#include "ipp_common.h"

#ifndef IPP_PREP_MINWAVES
#define IPP_PREP_MINWAVES 4
#endif

namespace ipp {

// Weight of source index s for destination index d along one axis of OpenCV's INTER_AREA table
// (computeResizeAreaTab, opencv 4.5.2 resize.cpp): the table's three clauses evaluated directly; weights are
// computed in double and stored as float there, hence the (float) round trip.  PARITY UNPINNED (cv2 absent).
__device__ inline double area_weight(int src, int dst, int d, int s) {
    const double scale = (double)src / (double)dst;
    const double a = d * scale, b = a + scale;
    const double cell = fmin(scale, (double)src - a);
    int s1 = (int)ceil(a);
    const int s2 = min((int)floor(b), src - 1);
    s1 = min(s1, s2);
    double w = 0.0;
    if (s == s1 - 1 && s1 - a > 1e-3) w += (double)(float)((s1 - a) / cell);
    if (s >= s1 && s < s2) w += (double)(float)(1.0 / cell);
    if (s == s2 && b - s2 > 1e-3) w += (double)(float)(fmin(fmin(b - s2, 1.0), cell) / cell);
    return w;
}

// LDS bytes of the small fp64 / table scratch of prepare_item (everything except the HT / P_FF staging area).
template <int MC>
__host__ __device__ constexpr size_t prep_small_bytes() {
    return (3 * MC * (MC + 1) + 3 * MC + 2 * 4 * MC) * sizeof(double) + sizeof(ItemHdr) + 16 +
           (4 * MC + 8 * MC + ((MC + 3) & ~3)) * sizeof(int) + MC * sizeof(double);
}

// Carve of the prologue's fp64 LDS scratch (prep_small_bytes<MC>() bytes, all offsets multiples of 8 B).
template <int MC>
struct PrepLds {
    static constexpr int FC = 4 * MC;
    static constexpr int LD = MC + 1;  // padded leading dimension of the small fp64 matrices
    double* S; double* L; double* Li; double* zz; double* vv; double* yy; double* sub; double* ktab;
    ItemHdr* hs; int* okflag; int* cellidx; int* bcell; int* bfi; int* bcnt; double* bwt;
    __device__ __forceinline__ explicit PrepLds(unsigned char* small) {
        S = reinterpret_cast<double*>(small);          // [MC][LD]
        L = S + MC * LD;                               // [MC][LD]   lower Cholesky factor / work
        Li = L + MC * LD;                              // [MC][LD]   upper-triangular inverse (or S^-1)
        zz = Li + MC * LD;                             // [MC] observation
        vv = zz + MC;                                  // [MC] innovation
        yy = vv + MC;                                  // [MC]
        sub = yy + MC;                                 // [FC] ground-truth crop
        ktab = sub + FC;                               // [FC] prior between footprint cells by (|drow|, |dcol|)
        hs = reinterpret_cast<ItemHdr*>(ktab + FC);
        okflag = reinterpret_cast<int*>(hs + 1);       // [4]
        cellidx = okflag + 4;                          // [FC] flat cell index of footprint cell fi
        bcell = cellidx + FC;                          // [MC][4] flat cell indices of block i
        bfi = bcell + 4 * MC;                          // [MC][4] footprint-local indices of block i
        bcnt = bfi + 4 * MC;                           // [MC] cells in block i (1, 2 or 4)
        bwt = reinterpret_cast<double*>(bcnt + ((MC + 3) & ~3));  // [MC] weight of block i
    }
};

// Footprint cell of a measurement block in PrepLds::bfi: flat footprint-local index | column << 8 | row << 16 (the row and
// column used to be recovered by integer divisions by the footprint width, per pair and cell of S: ~1000 instructions of an
// rf = 2 item)
__device__ __forceinline__ int bfi_pack(int ly, int lx, int w) { return (ly * w + lx) | (lx << 8) | (ly << 16); }
__device__ __forceinline__ int bfi_flat(int e) { return e & 0xff; }
__device__ __forceinline__ int bfi_x(int e) { return (e >> 8) & 0xff; }
__device__ __forceinline__ int bfi_y(int e) { return e >> 16; }

// The item header of a step: footprint, resolution factor, noise variance, cost, rank / status bookkeeping -- computed
// by every thread from the same inputs (fp64 like NumPy: sensors/cameras.py:34-75,122-125, sensors/models/sensor_models.py:27-36,
// planning/common/actions.py:8-41, mapping/mappings.py:125-126).  Shared by the prologue (prepare_item_ex) and the patch
// step kernel (k_step_patch.h).
template <int MC, int MODE>
__device__ __forceinline__ ItemHdr make_item_header(const View& v, int env0, int dst0, bool slots_ok, double ax, double ay, double az,
                                                    double px, double py, double pz, int rank_ld, double sv, double ls, unsigned flags,
                                                    bool with_cost = true) {
    // with_cost = false (wave-uniform): the caller's wave does not use cost / cost_d (k_step_patch: only wave 0's copy is kept)
    constexpr int FC = 4 * MC;
    ItemHdr h;
    h.env = env0;
    h.dst = dst0;
    h.status = IPP_STATUS_OK;
    h.fallback = 0;
    h.commit = (flags & IPP_PREDICT_ONLY) ? 0 : 1;
    h.t_lo = 0;
    h.t_hi = v.n_tiles - 1;
    bool ok = isfinite(ax) && isfinite(ay) && isfinite(az) && slots_ok;
    int xl = 0, xr = 0, yu = 0, yd = 0;
    if (ok) {
        const double ext_x = 2 * az * v.tanx, ext_y = 2 * az * v.tany;           // cameras.py:44-45
        const double cells_x = floor(ext_x / v.res), cells_y = floor(ext_y / v.res);  // :63-64
        const double gx = floor(ax / v.res), gy = floor(ay / v.res);              // :66
        const double rad_x = floor(0.5 * cells_x), rad_y = floor(0.5 * cells_y);  // :67
        xl = (int)fmin(fmax(gx - rad_x, 0.0), (double)(v.W - 1));                 // :69-73
        xr = (int)fmin(fmax(gx + rad_x, 0.0), (double)(v.W - 1));
        yu = (int)fmin(fmax(gy - rad_y, 0.0), (double)(v.H - 1));
        yd = (int)fmin(fmax(gy + rad_y, 0.0), (double)(v.H - 1));
        ok = (xr >= xl) && (yd >= yu);
    }
    h.xl = xl; h.xr = xr; h.yu = yu; h.yd = yd;
    h.rf = (az > v.rf_alt) ? 2 : 1;                                               // cameras.py:125
    h.w = xr - xl + 1;
    h.h = yd - yu + 1;
    // (rf is 1 or 2 and the operands are >= 0: a shift is the division -- a runtime integer division is ~35 vector instructions,
    // the header had six of them)
    const int rsh = h.rf - 1;
    h.nx = ((h.w - 1) >> rsh) + 1;                                                // sensor_models.py:57
    h.ny = ((h.h - 1) >> rsh) + 1;
    h.m = h.nx * h.ny;                                                            // mappings.py:125-126
    h.f = h.w * h.h;
    h.nv_d = v.coeff_a * (1.0 - exp(-v.coeff_b * az));                            // sensor_models.py:30
    h.nv = (float)h.nv_d;
    double cost = 0.0;
    if (with_cost) {
        const double dx = ax - px, dy = ay - py, dz = az - pz;
        const double dist = sqrt(dx * dx + dy * dy + dz * dz);                    // actions.py:15-16
        cost = dist;
        if (flags & IPP_USE_FLIGHT_TIME) {                                        // actions.py:32-41
            const double d_acc = fmin(dist * 0.5, v.vmax * v.vmax / (2 * v.amax));
            cost = (dist - 2 * d_acc) / v.vmax + 2 * sqrt(2 * d_acc / v.amax);
        }
    }
    h.cost_d = cost;
    h.cost = (float)cost;
    h.rank = ok ? rank_ld : 0;
    h.sv = ok ? (float)sv : 0.f;
    h.ls = ok ? (float)ls : 0.f;
    if (!ok || h.m > MC || h.f > FC) h.status = IPP_STATUS_BAD_FOOTPRINT;
    if (h.status == IPP_STATUS_OK && h.rf > 1 && !(flags & IPP_COV_ONLY)) {
        // area resampler is only restated for shrinking scales
        const int ocols = (h.h + h.rf - 1) >> rsh, orows = (h.w + h.rf - 1) >> rsh;
        if (h.w < ocols || h.h < orows) h.status = IPP_STATUS_BAD_FOOTPRINT;
    }
    if (MODE == IPP_FACTOR && h.status == IPP_STATUS_OK && h.commit && h.rank + h.m > v.rank_cap) {
        h.status = IPP_STATUS_RANK_FULL;
        h.commit = 0;
    }
    if (MODE == IPP_FACTOR && v.window_rows > 0 && ok) {
        // the appended columns are kept on the grid rows within window_rows of the footprint (whole tiles)
        const int row_lo = max(0, yu - v.window_rows), row_hi = min(v.H - 1, yd + v.window_rows);
        if (v.tile_shift >= 0) {  // (uniform)
            h.t_lo = (row_lo * v.W) >> v.tile_shift;
            h.t_hi = ((row_hi + 1) * v.W - 1) >> v.tile_shift;
        } else {
            h.t_lo = (row_lo * v.W) / v.tile_cells;
            h.t_hi = ((row_hi + 1) * v.W - 1) / v.tile_cells;
        }
    }
    h.rows = (MODE == IPP_FACTOR) ? h.rank : h.f;
    if (h.status == IPP_STATUS_BAD_FOOTPRINT) { h.m = 0; h.f = 0; h.rows = 0; h.commit = 0; }
    return h;
}

struct NoMidWork { __device__ __forceinline__ void operator()(const ItemHdr&) const {} };

// The per-item prologue as a device function so that it can run as its own kernel (k_prepare: dense state, exact
// factor state, ipp_observe) or at the head of the fused factor step kernel (k_step_factor.h).
//   small        LDS scratch of prep_small_bytes<MC>() bytes (16-byte aligned); the ItemHdr lives inside it
//   big, si, sk  staging area of HT (factor: HT(i,k) = big[i*si + k*sk]; si <= 0 selects the row-major
//                [MC][round4(rank)] layout) or P_FF (dense: [FC][FC+1])
//   q_out        Q rows [k][QS] (global scratch, or LDS where q_out == big with sk == QS: in place)
//   linv_f/_f2, y_f/_f2   fp32 copies of L^-1 and y for the streaming kernels (second pointers may be null)
//   span_s       optional LDS array receiving the tile spans of the stored columns (factor)
// Returns a pointer to the item header in LDS; header.m == 0 or status NOT_PD means nothing to stream.
//   FRONT_ONLY   stop after the gather (HT / P_FF staged in `big`, z and the innovation in the scratch): the fused
//                kernel lets one wave finish the m x m algebra (solve_wave) while the others already stream
//   mid_work     called once between issuing the footprint-dependent loads and consuming them (free compute slot)
//   CHAIN / cc   factor columns come from a chained tree state (ChainCols) instead of the env's own slab; rank_chain
//                is then the state's column count
// WAVE (NT == 64): the caller is ONE wave of a larger workgroup (the producer wave of the persistent kernel of rounds 2-4, deleted): every barrier of the
// prologue becomes a wave-level LDS fence instead of a workgroup barrier.
__device__ __forceinline__ void wave_lds_sync();
template <bool WAVE>
__device__ __forceinline__ void prep_sync() {
    if (WAVE) wave_lds_sync(); else __syncthreads();
}

// DEFER (fused step kernel): the observation -- ground-truth crop, INTER_AREA resampling, noise, innovation -- is NOT
// evaluated here.  Its inputs are loaded by the lanes of WAVE 1 (ObsRegs, out) and the caller lets that wave run
// observe_wave() after the prologue's last barrier, in parallel with wave 0's S / Cholesky / L^-1 (solve_wave_fast waits
// for the innovation only in front of y = L^-T v) and with the streams of waves 2, 3: the stream does not need the
// observation, and for items above rf_altitude (40 % of them) it was 7 of the prologue's ~15 us.
struct ObsRegs { float gt, eps, mean[4]; };
template <int MC> struct PrepLds;
template <int MC>
__device__ __forceinline__ void observe_wave(const View& v, const ItemHdr& h, unsigned flags, unsigned char* small, const ObsRegs o);

template <int MC, int MODE, int NT, bool FRONT_ONLY, typename Mid, bool CHAIN = false, bool WAVE = false, bool DEFER = false>
__device__ __forceinline__ ItemHdr* prepare_item_ex(const View& v, const int item, const int* __restrict__ env_ids,
                                                    const int* __restrict__ dst_ids, const double* __restrict__ action,
                                                    const double* __restrict__ prev_action,
                                                    const float* __restrict__ meas_noise, unsigned flags,
                                                    int* __restrict__ status_out, float* __restrict__ obs_out,
                                                    int* __restrict__ obs_m, int* __restrict__ obs_shape,
                                                    unsigned char* small, float* big, int si, int sk, float* q_out,
                                                    float* linv_f, float* linv_f2, float* y_f, float* y_f2, int* span_s,
                                                    Mid mid_work, const ChainCols* cc = nullptr, int rank_chain = 0,
                                                    ObsRegs* obs_regs = nullptr) {
    static_assert(!DEFER || (FRONT_ONLY && !WAVE && NT >= 2 * kWave && 4 * MC <= kWave), "deferred observation: fused kernel only");
    constexpr int kPrepThreads = NT;
    constexpr int FC = 4 * MC;
    constexpr int LD = MC + 1;
    const PrepLds<MC> pl(small);
    double* S = pl.S; double* L = pl.L; double* Li = pl.Li; double* zz = pl.zz; double* vv = pl.vv; double* yy = pl.yy;
    double* sub = pl.sub; double* ktab = pl.ktab; ItemHdr* hs = pl.hs; int* okflag = pl.okflag; int* cellidx = pl.cellidx;
    int* bcell = pl.bcell; int* bfi = pl.bfi; int* bcnt = pl.bcnt; double* bwt = pl.bwt;

    const int tid = threadIdx.x;
    IPP_TICK_DECL(tick);

    // The prologue is a chain of dependent global-memory round trips, and under the streaming kernels' load every
    // trip costs 4-5 us.  Loads are therefore issued in two batches: (1) everything whose address does not depend
    // on the footprint, then the header arithmetic (done redundantly by every thread: no LDS broadcast, no
    // barrier), (2) everything that does (ground-truth crop, mean at the footprint, the first pass of U rows).
    constexpr int MP = (MC <= 16) ? 16 : 32;  // lanes per streaming index: i = tid % MP, no runtime division
    constexpr int KS = kPrepThreads / MP;     // rows of U gathered per step of a pass
    constexpr int UN = (kPrepThreads >= 256) ? 12 : (WAVE ? 12 : 8);  // rows per thread and pass, all in flight together

    // ------------------------------------------------------------------ batch 1
    const int env0 = env_ids ? env_ids[item] : item + v.env_base;
    const int dst0 = dst_ids ? dst_ids[item] : env0;
    const bool slots_ok = env0 >= 0 && env0 < v.cap && dst0 >= 0 && dst0 < v.cap;
    const int envc = slots_ok ? env0 : 0;  // speculative loads stay in bounds
    const double ax = action[3 * item + 0], ay = action[3 * item + 1], az = action[3 * item + 2];
    const double px = prev_action[3 * item + 0], py = prev_action[3 * item + 1], pz = prev_action[3 * item + 2];
    const int rank_ld = CHAIN ? rank_chain : ((MODE == IPP_FACTOR) ? v.rank[envc] : 0);
    const double sv = v.prior[2 * envc + 0], ls = v.prior[2 * envc + 1];
    const int otid = DEFER ? tid - kWave : tid;  // thread index as the observation's inputs see it (DEFER: lanes of wave 1)
    const float eps_ld = (meas_noise && otid >= 0 && otid < MC) ? meas_noise[(size_t)item * MC + otid] : 0.f;
    const int* __restrict__ span = v.colspan + (size_t)envc * v.rank_cap;
    const int* __restrict__ rects = v.colrect + (size_t)envc * v.rank_cap;
    int sp_pre[UN];  // tile spans of this thread's first-pass rows (factor)
    unsigned rc_pre[UN];  // ... and their rectangles (View::rect_meta)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int k = tid / MP + u * KS;
        sp_pre[u] = CHAIN ? cc->span(min(k, max(rank_chain - 1, 0))) : ((MODE == IPP_FACTOR && k < v.rank_cap) ? span[k] : 0);
        rc_pre[u] = kRectFull;
        if (MODE == IPP_FACTOR && v.rect_meta)
            rc_pre[u] = CHAIN ? cc->rect(min(k, max(rank_chain - 1, 0))) : (k < v.rank_cap ? (unsigned)rects[k] : kRectFull);
    }

    // ------------------------------------------------------------------ header (every thread, fp64 like NumPy)
    ItemHdr h = make_item_header<MC, MODE>(v, env0, dst0, slots_ok, ax, ay, az, px, py, pz, rank_ld, sv, ls, flags);
    h = uniform_hdr(h);  // every thread computed the same values: keep them in SGPRs from here on
    if (tid == 0) {
        *hs = h;
        okflag[0] = 1;
    }
    const int m = h.m, f = h.f, r = h.rank;
    double* dbg = v.dbg + (size_t)item * (2 * MC * MC + 2 * MC);

    if (m == 0) {  // bad footprint: nothing to stream
        if (tid == 0) {
            v.hdr[item] = h;
            if (status_out) status_out[item] = h.status;
            if (obs_m) obs_m[item] = 0;
        }
        for (int i = tid; i < MC * MC; i += kPrepThreads) { linv_f[i] = 0.f; if (linv_f2) linv_f2[i] = 0.f; }
        for (int i = tid; i < MC; i += kPrepThreads) { y_f[i] = 0.f; if (y_f2) y_f2[i] = 0.f; }
        prep_sync<WAVE>();  // *hs visible to the caller's threads
        if ((flags & IPP_UPDATE_PREV) && tid == 0) {
            double* pw = const_cast<double*>(prev_action);
            pw[3 * item + 0] = ax; pw[3 * item + 1] = ay; pw[3 * item + 2] = az;
        }
        return hs;
    }

    const float* mean_env = v.mean + (size_t)h.env * v.Npad;
    const float* gt_env = gt_plane(v, h.env);
    const float* cov_env = v.cov + (size_t)h.env * v.cov_slot;
    const double R = (double)(h.rf * h.rf * h.rf) * h.nv_d;  // sensor_models.py:36
    const bool cov_only = (flags & IPP_COV_ONLY) != 0;
    // Batch 1 is complete here on every path the hardware takes, but not on every path hipcc's wait-count tracking
    // sees: it then waits for the spans at their first use, inside the gather, and that vmcnt(0) also drains the
    // ground-truth / mean requests issued in front of the gather (one more round trip).  Touching the spans here puts
    // that wait where it costs nothing.
#pragma unroll
    for (int u = 0; u < UN; ++u) { asm volatile("" : : "v"(sp_pre[u])); asm volatile("" : : "v"(rc_pre[u])); }
    IPP_TICK(v, 1, tick);
    if ((FRONT_ONLY || WAVE) && threadIdx.x == 0) IPP_MARK(item, 3);

    // ------------------------------------------------------------------ batch 2: footprint-dependent loads
    // (a) ground-truth crop (simulations/__init__.py:24-25), one cell per thread (f <= FC <= threads)
    float gt_ld[(FC + kPrepThreads - 1) / kPrepThreads];
#pragma unroll
    for (int q = 0; q < (FC + kPrepThreads - 1) / kPrepThreads; ++q) {
        const int fi = (DEFER ? otid : tid) + q * kPrepThreads;
        const int ly = max(fi, 0) / h.w, lx = max(fi, 0) - ly * h.w;
        gt_ld[q] = (!cov_only && fi >= 0 && fi < f) ? gt_env[(h.yu + ly) * v.W + h.xl + lx] : 0.f;
    }
    // (b) mean over the cells of measurement block tid (H x, mappings.py:195)
    const Block myb = block_of(min(tid, m - 1), h.nx, h.rf, h.w, h.h);  // sensor_models.py:57-79
    const Block mob = DEFER ? block_of(min(max(otid, 0), m - 1), h.nx, h.rf, h.w, h.h) : myb;
    float mean_ld[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int aa = min(a, mob.count() - 1);
        const int cell = (h.yu + mob.y0 + aa / mob.bw) * v.W + h.xl + mob.x0 + aa % mob.bw;
        mean_ld[a] = (!cov_only && otid >= 0 && otid < m && a < mob.count()) ? mean_env[cell] : 0.f;
    }
    if (DEFER) {
        obs_regs->gt = gt_ld[0]; obs_regs->eps = eps_ld;
#pragma unroll
        for (int a = 0; a < 4; ++a) obs_regs->mean[a] = mean_ld[a];
    }
    // (c) factor: first pass of the HT gather, HT[i][k] = sum_{cells of block i} w * U[k][cell]   (m x r).
    // A column contributes only where it is stored (its tile span): cells outside hold nothing and count as zero.
    const int gi = tid & (MP - 1);
    const Block gb = block_of(min(gi, m - 1), h.nx, h.rf, h.w, h.h);
    int gc[4], gtile[4], grow[4], gcol[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int aa = min(a, gb.count() - 1);
        grow[a] = h.yu + gb.y0 + aa / gb.bw;
        gcol[a] = h.xl + gb.x0 + aa % gb.bw;
        gc[a] = grow[a] * v.W + gcol[a];
        gtile[a] = gc[a] / v.tile_cells;
    }
    const bool rect_meta = v.rect_meta != 0;  // (columns written on rectangle tiles hold nothing outside their rectangle)
    const int gcnt = gb.count();
    const float gw = (float)gb.weight;
    // All requests of a pass leave before the first is waited for.  A column is requested only where it is stored
    // (its tile span; about 40 % of the columns cover a given footprint), and rf = 1 items (one cell per block,
    // wave-uniform) request only that cell.  Written as "test, load, add" per cell, hipcc waits for every load
    // before it issues the next one: 12 .. 36 dependent round trips per pass.
    const bool one_cell = (h.rf == 1);
    auto gather_issue = [&](int k0, const int (&sp)[UN], const unsigned (&rc)[UN], float (&l)[UN][4]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = min(k0 + u * KS, r - 1);
            const float* row = CHAIN ? cc->row(k) : cov_env + (size_t)k * v.Npad;
            const int lo = sp[u] & 0xffff, hi = sp[u] >> 16;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                l[u][a] = 0.f;
                if (a == 0 || (!one_cell && a < gcnt))
                    if (gtile[a] >= lo && gtile[a] <= hi && (!rect_meta || rect_has(rc[u], grow[a], gcol[a]))) l[u][a] = row[gc[a]];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto gather_sum = [&](const float (&l)[UN][4], float (&sacc)[UN]) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            float t = l[u][0];
            if (gcnt > 1) t += l[u][1];
            if (gcnt > 2) t += l[u][2] + l[u][3];
            sacc[u] = t;
        }
    };
    auto gather_rows = [&](int k0, const int (&sp)[UN], const unsigned (&rc)[UN], float (&sacc)[UN]) {
        float l[UN][4];
        gather_issue(k0, sp, rc, l);
        gather_sum(l, sacc);
    };
    float sacc0[UN];
    // (ipp_observe stops behind the observation: it must not touch the covariance state -- on a patch-layout engine the column
    // addresses of this band-tile gather do not even exist, a slot holds rank_cap * pstride floats)
    const bool gather_on = (MODE == IPP_FACTOR) && gi < m && r > 0 && !obs_out;
    // rows past the rank clamp to r - 1: their span was read for an unused column, the value is discarded below.
    // The pass stays in flight across mid_work (block tables and prior table: arithmetic on the header): its 1.5 us run
    // under the 4 us round trip.  (While mid_work also loaded the mean / diag of the window for the mask this cost 48 live
    // registers and 12 spills and was 4 % slower; the mask is per tile now.)
    {
        float l0[UN][4];
        if (gather_on) gather_issue(tid / MP, sp_pre, rc_pre, l0);
        mid_work(h);
        if (gather_on) gather_sum(l0, sacc0);
    }

    // ------------------------------------------------------------------ footprint tables (no divisions later)
    for (int fi = tid; fi < f; fi += kPrepThreads) {
        const int ly = fi / h.w, lx = fi - ly * h.w;
        cellidx[fi] = (h.yu + ly) * v.W + h.xl + lx;
    }
    if (tid < m) {
        bcnt[tid] = myb.count();
        bwt[tid] = myb.weight;
        for (int a = 0; a < 4; ++a) {
            const int aa = min(a, myb.count() - 1);
            const int ly = myb.y0 + aa / myb.bw, lx = myb.x0 + aa % myb.bw;
            bfi[4 * tid + a] = bfi_pack(ly, lx, h.w);
            bcell[4 * tid + a] = (h.yu + ly) * v.W + h.xl + lx;
        }
    }
    if (MODE == IPP_FACTOR && tid < f) ktab[tid] = matern_d(tid / h.w, tid % h.w, v.res, sv, ls);
#pragma unroll
    for (int q = 0; q < (FC + kPrepThreads - 1) / kPrepThreads; ++q) {
        const int fi = tid + q * kPrepThreads;
        float gt_f = gt_ld[q];
        asm volatile("" : "+v"(gt_f));  // (conversion pinned here, see eps below: hipcc otherwise waits for this load right at its request)
        if (!DEFER && !cov_only && fi < f) sub[fi] = (double)gt_f;
    }
    prep_sync<WAVE>();
    if constexpr (!DEFER) {

    // ------------------------------------------------------------------ observation + innovation
    // rf = 2: the INTER_AREA weights of both axes (orows x h and ocols x w entries, <= 2 MC each) are evaluated one
    // per lane into LDS first (the S / L scratch is free here): computed serially by the m output lanes they were
    // ~30 fp64 calls with divisions per lane, 10 us of the prologue for 40 % of the items
    double* wyt = L;           // [orows][h.h]
    double* wxt = L + 2 * MC;  // [ocols][h.w]
    if (!cov_only && h.rf > 1) {
        const int ocols = (h.h + h.rf - 1) / h.rf, orows = (h.w + h.rf - 1) / h.rf;
        for (int idx = tid; idx < orows * h.h; idx += kPrepThreads) wyt[idx] = area_weight(h.h, orows, idx / h.h, idx % h.h);
        for (int idx = tid; idx < ocols * h.w; idx += kPrepThreads) wxt[idx] = area_weight(h.w, ocols, idx / h.w, idx % h.w);
        prep_sync<WAVE>();
    }
    if (!cov_only) {
        if (tid < m) {
            double val;
            if (h.rf == 1) {
                val = sub[tid];
            } else {
                // cv2.resize(sub, dsize=(ceil(h/rf), ceil(w/rf))) -> width=ceil(h/rf), height=ceil(w/rf)
                const int ocols = (h.h + h.rf - 1) / h.rf;
                const int orow = tid / ocols, ocol = tid - orow * ocols;
                val = 0.0;
                for (int sy = 0; sy < h.h; ++sy) {
                    const double wy = wyt[orow * h.h + sy];
                    if (wy == 0.0) continue;
                    for (int sx = 0; sx < h.w; ++sx) {
                        const double wx = wxt[ocol * h.w + sx];
                        if (wx != 0.0) val += sub[sy * h.w + sx] * wx * wy;
                    }
                }
            }
            // (the conversion is pinned here: hipcc otherwise moves it up into the predicated block of the load and
            // waits for the load there, a full memory round trip in front of every other request of the prologue)
            float eps_f = eps_ld;
            asm volatile("" : "+v"(eps_f));
            const double eps = (double)eps_f;
            if (flags & IPP_GIVEN_OBSERVATION)
                val = eps;  // caller supplies z (update_grid_map(pos, z), mappings.py:114-121)
            else
                val = fmin(fmax(val + h.nv_d * eps, 0.0), 1.0);  // sensor_manipulations.py:56-57 (variance used as std)
            zz[tid] = val;
            double hx = 0.0;
            for (int a = 0; a < myb.count(); ++a) hx += myb.weight * (double)mean_ld[a];
            vv[tid] = val - hx;  // mappings.py:195
        }
    } else if (tid < MC) {
        zz[tid] = 0.0;
        vv[tid] = 0.0;
    }
    }  // (!DEFER)
    IPP_TICK(v, 2, tick);
    if ((FRONT_ONLY || WAVE) && threadIdx.x == 0) IPP_MARK(item, 4);

    if (obs_out) {  // ipp_observe: observation only
        prep_sync<WAVE>();
        if (tid < MC) obs_out[(size_t)item * MC + tid] = (tid < m) ? (float)zz[tid] : 0.f;
        if (tid == 0) {
            obs_m[item] = m;
            if (obs_shape) {
                // reference shapes: rf=1 -> (h, w); rf=2 -> cv2 dsize transposition -> (ceil(w/rf), ceil(h/rf))
                obs_shape[2 * item + 0] = (h.rf == 1) ? h.h : (h.w + h.rf - 1) / h.rf;
                obs_shape[2 * item + 1] = (h.rf == 1) ? h.w : (h.h + h.rf - 1) / h.rf;
            }
            if (status_out) status_out[item] = h.status;
            hs->m = 0;  // nothing to stream
        }
        prep_sync<WAVE>();
        return hs;
    }

    // ------------------------------------------------------------------ gather the state rows of the footprint
    int ht_ld = 0;
    if (MODE == IPP_FACTOR) {
        ht_ld = (r + 3) & ~3;
        const int SI = (si > 0) ? si : ht_ld, SK = (si > 0) ? sk : 1;
        if (span_s)
            for (int k = tid; k < r; k += kPrepThreads) span_s[k] = CHAIN ? cc->span(k) : span[k];
        if (gather_on) {
            const int k00 = tid / MP;
#pragma unroll
            for (int u = 0; u < UN; ++u)
                if (k00 + u * KS < r) big[gi * SI + (k00 + u * KS) * SK] = sacc0[u] * gw;
            for (int k0 = k00 + UN * KS; k0 < r; k0 += UN * KS) {  // further passes (rank > UN * KS)
                int sp[UN];
                unsigned rc[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int kc = min(k0 + u * KS, r - 1);
                    sp[u] = CHAIN ? cc->span(kc) : span[kc];
                    rc[u] = !rect_meta ? kRectFull : (CHAIN ? cc->rect(kc) : (unsigned)rects[kc]);
                }
                float sacc[UN];
                gather_rows(k0, sp, rc, sacc);
#pragma unroll
                for (int u = 0; u < UN; ++u)
                    if (k0 + u * KS < r) big[gi * SI + (k0 + u * KS) * SK] = sacc[u] * gw;
            }
        }
    } else {
        // PFF[a][b] = P[F_a][F_b]   (f x f)
        for (int a = tid / 32; a < f; a += kPrepThreads / 32)
            for (int b = tid & 31; b < f; b += 32) big[a * (FC + 1) + b] = cov_env[(size_t)cellidx[a] * v.Npad + cellidx[b]];
    }
    prep_sync<WAVE>();
    IPP_TICK(v, 3, tick);
    if ((FRONT_ONLY || WAVE) && threadIdx.x == 0) IPP_MARK(item, 5);
    if ((flags & IPP_UPDATE_PREV) && tid == 0) {
        // every thread has taken its copy of prev_action (batch 1) before the barrier above
        double* pw = const_cast<double*>(prev_action);
        pw[3 * item + 0] = ax; pw[3 * item + 1] = ay; pw[3 * item + 2] = az;
    }
    if (FRONT_ONLY) return hs;

    // ------------------------------------------------------------------ S = H P_FF H^T + R  (mappings.py:182-183)
    // one 8-lane group per (i <= j) pair: prior / P_FF part over the <= rf^4 cell combinations, factor part
    // -sum_k HT[i][k] HT[j][k] split over the 8 lanes, fp64 accumulation, 3 shuffle steps to reduce.
    {
        constexpr int GL = 8;
        const int grp = tid / GL, sl = tid & (GL - 1), n_grp = kPrepThreads / GL;
        const int npairs = m * (m + 1) / 2;
        for (int p0 = 0; p0 < npairs; p0 += n_grp) {
            const int p = p0 + grp;
            const bool on = p < npairs;
            int i = 0, j = 0;
            if (on) {
                j = (int)((sqrtf(8.0f * p + 1.0f) - 1.0f) * 0.5f);
                while (j * (j + 1) / 2 > p) --j;
                while ((j + 1) * (j + 2) / 2 <= p) ++j;
                i = p - j * (j + 1) / 2;  // i <= j
            }
            double acc = 0.0;
            if (on) {
                const int ci = bcnt[i], cj = bcnt[j];        // 1, 2 or 4
                const int sh = (cj == 4) ? 2 : (cj == 2 ? 1 : 0);
                const double wij = bwt[i] * bwt[j];
                for (int c = sl; c < ci * cj; c += GL) {
                    const int fa = bfi[4 * i + (c >> sh)], fb = bfi[4 * j + (c & (cj - 1))];
                    if (MODE == IPP_FACTOR) {
                        acc += wij * ktab[abs(bfi_y(fa) - bfi_y(fb)) * h.w + abs(bfi_x(fa) - bfi_x(fb))];
                    } else {
                        acc += wij * (double)big[bfi_flat(fa) * (FC + 1) + bfi_flat(fb)];
                    }
                }
                if (MODE == IPP_FACTOR) {
                    const int SI = (si > 0) ? si : ht_ld, SK = (si > 0) ? sk : 1;
                    const float* hi = big + i * SI;
                    const float* hj = big + j * SI;
                    for (int k = sl; k < r; k += GL) acc -= (double)hi[k * SK] * (double)hj[k * SK];
                }
            }
#pragma unroll
            for (int off = GL / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, GL);
            if (on && sl == 0) {
                if (i == j) acc += R;
                S[i * LD + j] = acc;
                S[j * LD + i] = acc;
            }
        }
    }
    prep_sync<WAVE>();
    IPP_TICK(v, 4, tick);

    // ------------------------------------------------------------------ Cholesky S = C C^T (C lower), fp64
    // np.linalg.cholesky(S) returns C; the reference uses L = C^T (upper).  mappings.py:185
    for (int c = 0; c < m; ++c) {
        if (tid == 0) {
            double d = S[c * LD + c];
            for (int k = 0; k < c; ++k) d -= L[c * LD + k] * L[c * LD + k];
            if (!(d > 0.0)) okflag[0] = 0;
            L[c * LD + c] = sqrt(d);
        }
        prep_sync<WAVE>();
        if (okflag[0] == 0) break;
        if (tid > c && tid < m) {
            double s = S[tid * LD + c];
            for (int k = 0; k < c; ++k) s -= L[tid * LD + k] * L[c * LD + k];
            L[tid * LD + c] = s / L[c * LD + c];
        }
        prep_sync<WAVE>();
    }
    const bool pd = okflag[0] != 0;
    int status = h.status;
    int fallback = 0;

    if (pd) {
        // L_inv = inv(C^T): column j by back substitution on the upper factor U = C^T. mappings.py:186
        if (tid < m) {
            const int j = tid;
            for (int i = 0; i < m; ++i) Li[i * LD + j] = 0.0;
            Li[j * LD + j] = 1.0 / L[j * LD + j];
            for (int i = j - 1; i >= 0; --i) {
                double s = 0.0;
                for (int k = i + 1; k <= j; ++k) s += L[k * LD + i] * Li[k * LD + j];  // U[i][k] = C[k][i]
                Li[i * LD + j] = -s / L[i * LD + i];
            }
        }
        prep_sync<WAVE>();
        if (tid < m) {  // y = L_inv^T v   (mappings.py:189,196: W v = Wc L^-T v)
            double s = 0.0;
            for (int i = 0; i <= tid; ++i) s += Li[i * LD + tid] * vv[i];
            yy[tid] = s;
        }
    } else if (MODE == IPP_DENSE) {
        // mappings.py:200-215: S_inv = inv(S) (Gauss-Jordan with partial pivoting, one thread: rare path)
        fallback = 1;
        status = IPP_STATUS_CHOL_FALLBACK;
        if (tid == 0) {
            for (int i = 0; i < m; ++i)
                for (int j = 0; j < m; ++j) { L[i * LD + j] = S[i * LD + j]; Li[i * LD + j] = (i == j) ? 1.0 : 0.0; }
            for (int c = 0; c < m; ++c) {
                int piv = c;
                double best = fabs(L[c * LD + c]);
                for (int i = c + 1; i < m; ++i)
                    if (fabs(L[i * LD + c]) > best) { best = fabs(L[i * LD + c]); piv = i; }
                if (piv != c)
                    for (int j = 0; j < m; ++j) {
                        double t = L[c * LD + j]; L[c * LD + j] = L[piv * LD + j]; L[piv * LD + j] = t;
                        t = Li[c * LD + j]; Li[c * LD + j] = Li[piv * LD + j]; Li[piv * LD + j] = t;
                    }
                const double inv = 1.0 / L[c * LD + c];
                for (int j = 0; j < m; ++j) { L[c * LD + j] *= inv; Li[c * LD + j] *= inv; }
                for (int i = 0; i < m; ++i)
                    if (i != c) {
                        const double fct = L[i * LD + c];
                        if (fct != 0.0)
                            for (int j = 0; j < m; ++j) { L[i * LD + j] -= fct * L[c * LD + j]; Li[i * LD + j] -= fct * Li[c * LD + j]; }
                    }
            }
        }
        prep_sync<WAVE>();
        if (tid < m) {  // y = S_inv v
            double s = 0.0;
            for (int i = 0; i < m; ++i) s += Li[tid * LD + i] * vv[i];
            yy[tid] = s;
        }
    } else {
        status = IPP_STATUS_NOT_PD;  // factor form cannot hold an indefinite update
    }
    prep_sync<WAVE>();
    IPP_TICK(v, 5, tick);

    if (WAVE && threadIdx.x == 0) IPP_MARK(item, 7);
    // ------------------------------------------------------------------ outputs for the streaming kernels
    const bool dead = (MODE == IPP_FACTOR) && !pd;
    for (int idx = tid; idx < MC * MC; idx += kPrepThreads) {
        const int i = idx / MC, j = idx - i * MC;
        const double val = (!dead && i < m && j < m) ? Li[i * LD + j] : 0.0;
        linv_f[idx] = (float)val;
        if (linv_f2) linv_f2[idx] = (float)val;
        dbg[MC * MC + idx] = val;
        dbg[idx] = (i < m && j < m) ? S[i * LD + j] : 0.0;
    }
    for (int i = tid; i < MC; i += kPrepThreads) {
        const double yval = (!dead && !cov_only && i < m) ? yy[i] : 0.0;
        y_f[i] = (float)yval;
        if (y_f2) y_f2[i] = (float)yval;
        dbg[2 * MC * MC + i] = (i < m) ? zz[i] : 0.0;
        dbg[2 * MC * MC + MC + i] = yval;
    }
    constexpr int QS = (MC + 3) & ~3;
    constexpr int QP = (QS <= 16) ? 16 : 32;
    if (MODE == IPP_FACTOR) {
        // Q[k][j] = -sum_{i<=j} HT[i][k] L_inv[i][j]:  Wc = P0[:,F] G - U (U[F,:]^T G), sign folded into Q
        const int j = tid & (QP - 1);
        if (j < QS) {
            for (int k = tid / QP; k < r; k += kPrepThreads / QP) {
                double sacc = 0.0;
                const int SI = (si > 0) ? si : ht_ld, SK = (si > 0) ? sk : 1;
                if (!dead && j < m)
                    for (int i = 0; i <= j; ++i) sacc += (double)big[i * SI + k * SK] * Li[i * LD + j];
                q_out[k * QS + j] = (float)(-sacc);  // (fused kernel: in place over HT row k, reads precede the write)
            }
        }
    } else {
        // Q[fi][j] = w_f L_inv[blk(f)][j]  (normal)   or  w_f [blk(f) == j]  (fallback: stream PH^T)
        const int j = tid & (QP - 1);
        if (j < QS) {
            for (int fi = tid / QP; fi < f; fi += kPrepThreads / QP) {
                const int ly = fi / h.w, lx = fi - ly * h.w;
                const int bi = (ly / h.rf) * h.nx + lx / h.rf;
                double sacc = 0.0;
                if (j < m) sacc = fallback ? ((bi == j) ? bwt[bi] : 0.0) : bwt[bi] * Li[bi * LD + j];
                q_out[fi * QS + j] = (float)sacc;
            }
        }
    }
    {   // the gain kernel's LDS-DMA copies whole rows past the end: keep the 8 rows after Q zero
        const int qrows = (MODE == IPP_FACTOR) ? r : f;
        for (int idx = tid; idx < 8 * QS; idx += kPrepThreads) q_out[(size_t)qrows * QS + idx] = 0.f;
    }
    if (tid == 0) {
        ItemHdr ho = h;
        ho.status = status;
        ho.fallback = fallback;
        if (dead) { ho.commit = 0; ho.rows = 0; }
        v.hdr[item] = ho;
        *hs = ho;
        if (status_out) status_out[item] = status;
    }
    IPP_TICK(v, 6, tick);
    return hs;
}

template <int MC, int MODE, int NT>
__device__ __forceinline__ ItemHdr* prepare_item(const View& v, const int item, const int* __restrict__ env_ids,
                                                 const int* __restrict__ dst_ids, const double* __restrict__ action,
                                                 const double* __restrict__ prev_action,
                                                 const float* __restrict__ meas_noise, unsigned flags,
                                                 int* __restrict__ status_out, float* __restrict__ obs_out,
                                                 int* __restrict__ obs_m, int* __restrict__ obs_shape,
                                                 unsigned char* small, float* big, int si, int sk, float* q_out,
                                                 float* linv_f, float* linv_f2, float* y_f, float* y_f2, int* span_s) {
    return prepare_item_ex<MC, MODE, NT, false>(v, item, env_ids, dst_ids, action, prev_action, meas_noise, flags, status_out,
                                                obs_out, obs_m, obs_shape, small, big, si, sk, q_out, linv_f, linv_f2, y_f,
                                                y_f2, span_s, NoMidWork());
}

// The m x m algebra of a factor-state item by ONE wave (no workgroup barrier): S = H P_FF H^T + R from the prior
// and the HT rows staged in LDS (HT(i,k) = ht[k*QS + i]), Cholesky, L^-1, y.  Called by wave 0 of the fused step
// kernel after prepare_item_ex<FRONT_ONLY> while the other waves already stream.  Writes L^-1 / y (fp32) for the
// tile epilogues, the debug copies, the final item header and status.  Returns the item status.
// mapping/mappings.py:178-197.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // LDS writes of this wave visible to its other lanes
    __builtin_amdgcn_wave_barrier();
}

template <int MC>
__device__ __forceinline__ int solve_wave(const View& v, const ItemHdr& h, const int item, unsigned flags,
                                          unsigned char* small, const float* ht, float* linv_f, float* y_f,
                                          int* __restrict__ status_out) {
    constexpr int LD = MC + 1;
    constexpr int QS = (MC + 3) & ~3;
    const PrepLds<MC> pl(small);
    double* S = pl.S; double* L = pl.L; double* Li = pl.Li; double* zz = pl.zz; double* vv = pl.vv; double* yy = pl.yy;
    const double* ktab = pl.ktab; const int* bfi = pl.bfi; const int* bcnt = pl.bcnt; const double* bwt = pl.bwt;
    const int lane = threadIdx.x & (kWave - 1);
    const int m = h.m, r = h.rank;
    const double R = (double)(h.rf * h.rf * h.rf) * h.nv_d;  // sensor_models.py:36
    const bool cov_only = (flags & IPP_COV_ONLY) != 0;
    double* dbg = v.dbg + (size_t)item * (2 * MC * MC + 2 * MC);

    {   // S: one 8-lane group per (i <= j) pair, fp64
        constexpr int GL = 8;
        const int grp = lane / GL, sl = lane & (GL - 1), n_grp = kWave / GL;
        const int npairs = m * (m + 1) / 2;
        for (int p0 = 0; p0 < npairs; p0 += n_grp) {
            const int p = p0 + grp;
            const bool on = p < npairs;
            int i = 0, j = 0;
            if (on) {
                j = (int)((sqrtf(8.0f * p + 1.0f) - 1.0f) * 0.5f);
                while (j * (j + 1) / 2 > p) --j;
                while ((j + 1) * (j + 2) / 2 <= p) ++j;
                i = p - j * (j + 1) / 2;  // i <= j
            }
            double acc = 0.0;
            if (on) {
                const int ci = bcnt[i], cj = bcnt[j];        // 1, 2 or 4
                const int sh = (cj == 4) ? 2 : (cj == 2 ? 1 : 0);
                const double wij = bwt[i] * bwt[j];
                for (int c = sl; c < ci * cj; c += GL) {
                    const int fa = bfi[4 * i + (c >> sh)], fb = bfi[4 * j + (c & (cj - 1))];
                    acc += wij * ktab[abs(bfi_y(fa) - bfi_y(fb)) * h.w + abs(bfi_x(fa) - bfi_x(fb))];
                }
                for (int k = sl; k < r; k += GL) acc -= (double)ht[k * QS + i] * (double)ht[k * QS + j];
            }
#pragma unroll
            for (int off = GL / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, GL);
            if (on && sl == 0) {
                if (i == j) acc += R;
                S[i * LD + j] = acc;
                S[j * LD + i] = acc;
            }
        }
    }
    wave_lds_sync();

    // Cholesky S = C C^T (C lower), fp64; the reference uses L = C^T (upper).  mappings.py:185
    bool pd = true;
    for (int c = 0; c < m; ++c) {
        double d = 0.0;
        if (lane == 0) {
            d = S[c * LD + c];
            for (int k = 0; k < c; ++k) d -= L[c * LD + k] * L[c * LD + k];
            L[c * LD + c] = sqrt(d);
        }
        d = __shfl(d, 0, kWave);
        if (!(d > 0.0)) { pd = false; break; }
        wave_lds_sync();
        if (lane > c && lane < m) {
            double sacc = S[lane * LD + c];
            for (int k = 0; k < c; ++k) sacc -= L[lane * LD + k] * L[c * LD + k];
            L[lane * LD + c] = sacc / L[c * LD + c];
        }
        wave_lds_sync();
    }
    int status = h.status;
    if (pd) {
        // L_inv = inv(C^T): column j by back substitution on the upper factor U = C^T. mappings.py:186
        if (lane < m) {
            const int j = lane;
            for (int i = 0; i < m; ++i) Li[i * LD + j] = 0.0;
            Li[j * LD + j] = 1.0 / L[j * LD + j];
            for (int i = j - 1; i >= 0; --i) {
                double sacc = 0.0;
                for (int k = i + 1; k <= j; ++k) sacc += L[k * LD + i] * Li[k * LD + j];  // U[i][k] = C[k][i]
                Li[i * LD + j] = -sacc / L[i * LD + i];
            }
        }
        wave_lds_sync();
        if (lane < m) {  // y = L_inv^T v   (mappings.py:189,196: W v = Wc L^-T v)
            double sacc = 0.0;
            for (int i = 0; i <= lane; ++i) sacc += Li[i * LD + lane] * vv[i];
            yy[lane] = sacc;
        }
        wave_lds_sync();
    } else {
        status = IPP_STATUS_NOT_PD;  // factor form cannot hold an indefinite update
    }
    const bool dead = !pd;
    for (int idx = lane; idx < MC * MC; idx += kWave) {
        const int i = idx / MC, j = idx - i * MC;
        const double val = (!dead && i < m && j < m) ? Li[i * LD + j] : 0.0;
        linv_f[idx] = (float)val;
        dbg[MC * MC + idx] = val;
        dbg[idx] = (i < m && j < m) ? S[i * LD + j] : 0.0;
    }
    for (int i = lane; i < MC; i += kWave) {
        const double yval = (!dead && !cov_only && i < m) ? yy[i] : 0.0;
        y_f[i] = (float)yval;
        dbg[2 * MC * MC + i] = (i < m) ? zz[i] : 0.0;
        dbg[2 * MC * MC + MC + i] = yval;
    }
    if (lane == 0) {
        ItemHdr ho = h;
        ho.status = status;
        ho.fallback = 0;
        if (dead) { ho.commit = 0; ho.rows = 0; }
        v.hdr[item] = ho;
        if (status_out) status_out[item] = status;
    }
    return status;
}

// ---------------------------------------------------------------------------------------------------------------
// The same m x m algebra by one wave with the matrices in REGISTERS (MC = 9; other caps use solve_wave): the LDS
// version above is a chain of ~40 dependent LDS round trips with a wave fence between them (14-17 us per item, measured
// in the fused kernels' timelines); here lane i holds row i of S / C and lane j column j of L^-1, wave-uniform values
// travel through v_readlane, and the factor part of S is summed with one lane per stored column k (no serial loop over
// the rank).  HT(i,k) = ht[i*si + k*sk].  Also writes Q = -HT^T L^-1 (fp32 rows [k][QS], then 8 zero rows) when q_out is
// not null: the consumers of the (deleted) persistent kernel streamed against Q, so their tile epilogue needs no L^-1.
// mapping/mappings.py:178-197.  Same outputs as solve_wave (L^-1 / y in fp32, debug copies in fp64, header, status).
// The observation of an item by ONE wave (DEFER mode of prepare_item_ex; the same arithmetic in the same order as the
// block-wide code there): z = clip(INTER_AREA(ground-truth crop) + nv * eps), innovation v = z - H x into the prologue's
// LDS scratch (zz, vv).  simulations/simulations.py:26-34, sensor_manipulations.py:7-57, mappings.py:195.
// The caller publishes the result to the solving wave (solve_wave_fast, obs_flag).
template <int MC>
__device__ __forceinline__ void observe_wave(const View& v, const ItemHdr& h, unsigned flags, unsigned char* small, const ObsRegs o) {
    const PrepLds<MC> pl(small);
    double* L = pl.L; double* zz = pl.zz; double* vv = pl.vv; double* sub = pl.sub;
    const int lane = threadIdx.x & (kWave - 1);
    const int m = h.m, f = h.f;
    const bool cov_only = (flags & IPP_COV_ONLY) != 0;
    if (cov_only) {
        if (lane < MC) { zz[lane] = 0.0; vv[lane] = 0.0; }
        wave_lds_sync();
        return;
    }
    float gt_f = o.gt;
    asm volatile("" : "+v"(gt_f));
    if (lane < f) sub[lane] = (double)gt_f;
    double* wyt = L;           // [orows][h.h]   (the L / Li scratch is not used by solve_wave_fast)
    double* wxt = L + 2 * MC;  // [ocols][h.w]
    if (h.rf > 1) {
        const int ocols = (h.h + h.rf - 1) / h.rf, orows = (h.w + h.rf - 1) / h.rf;
        for (int idx = lane; idx < orows * h.h; idx += kWave) wyt[idx] = area_weight(h.h, orows, idx / h.h, idx % h.h);
        for (int idx = lane; idx < ocols * h.w; idx += kWave) wxt[idx] = area_weight(h.w, ocols, idx / h.w, idx % h.w);
    }
    wave_lds_sync();
    if (lane < m) {
        const Block myb = block_of(lane, h.nx, h.rf, h.w, h.h);
        double val;
        if (h.rf == 1) {
            val = sub[lane];
        } else {
            const int ocols = (h.h + h.rf - 1) / h.rf;
            const int orow = lane / ocols, ocol = lane - orow * ocols;
            val = 0.0;
            for (int sy = 0; sy < h.h; ++sy) {
                const double wy = wyt[orow * h.h + sy];
                if (wy == 0.0) continue;
                for (int sx = 0; sx < h.w; ++sx) {
                    const double wx = wxt[ocol * h.w + sx];
                    if (wx != 0.0) val += sub[sy * h.w + sx] * wx * wy;
                }
            }
        }
        float eps_f = o.eps;
        asm volatile("" : "+v"(eps_f));
        const double eps = (double)eps_f;
        if (flags & IPP_GIVEN_OBSERVATION)
            val = eps;
        else
            val = fmin(fmax(val + h.nv_d * eps, 0.0), 1.0);
        zz[lane] = val;
        double hx = 0.0;
        // (fixed trip count: with the block's cell count as the bound, o.mean[] was indexed dynamically and ObsRegs lived in scratch --
        // the only scratch object of the patch kernels; same additions in the same order)
        const int cnt_b = myb.count();
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < cnt_b) hx += myb.weight * (double)o.mean[a];
        vv[lane] = val - hx;
    } else if (lane < MC) {
        zz[lane] = 0.0;
        vv[lane] = 0.0;
    }
    wave_lds_sync();
}


template <int MC>
__device__ __forceinline__ int solve_wave_fast(const View& v, const ItemHdr& h, const int item, unsigned flags,
                                               unsigned char* small, const float* ht, int si, int sk, float* linv_f,
                                               float* y_f, float* __restrict__ q_out, int* __restrict__ status_out,
                                               int* obs_flag = nullptr, int r_ht = -1, const float* ht2 = nullptr, int r_ht2 = 0) {
    // r_ht >= 0: `ht` holds that many rows instead of h.rank (k_step_patch.h: only the columns that reach the footprint are
    // staged); ht2 / r_ht2: further rows with the same strides in a second array (rows that did not fit the LDS staging)
    static_assert(MC == 9, "register layout written for MC = 9");
    constexpr int LD = MC + 1;
    constexpr int QS = (MC + 3) & ~3;
    const PrepLds<MC> pl(small);
    double* S = pl.S; double* zz = pl.zz; double* vv = pl.vv;
    const double* ktab = pl.ktab; const int* bfi = pl.bfi; const int* bcnt = pl.bcnt; const double* bwt = pl.bwt;
    const int lane = threadIdx.x & (kWave - 1);
    const int m = h.m, r = r_ht >= 0 ? r_ht : h.rank;
    const double R = (double)(h.rf * h.rf * h.rf) * h.nv_d;  // sensor_models.py:36
    const bool cov_only = (flags & IPP_COV_ONLY) != 0;
    double* dbg = v.dbg + (size_t)item * (2 * MC * MC + 2 * MC);

    // ---- S, prior part + R: lane p owns pair p = (i <= j), p = j (j + 1) / 2 + i  (45 pairs <= 64 lanes)
    int pi = 0, pj = 0;
    {
        int j = (int)((sqrtf(8.0f * lane + 1.0f) - 1.0f) * 0.5f);
        while (j * (j + 1) / 2 > lane) --j;
        while ((j + 1) * (j + 2) / 2 <= lane) ++j;
        pj = j;
        pi = lane - j * (j + 1) / 2;
    }
    const bool pair_on = pj < m;  // (pi <= pj)
    double mine = 0.0;
    if (pair_on) {
        const int ci = bcnt[pi], cj = bcnt[pj];  // 1, 2 or 4
        const int sh = (cj == 4) ? 2 : (cj == 2 ? 1 : 0);
        const double wij = bwt[pi] * bwt[pj];
        for (int c = 0; c < ci * cj; ++c) {
            const int fa = bfi[4 * pi + (c >> sh)], fb = bfi[4 * pj + (c & (cj - 1))];
            mine += wij * ktab[abs(bfi_y(fa) - bfi_y(fb)) * h.w + abs(bfi_x(fa) - bfi_x(fb))];
        }
        if (pi == pj) mine += R;
    }
    // ---- factor part: - sum_k HT[i][k] HT[j][k] by the pair's own lane, all pairs in lock step (two LDS reads per k,
    // eight k in flight; no cross-lane reduction: 45 wave-wide fp64 butterflies through ds_bpermute took 25 us);
    // fp64 products of fp32 values are exact
    // (order of the sum: groups of eight columns on four accumulators, then the remainder on the first -- the same for a column
    // wherever it is staged, so S does not depend on how many records fit the LDS of the kernel that runs the solve: the staging
    // capacities are multiples of eight, ht2 continues the groups of ht)
    if (pair_on) {
        const float* hi = ht + pi * si;
        const float* hj = ht + pj * si;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        auto groups = [&](const float* xi, const float* xj, int n) {
            int k = 0;
            for (; k + 8 <= n; k += 8) {
                float x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { x[u] = xi[(k + u) * sk]; y[u] = xj[(k + u) * sk]; }
                a0 = fma((double)x[0], (double)y[0], a0); a1 = fma((double)x[1], (double)y[1], a1);
                a2 = fma((double)x[2], (double)y[2], a2); a3 = fma((double)x[3], (double)y[3], a3);
                a0 = fma((double)x[4], (double)y[4], a0); a1 = fma((double)x[5], (double)y[5], a1);
                a2 = fma((double)x[6], (double)y[6], a2); a3 = fma((double)x[7], (double)y[7], a3);
            }
            for (; k < n; ++k) a0 = fma((double)xi[k * sk], (double)xj[k * sk], a0);
        };
        groups(hi, hj, r);
        if (ht2) groups(ht2 + pi * si, ht2 + pj * si, r_ht2);  // (r is then a multiple of eight: no remainder in front of these)
        mine -= (a0 + a1) + (a2 + a3);
    }
    if (pair_on) {
        S[pi * LD + pj] = mine;
        S[pj * LD + pi] = mine;
    }
    wave_lds_sync();

    // ---- Cholesky S = C C^T in registers: lane i holds row i (c[k] = C[i][k]); the reference uses L = C^T.  mappings.py:185
    double c[MC];
#pragma unroll
    for (int k = 0; k < MC; ++k) c[k] = (lane < m && k < m) ? S[min(lane, MC - 1) * LD + k] : ((k == lane) ? 1.0 : 0.0);
    const bool capture = v.dbg_capture == 1;  // (1.4 KB of fp64 stores per item: 6 MB per launch of the headline batch -- only on request)
    if (capture && lane < MC) {  // debug copy of S (tests read it through ipp_debug_step_item)
#pragma unroll
        for (int k = 0; k < MC; ++k) dbg[lane * MC + k] = (lane < m && k < m) ? c[k] : 0.0;
    }
    bool pd = true;
    double rd[MC];  // 1 / C[j][j] (wave-uniform), shared by the column scaling here and by L^-1 below
#pragma unroll
    for (int j = 0; j < MC; ++j) {
        rd[j] = 1.0;
        if (j < m) {  // wave-uniform
            double t = c[j];
#pragma unroll
            for (int k = 0; k < j; ++k) t = fma(-c[k], bcast_lane(c[k], j), t);  // - C[i][k] C[j][k]
            const double d = bcast_lane(t, j);
            if (!(d > 0.0)) pd = false;
            // 1 / sqrt(d): hardware estimate + two Newton steps (to an ulp or two in fp64; the 27 fp64 divisions and 9
            // square roots of the textbook form were ~5 % of the item's vector instructions)
            double y = __builtin_amdgcn_rsq(d);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const double e = fma(-(d * y), y, 1.0);
                y = fma(0.5 * y, e, y);
            }
            rd[j] = y;
            c[j] = (lane == j) ? d * y : t * y;  // rows above the diagonal hold junk that is never read
        }
    }
    int status = h.status;
    // ---- L^-1 = inv(C^T) (upper triangular): lane j holds column j, li[i] = Linv[i][j].  mappings.py:186
    double li[MC];
#pragma unroll
    for (int i = 0; i < MC; ++i) li[i] = 0.0;
    if (pd) {
#pragma unroll
        for (int i = MC - 1; i >= 0; --i) {
            if (i < m) {
                double sacc = 0.0;
#pragma unroll
                for (int k = i + 1; k < MC; ++k) {
                    if (k < m) {
                        const double cki = bcast_lane(c[i], k);  // C[k][i] = U[i][k]
                        if (k <= lane) sacc = fma(cki, li[k], sacc);
                    }
                }
                li[i] = (lane == i) ? rd[i] : ((lane > i && lane < m) ? -sacc * rd[i] : 0.0);
            }
        }
    } else {
        status = IPP_STATUS_NOT_PD;  // factor form cannot hold an indefinite update
    }
    const bool dead = !pd;
    if (obs_flag) {  // the innovation (and z) come from the observing wave (observe_wave)
        while (__hip_atomic_load(obs_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) __builtin_amdgcn_s_sleep(2);
    }
    // y = L_inv^T v   (mappings.py:189,196: W v = Wc L^-T v)
    double yv = 0.0;
    if (pd && lane < m) {
#pragma unroll
        for (int i = 0; i < MC; ++i)
            if (i <= lane) yv = fma(li[i], vv[i], yv);
    }
    if (lane < MC) {
#pragma unroll
        for (int i = 0; i < MC; ++i) {
            const double val = (!dead && i < m && lane < m) ? li[i] : 0.0;
            linv_f[i * MC + lane] = (float)val;
            if (capture) dbg[MC * MC + i * MC + lane] = val;
        }
        const double yval = (!dead && !cov_only && lane < m) ? yv : 0.0;
        y_f[lane] = (float)yval;
        if (capture) {
            dbg[2 * MC * MC + lane] = (lane < m) ? zz[lane] : 0.0;
            dbg[2 * MC * MC + MC + lane] = yval;
        }
    }
    if (lane == 0) {
        ItemHdr ho = h;
        ho.status = status;
        ho.fallback = 0;
        if (dead) { ho.commit = 0; ho.rows = 0; }
        v.hdr[item] = ho;
        *pl.hs = ho;
        if (status_out) status_out[item] = status;
    }
    wave_lds_sync();
    if (q_out) {
        // Q[k][j] = -sum_{i<=j} HT[i][k] Linv[i][j]: lane <-> k, L^-1 (fp32) broadcast from LDS
        for (int k0 = 0; k0 < r; k0 += kWave) {
            const int k = k0 + lane;
            if (k < r) {
                float hv[MC], q[QS];
#pragma unroll
                for (int i = 0; i < MC; ++i) hv[i] = (i < m) ? ht[i * si + k * sk] : 0.f;
#pragma unroll
                for (int j = 0; j < QS; ++j) {
                    float acc = 0.f;
                    if (j < MC) {
#pragma unroll
                        for (int i = 0; i <= j; ++i) acc = fmaf(hv[i], linv_f[i * MC + j], acc);
                    }
                    q[j] = -acc;
                }
                float4* dst = reinterpret_cast<float4*>(q_out + (size_t)k * QS);
#pragma unroll
                for (int j4 = 0; j4 < QS / 4; ++j4) dst[j4] = make_float4(q[4 * j4], q[4 * j4 + 1], q[4 * j4 + 2], q[4 * j4 + 3]);
            }
        }
        for (int idx = lane; idx < 8 * QS; idx += kWave) q_out[(size_t)r * QS + idx] = 0.f;  // zero rows behind Q (pipeline tail)
    }
    return status;
}

// Stand-alone prologue kernel: one workgroup per item.
template <int MC, int MODE>
__global__ __launch_bounds__(kPrepThreads, IPP_PREP_MINWAVES) void k_prepare(View v, const int* __restrict__ env_ids,
                                                          const int* __restrict__ dst_ids, int n_items,
                                                          const double* __restrict__ action,
                                                          const double* __restrict__ prev_action,
                                                          const float* __restrict__ meas_noise, unsigned flags,
                                                          int* __restrict__ status_out, float* __restrict__ obs_out,
                                                          int* __restrict__ obs_m, int* __restrict__ obs_shape) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if ((int)blockIdx.x >= n_items) return;
    const int item = launch_item(v, blockIdx.x, n_items);
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    float* blk_out = v.q + (size_t)item * v.q_item;  // [L^-1 | y | pad | Q rows | zero rows]: one block for the gain kernel
    float* big = reinterpret_cast<float*>(smem + ((prep_small_bytes<MC>() + 15) & ~(size_t)15));
    if constexpr (MC == 9 && MODE == IPP_FACTOR) {
        if (!obs_out) {
            // factor state: the workgroup stops after the gather, wave 0 finishes the m x m algebra in registers and writes
            // Q (solve_wave_fast): the block-wide LDS Cholesky took ~15 us of a ~45 us item
            ItemHdr* hs = prepare_item_ex<MC, MODE, kPrepThreads, true>(v, item, env_ids, dst_ids, action, prev_action, meas_noise, flags,
                                                                        status_out, nullptr, nullptr, nullptr, smem, big, 0, 1, nullptr,
                                                                        v.linv + (size_t)item * MC * MC, blk_out, v.yv + (size_t)item * MC,
                                                                        blk_out + MC * MC, nullptr, NoMidWork());
            if (threadIdx.x >= kWave) return;
            const ItemHdr h = uniform_hdr(*hs);
            if (h.m == 0) return;
            const PrepLds<MC> pl(smem);
            float* Ls = reinterpret_cast<float*>(pl.L);  // (the L scratch is free: 90 doubles >= 81 + 9 floats)
            float* ys = Ls + MC * MC;
            solve_wave_fast<MC>(v, h, item, flags, smem, big, (h.rank + 3) & ~3, 1, Ls, ys, blk_out + LQ, status_out);
            wave_lds_sync();
            const int lane = threadIdx.x;
            for (int i = lane; i < MC * MC; i += kWave) { const float x = Ls[i]; v.linv[(size_t)item * MC * MC + i] = x; blk_out[i] = x; }
            if (lane < MC) { const float x = ys[lane]; v.yv[(size_t)item * MC + lane] = x; blk_out[MC * MC + lane] = x; }
            return;
        }
    }
    prepare_item<MC, MODE, kPrepThreads>(v, item, env_ids, dst_ids, action, prev_action, meas_noise, flags, status_out,
                                         obs_out, obs_m, obs_shape, smem, big, 0, 1, blk_out + LQ,
                                         v.linv + (size_t)item * MC * MC, blk_out, v.yv + (size_t)item * MC,
                                         blk_out + MC * MC, nullptr);
}

}  // namespace ipp
