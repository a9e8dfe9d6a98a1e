// Factor-state gain kernel, ONE WAVE per step item (windowed factor state).
//
// Why one wave: the per-item LDS footprint decides how many items a CU keeps in flight.  A workgroup that stages
// the item's Q block (rank_cap x 48 B) and a full prior table in LDS fits 4-5 times per CU; here Q stays in the
// prologue's global scratch block and reaches the FMAs through the SCALAR cache (one s_load_dwordx4 x3 per
// streamed row, wave-uniform, no VGPRs), and the prior table is built per tile for just the |drow| range the
// tile needs.  LDS drops to ~7 KB per item, so all 16 waves a CU can hold at 128 VGPRs are 16 different items:
// no intra-workgroup imbalance, no workgroup barriers, and 4x more independent row streams per CU.
// (tools/probes/stream_probe.hip, "wave-per-item scalarQ": 5.8 TB/s on the same access pattern.)
//
// Per tile (64 x VEC cells) the wave
//   * compacts (ballot / popcount, increasing k) the columns of U stored on the tile (ipp_config.window_rows),
//   * evaluates the prior term Wc0 = P0[:,F] H_F^T L^-1 from the tile's prior table,
//   * streams the stored rows (1 KiB per instruction, non-temporal) against Q rows from SGPRs,
//   * runs the fused epilogue (masked trace reduction, diag, mean, append the m new rows of U).
// mapping/mappings.py:188-197, planning/common/rewards.py:8-31.
#pragma once
#include "ipp_common.h"
#include "k_gain.h"

#ifndef IPP_GW_PIPE
#define IPP_GW_PIPE 12  // rows of U requested per group (A/B on MI355X: 4: 0.46 ms, 8: 0.42, 12: 0.40; ping-pong 2 x 4: 0.44)
#endif

namespace ipp {

constexpr int kTileLut = 1024;  // floats of per-tile prior table (|drow| range x W); larger needs fall back to sqrt/exp

template <int MC, int VEC>
__global__ __launch_bounds__(kWave, 4) void k_gain_wave(View v, const float* __restrict__ q_all, int n_items,
                                                        unsigned flags, float* __restrict__ reward_out) {
    constexpr int kWaveTile = VEC * kWave;
    constexpr int KP = IPP_GW_PIPE;
    constexpr int QS = (MC + 3) & ~3;
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gw[];
    // LDS carve: Ls[LQ] (L^-1 | y) | lut[kTileLut] | span[rank_cap] i32 | block tables | ridx[rank_cap + 8] u16
    float* Ls = reinterpret_cast<float*>(smem_gw);
    float* ys = Ls + MC * MC;
    float* lut = Ls + LQ;
    int* span_s = reinterpret_cast<int*>(lut + kTileLut);
    int* fb_yx = span_s + v.rank_cap;                                   // [MC][4] footprint cell (row << 16 | col) of block b
    float* fb_w = reinterpret_cast<float*>(fb_yx + 4 * MC);               // [MC][4] weight of that cell (0 for padding)
    unsigned short* ridx = reinterpret_cast<unsigned short*>(fb_w + 4 * MC);

    if ((int)blockIdx.x >= n_items) return;
    const int item = xcd_item(blockIdx.x, n_items);
    const int lane = threadIdx.x;
    __builtin_amdgcn_s_dcache_inv();  // (Q through the non-coherent scalar cache: nothing of an earlier launch may be served)
    const ItemHdr h = uniform_hdr(v.hdr[item]);
    const int m = h.m, r = h.rank;
    if (m == 0 || h.status == IPP_STATUS_NOT_PD) {
        if (lane == 0) reward_out[item] = (h.status == IPP_STATUS_NOT_PD) ? NAN : 0.f;
        return;
    }

    const float* __restrict__ blk = q_all + (size_t)item * v.q_item;  // [L^-1 | y | pad | Q rows | zero rows]
    const float* __restrict__ qrows = blk + LQ;
    for (int i = lane; i < LQ; i += kWave) Ls[i] = blk[i];
    for (int k = lane; k < r; k += kWave) span_s[k] = v.colspan[(size_t)h.env * v.rank_cap + k];
    if (lane < MC) {  // measurement blocks of the footprint (sensors/models/sensor_models.py:57-79), once per item
        const Block bb = block_of(min(lane, m - 1), h.nx, h.rf, h.w, h.h);
        for (int a = 0; a < 4; ++a) {
            const int aa = min(a, bb.count() - 1);
            fb_yx[4 * lane + a] = ((h.yu + bb.y0 + aa / bb.bw) << 16) | (h.xl + bb.x0 + aa % bb.bw);
            fb_w[4 * lane + a] = (lane < m && a < bb.count()) ? (float)bb.weight : 0.f;
        }
    }
    __syncthreads();

    const float* cov_src = v.cov + (size_t)h.env * v.cov_slot;
    float* cov_dst = v.cov + (size_t)h.dst * v.cov_slot;
    const bool adaptive = (flags & IPP_ADAPTIVE) != 0;
    const size_t npad = (size_t)v.Npad;
    const float s3 = (float)(kSqrt3 * v.res) / h.ls;
    double item_part = 0.0;
    unsigned long long units = 0;

    for (int tile = h.t_lo; tile <= h.t_hi; ++tile) {
        const int cell0 = tile * kWaveTile + VEC * lane;
        float mean_in[VEC], diag_in[VEC];  // requested now, consumed in the tile's epilogue
        load_vec<VEC>(v.mean + (size_t)h.env * npad + cell0, mean_in);
        load_vec<VEC>(v.diag + (size_t)h.env * npad + cell0, diag_in);

        // ---- ordered compaction of the columns stored on this tile
        int nact = 0;
        for (int k0 = 0; k0 < r; k0 += kWave) {
            const int k = k0 + lane;
            bool on = false;
            if (k < r) {
                const int sp = span_s[k];
                on = tile >= (sp & 0xffff) && tile <= (sp >> 16);
            }
            const unsigned long long mask = __ballot(on);
            if (on) ridx[nact + __popcll(mask & ((1ull << lane) - 1ull))] = (unsigned short)k;
            nact += __popcll(mask);
        }
        if (lane < 8) ridx[nact + lane] = (unsigned short)r;  // pipeline tail: the zero Q row, any valid U row

        // ---- prior table for this tile: P0(|drow|, |dcol|) for the |drow| values between tile rows and footprint rows
        const int trow0 = (tile * kWaveTile) / v.W, trow1 = min(tile * kWaveTile + kWaveTile - 1, v.N - 1) / v.W;
        const int dmin = max(0, max(trow0 - h.yd, h.yu - trow1));
        const int dmax = max(abs(trow0 - h.yu), max(abs(trow0 - h.yd), max(abs(trow1 - h.yu), abs(trow1 - h.yd))));
        const int nlut = (dmax - dmin + 1) * v.W;
        const bool use_lut = nlut <= kTileLut;
        if (use_lut) {
            for (int i = lane; i < nlut; i += kWave) {
                const int dr = i / v.W, dc = i - dr * v.W;
                lut[i] = matern_f(dmin + dr, dc, s3, h.sv);
            }
        }
        __syncthreads();  // single-wave workgroup: orders the LDS writes above before the reads below

        // ---- base term from the analytic prior: Wc0[i,:] = sum_b (sum_{f in block b} w_f P0[i, F_f]) L_inv[b,:]
        float acc[VEC][MC];
#pragma unroll
        for (int c = 0; c < VEC; ++c)
#pragma unroll
            for (int j = 0; j < MC; ++j) acc[c][j] = 0.f;
        {
            int crow[VEC], ccol[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                const int cell = min(cell0 + c, v.N - 1);
                crow[c] = cell / v.W;
                ccol[c] = cell - crow[c] * v.W;
            }
            // per block: 4 (padded, weight 0) footprint cells x VEC grid cells = 4*VEC independent table lookups in
            // flight, so the LDS latency is paid once per block instead of once per lookup
            for (int b = 0; b < m; ++b) {
                float cb[VEC];
#pragma unroll
                for (int c = 0; c < VEC; ++c) cb[c] = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int yx = fb_yx[4 * b + a];
                    const float wa = fb_w[4 * b + a];
                    const int fy = yx >> 16, fx = yx & 0xffff;
#pragma unroll
                    for (int c = 0; c < VEC; ++c) {
                        const int dr = abs(crow[c] - fy), dc = abs(ccol[c] - fx);
                        const float p0 = use_lut ? lut[(dr - dmin) * v.W + dc] : matern_f(dr, dc, s3, h.sv);
                        cb[c] = fmaf(wa, p0, cb[c]);
                    }
                }
#pragma unroll
                for (int j = 0; j < MC; ++j) {
                    const float l = Ls[b * MC + j];
#pragma unroll
                    for (int c = 0; c < VEC; ++c) acc[c][j] = fmaf(cb[c], l, acc[c][j]);
                }
            }
        }

        // ---- stream the stored rows: acc += row_k[cells] * Q[k,:]; Q row k is wave-uniform -> scalar loads
        if (nact > 0) {
            auto col_of = [&](int a) -> int { return __builtin_amdgcn_readfirstlane((int)ridx[min(a, nact + 7)]); };
            const int safe_k = col_of(0);  // nact > 0: the first column stored on this tile
            // the row registers stay VEC-wide vector values so that the loop-carried group is one register tuple per
            // row (scalarised, the back edge needs copies, and every copy waits for its load)
            typedef float rowv __attribute__((ext_vector_type(VEC)));
            auto fetch = [&](int k) -> rowv {
                // padding entries (k == r, zero Q row) read a column that IS stored on this tile: a column that is not
                // stored here holds uninitialised memory, and NaN * 0 would poison the accumulators
                return __builtin_nontemporal_load(reinterpret_cast<const rowv*>(cov_src + (size_t)(k < r ? k : safe_k) * npad + cell0));
            };
            auto consume = [&](const rowv (&u)[KP], const int (&kq)[KP]) {
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const float* __restrict__ qr = qrows + (size_t)kq[i] * QS;
                    float qv[MC];
#pragma unroll
                    for (int j = 0; j < MC; ++j) qv[j] = qr[j];
#pragma unroll
                    for (int j = 0; j < MC; ++j)
#pragma unroll
                        for (int c = 0; c < VEC; ++c) acc[c][j] = fmaf(u[i][c], qv[j], acc[c][j]);
                }
            };
            // Groups of KP rows, requested together and then consumed in order.  No row registers are carried
            // across the back edge: hipcc turns a carried (ping-pong) group into register copies at the loop end,
            // and each copy waits for its load, which empties the memory pipe once per iteration.  Overlap across
            // groups comes from the other 15 waves of the CU.
            for (int a = 0; a < nact; a += KP) {
                rowv u[KP];
                int kk[KP];
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    kk[i] = col_of(a + i);
                    u[i] = fetch(kk[i]);
                }
                __builtin_amdgcn_sched_barrier(0);  // all KP requests leave before the first wait
                consume(u, kk);
            }
        }

        // ---- epilogue for this tile
        float dred[VEC], dmean[VEC];
        double part = 0.0;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const bool valid = (cell0 + c) < v.N;
            float w2 = 0.f, dm = 0.f;
#pragma unroll
            for (int j = 0; j < MC; ++j) w2 = fmaf(acc[c][j], acc[c][j], w2);
#pragma unroll
            for (int j = 0; j < MC; ++j) dm = fmaf(acc[c][j], ys[j], dm);
            if (!valid) {
                w2 = 0.f; dm = 0.f;
#pragma unroll
                for (int j = 0; j < MC; ++j) acc[c][j] = 0.f;
            }
            dred[c] = w2;
            dmean[c] = dm;
            // rewards.py:11 mask from the pre-step mean and pre-step diag(P); rewards.py:23-30 trace reduction
            const bool in_mask = !adaptive || ((double)mean_in[c] + v.kf * (double)diag_in[c] >= v.thr);
            if (valid && in_mask) part += (double)w2;
        }
        item_part += wave_sum(part);
        const int valid_cells = max(0, min(kWaveTile, v.N - tile * kWaveTile));
        units += (unsigned long long)(nact + (h.commit ? m + 4 : 2)) * valid_cells;
        if (h.commit) {
            float outv[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) outv[c] = diag_in[c] - dred[c];
            store_vec<VEC>(v.diag + (size_t)h.dst * npad + cell0, outv);
            if (!(flags & IPP_COV_ONLY)) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) outv[c] = mean_in[c] + dmean[c];
                store_vec<VEC>(v.mean + (size_t)h.dst * npad + cell0, outv);
            }
#pragma unroll
            for (int j = 0; j < MC; ++j)
                if (j < m) {
#pragma unroll
                    for (int c = 0; c < VEC; ++c) outv[c] = acc[c][j];
                    store_stream<VEC>(cov_dst + (size_t)(r + j) * npad + cell0, outv);
                }
        }
        __syncthreads();  // the next tile rewrites ridx / lut
    }

    // ------------------------------------------------------------------ per-item results (tiles were summed in order)
    if (lane == 0) {
        reward_out[item] = (float)(item_part / (h.cost_d + 1.0));  // rewards.py:31
        if (h.commit) v.rank[h.dst] = r + m;
        if (units) atomicAdd(v.counters + (size_t)(item & (kCountSlots - 1)) * 16, units);
    }
    if (h.commit && lane < m) {
        v.colspan[(size_t)h.dst * v.rank_cap + r + lane] = h.t_lo | (h.t_hi << 16);
        v.colrect[(size_t)h.dst * v.rank_cap + r + lane] = (int)kRectFull;  // (band tiles: the whole span is written)
    }
}

}  // namespace ipp
