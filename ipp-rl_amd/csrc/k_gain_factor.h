// Factor-state gain kernel: ONE workgroup per step item, wave-granular cell tiles.
//
//   Wc = P0[:,F] H_F^T L^-1  -  U (U[F,:]^T H_F^T L^-1)          (mapping/mappings.py:188, P = P0 - U U^T)
//
// The workgroup stages the item's scratch block ([L^-1 | y | Q], written by k_prepare) and the prior table
// P0(|drow|, |dcol|) in LDS once; then every wave walks over the (64 x VEC)-cell tiles assigned to it.  Per tile a wave
//   * compacts (ballot / popcount, in increasing k) the columns of U that are stored on the tile -- all of them
//     when window_rows == 0, else those whose footprint lies within window_rows grid rows (ipp_config),
//   * evaluates the prior term from the table, streams the stored rows (1 KiB per wave instruction,
//     non-temporal, 2 x KP deep ping-pong) against Q broadcast from LDS,
//   * runs the fused epilogue: masked trace reduction (reward), diag -= |Wc_i|^2, mean += Wc y, append the m new
//     rows of U on the tile (planning/common/rewards.py:8-31, mapping/mappings.py:190-197).
// Tiles are handed out to waves dynamically; there is no workgroup barrier inside the tile loop; tiles outside the span of the appended columns are skipped.
// HBM-bound: 4*MC FMAs per streamed float4; the prologue is paid once per item, not once per tile.
#pragma once
#include "ipp_common.h"
#include "k_gain.h"

#ifndef IPP_GF_PIPE
#define IPP_GF_PIPE 2
#endif
#ifndef IPP_GF_MINWAVES
#define IPP_GF_MINWAVES 4
#endif

namespace ipp {

// LDS layout shared by k_gain_factor and k_step_factor.
template <int MC>
struct GainLds {
    static constexpr int QS = (MC + 3) & ~3;
    static constexpr int LQ = (MC * MC + MC + 3) & ~3;
    float* Ls; float* ys; float* Qs; float* lut; double* red; int* next_tile; int* done_waves; int* span_s;
    unsigned short* ridx_all;
    __device__ __forceinline__ GainLds(unsigned char* base, int rank_cap, int lut_floats) {
        Ls = reinterpret_cast<float*>(base);
        ys = Ls + MC * MC;
        Qs = Ls + LQ;
        lut = Qs + (size_t)(rank_cap + 8) * QS;
        red = reinterpret_cast<double*>(lut + ((lut_floats + 3) & ~3));
        next_tile = reinterpret_cast<int*>(red + 15);  // red[15] is unused by the reduction
        done_waves = next_tile + 1;
        span_s = reinterpret_cast<int*>(red + 16);
        ridx_all = reinterpret_cast<unsigned short*>(span_s + rank_cap);
    }
};

// Tile loop + per-item results; expects Ls / ys / Qs (rows 0..r-1 and a zero row r), the prior table (when
// use_lut), span_s[0..r) and the two counters (zeroed) in LDS, visible to the whole workgroup.
template <int MC, int VEC>
__device__ __forceinline__ void gain_tiles(const View& v, const ItemHdr& h, const int item, unsigned flags, bool use_lut,
                                           const GainLds<MC>& lds, float* __restrict__ reward_out) {
    constexpr int kWaveTile = VEC * kWave;  // cells per wave tile
    constexpr int KP = IPP_GF_PIPE;          // rows per ping-pong group (2 groups in flight per wave)
    constexpr int QS = (MC + 3) & ~3;
    float* Ls = lds.Ls; float* ys = lds.ys; float* Qs = lds.Qs; float* lut = lds.lut; double* red = lds.red;
    int* next_tile = lds.next_tile; int* done_waves = lds.done_waves; int* span_s = lds.span_s;
    unsigned short* ridx_all = lds.ridx_all;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave, nw = T / kWave;
    const int m = h.m, r = h.rank;
    const float s3 = (float)(kSqrt3 * v.res) / h.ls;

    const float* cov_src = v.cov + (size_t)h.env * v.cov_slot;
    float* cov_dst = v.cov + (size_t)h.dst * v.cov_slot;
    unsigned short* ridx = ridx_all + (size_t)wave * (v.rank_cap + 8);
    const bool adaptive = (flags & IPP_ADAPTIVE) != 0;
    const size_t npad = (size_t)v.Npad;
    double wave_part = 0.0;
    unsigned long long units = 0;

    // tiles are handed out dynamically (LDS counter): a wave that finishes a short tile takes the next one
    for (;;) {
        int tile = 0;
        if (lane == 0) tile = h.t_lo + atomicAdd(next_tile, 1);
        tile = __builtin_amdgcn_readfirstlane(tile);
        if (tile > h.t_hi) break;
        const int cell0 = tile * kWaveTile + VEC * lane;
        float mean_in[VEC], diag_in[VEC];  // requested now, consumed in the tile's epilogue
        load_vec<VEC>(v.mean + (size_t)h.env * npad + cell0, mean_in);
        load_vec<VEC>(v.diag + (size_t)h.env * npad + cell0, diag_in);

        // ---- ordered compaction of the columns stored on this tile (wave-local, no barrier)
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
        if (lane < 8) ridx[nact + lane] = (unsigned short)r;  // pipeline tail: zero Q row, any valid U row
        __builtin_amdgcn_wave_barrier();

        // ---- base term from the analytic prior: Wc0[i,:] = sum_b (sum_{f in block b} w_f P0[i, F_f]) L_inv[b,:]
        float acc[VEC][MC];
#pragma unroll
        for (int c = 0; c < VEC; ++c)
#pragma unroll
            for (int j = 0; j < MC; ++j) acc[c][j] = 0.f;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
            const int cell = min(cell0 + c, v.N - 1);
            const int row = cell / v.W, col = cell - row * v.W;
            for (int b = 0; b < m; ++b) {
                const Block blk = block_of(b, h.nx, h.rf, h.w, h.h);
                float cb = 0.f;
                for (int a = 0; a < blk.count(); ++a) {
                    const int ly = blk.y0 + a / blk.bw, lx = blk.x0 + a % blk.bw;
                    const int dr = abs(row - (h.yu + ly)), dc = abs(col - (h.xl + lx));
                    cb += use_lut ? lut[dr * v.W + dc] : matern_f(dr, dc, s3, h.sv);
                }
                cb *= (float)blk.weight;
#pragma unroll
                for (int j = 0; j < MC; ++j) acc[c][j] = fmaf(cb, Ls[b * MC + j], acc[c][j]);
            }
        }

        // ---- stream the stored rows: acc += row_k[cells] * Q[k,:]  (Q carries the sign of the downdate)
        {
            const int last = max(r - 1, 0);
            auto urow = [&](int a) -> const float* {
                const int k = __builtin_amdgcn_readfirstlane((int)ridx[min(a, nact + 7)]);
                return cov_src + (size_t)min(k, last) * npad + cell0;
            };
            auto consume = [&](const float (&u)[KP][VEC], int abase) {
#pragma unroll
                for (int i = 0; i < KP; ++i) {
                    const int k = __builtin_amdgcn_readfirstlane((int)ridx[abase + i]);
                    float qv[QS];
#pragma unroll
                    for (int t4 = 0; t4 < QS / 4; ++t4) {
                        const float4 q4 = *reinterpret_cast<const float4*>(&Qs[k * QS + 4 * t4]);
                        qv[4 * t4 + 0] = q4.x; qv[4 * t4 + 1] = q4.y; qv[4 * t4 + 2] = q4.z; qv[4 * t4 + 3] = q4.w;
                    }
#pragma unroll
                    for (int j = 0; j < MC; ++j)
#pragma unroll
                        for (int c = 0; c < VEC; ++c) acc[c][j] = fmaf(u[i][c], qv[j], acc[c][j]);
                }
            };
            if (nact > 0) {
                float ua[KP][VEC], ub[KP][VEC];
#pragma unroll
                for (int i = 0; i < KP; ++i) load_stream<VEC>(urow(i), ua[i]);
                for (int a = 0; a < nact; a += 2 * KP) {
#pragma unroll
                    for (int i = 0; i < KP; ++i) load_stream<VEC>(urow(a + KP + i), ub[i]);
                    consume(ua, a);
#pragma unroll
                    for (int i = 0; i < KP; ++i) load_stream<VEC>(urow(a + 2 * KP + i), ua[i]);
                    // second half of the ping-pong: indices a + KP .. may run past nact: they hit the zero Q row
                    consume(ub, min(a + KP, nact));
                }
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
        wave_part += wave_sum(part);
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
        __builtin_amdgcn_wave_barrier();
    }

    // ------------------------------------------------------------------ per-item results
    // No closing barrier: a wave that has no tile left publishes its partial sum and exits, freeing its slot;
    // the last wave to arrive (LDS counter) adds the partials in wave order (bit-reproducible) and writes the
    // item's reward, rank and the span of the appended columns.
    if (lane == 0) {
        red[wave] = wave_part;
        if (units) atomicAdd(v.counters, units);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    int arrived = 0;
    if (lane == 0) arrived = atomicAdd(done_waves, 1);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived != nw - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane == 0) {
        double tot = 0.0;
        for (int w = 0; w < nw; ++w) tot += red[w];
        reward_out[item] = (float)(tot / (h.cost_d + 1.0));  // rewards.py:31
        if (h.commit) v.rank[h.dst] = r + m;
    }
    if (h.commit && lane < m) v.colspan[(size_t)h.dst * v.rank_cap + r + lane] = h.t_lo | (h.t_hi << 16);
}

// Stand-alone gain kernel (after k_prepare): stages the item's scratch block and the prior table, then gain_tiles.
template <int MC, int VEC>
__global__ __launch_bounds__(512, IPP_GF_MINWAVES) void k_gain_factor(View v, int n_items, unsigned flags, int lut_cap,
                                                                   float* __restrict__ reward_out) {
    constexpr int QS = (MC + 3) & ~3;
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gf[];
    const GainLds<MC> lds(smem_gf, v.rank_cap, lut_cap);
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const int tid = threadIdx.x, T = blockDim.x;
    const ItemHdr h = v.hdr[item];
    const int r = h.rank;
    if (h.m == 0 || h.status == IPP_STATUS_NOT_PD) {
        if (tid == 0) reward_out[item] = (h.status == IPP_STATUS_NOT_PD) ? NAN : 0.f;
        return;
    }
    {
        const float4* src = reinterpret_cast<const float4*>(v.q + (size_t)item * v.q_item);
        float4* dst = reinterpret_cast<float4*>(lds.Ls);
        const int blk4 = (LQ + (r + 8) * QS) / 4;  // k_prepare zero-fills the 8 rows after Q: row r is the zero row
        for (int i = tid; i < blk4; i += T) dst[i] = src[i];
    }
    if (tid == 0) { *lds.next_tile = 0; *lds.done_waves = 0; }
    for (int k = tid; k < r; k += T) lds.span_s[k] = v.colspan[(size_t)h.env * v.rank_cap + k];
    const bool use_lut = v.N <= lut_cap;
    if (use_lut) {
        const float s3 = (float)(kSqrt3 * v.res) / h.ls;
        for (int i = tid; i < v.N; i += T) {
            const int dr = i / v.W, dc = i - dr * v.W;
            lds.lut[i] = matern_f(dr, dc, s3, h.sv);
        }
    }
    __syncthreads();
    gain_tiles<MC, VEC>(v, h, item, flags, use_lut, lds, reward_out);
}

}  // namespace ipp
