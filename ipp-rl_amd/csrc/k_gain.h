// Streaming kernels of the env step.
//
// k_gain: Wc = base + rows * Q for one tile of cells of one item, fused with everything that consumes Wc
// (mapping/mappings.py:188-197, planning/common/rewards.py:8-31):
//   factor state : rows = U[k][:] (k < rank), base = P0[:,F] H_F^T L^-1 evaluated from the analytic prior
//   dense state  : rows = P[F_f][:] (f footprint rows; P symmetric so rows == columns), base = 0
//   epilogue     : masked trace reduction -> reward, diag -= |Wc_i|^2, mean += Wc y, append Wc to U
//                  (factor) or park Wc for the downdate kernel (dense).
// k_downdate (dense only): P -= Wc Wc^T, one pass over P (mapping/mappings.py:190).
//
// Both are HBM-bound: every streamed 16 B feed 4*MC FMAs, the small operand (Q / Wc rows) is broadcast
// from LDS, consecutive lanes read consecutive 16 B (1 KiB per wave instruction), and the row loads are
// software-pipelined kPipe deep so each wave keeps kPipe KiB in flight.
#pragma once
#include "ipp_common.h"

namespace ipp {

// Tuning knobs (defaults are the measured best on MI355X); -D overrides are for A/B builds only.
#ifndef IPP_KPIPE
#define IPP_KPIPE 4
#endif
#ifndef IPP_MINWAVES
#define IPP_MINWAVES 4
#endif
#ifndef IPP_NT_STORES
#define IPP_NT_STORES 1  // appended rows: -2 % kernel time (A/B on MI355X)
#endif
#ifndef IPP_NT_LOADS
#define IPP_NT_LOADS 1  // +6..10 % on MI355X: rows are streamed once, Q / headers stay in L2
#endif
#ifndef IPP_GAIN_GROUP
#define IPP_GAIN_GROUP 4  // rows per request group (a carried 2 x 4 ping-pong measured 1.5 % slower, groups of 8 spill)
#endif
constexpr int kPipe = IPP_KPIPE;  // zero Q rows kept behind a staged chunk: 2 * kPipe (>= IPP_GAIN_GROUP); row pipeline of k_downdate

template <int VEC> struct VecIO;
template <> struct VecIO<4> {
    using T = float4;
    static __device__ __forceinline__ void unpack(const float4& a, float (&o)[4]) { o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; }
    static __device__ __forceinline__ float4 pack(const float (&o)[4]) { return make_float4(o[0], o[1], o[2], o[3]); }
};
template <> struct VecIO<2> {
    using T = float2;
    static __device__ __forceinline__ void unpack(const float2& a, float (&o)[2]) { o[0] = a.x; o[1] = a.y; }
    static __device__ __forceinline__ float2 pack(const float (&o)[2]) { return make_float2(o[0], o[1]); }
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&o)[VEC]) {
    VecIO<VEC>::unpack(*reinterpret_cast<const typename VecIO<VEC>::T*>(p), o);
}
// streamed-once rows: optional non-temporal hint (keeps Q / headers resident in L2)
template <int VEC>
__device__ __forceinline__ void load_stream(const float* p, float (&o)[VEC]) {
#if IPP_NT_LOADS
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    const vec_t t = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p));
#pragma unroll
    for (int c = 0; c < VEC; ++c) o[c] = t[c];
#else
    load_vec<VEC>(p, o);
#endif
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&o)[VEC]) {
    *reinterpret_cast<typename VecIO<VEC>::T*>(p) = VecIO<VEC>::pack(o);
}
// appended rows are not read again before the next step's kernels: optional non-temporal hint
template <int VEC>
__device__ __forceinline__ void store_stream(float* p, const float (&o)[VEC]) {
#if IPP_NT_STORES
    typedef float vec_t __attribute__((ext_vector_type(VEC)));
    vec_t t;
#pragma unroll
    for (int c = 0; c < VEC; ++c) t[c] = o[c];
    __builtin_nontemporal_store(t, reinterpret_cast<vec_t*>(p));
#else
    store_vec<VEC>(p, o);
#endif
}

// acc[c][j] += sum_k row_k[cell0 + c] * Q[k][j] over the item's streaming rows (the prologue stores Q with the sign
// of the update folded in).  rowidx (LDS) maps the streaming index to the row of the covariance slab (dense:
// footprint cell of P; factor: the columns of U stored on this tile).
// The Q rows are read straight from the item's global block through the constant address space
// (scalar loads: wave-uniform address, written by the prologue kernel before this launch): no LDS staging, no
// barriers in the loop, and with Q out of the vector registers the request groups hold G = 8 rows instead of 4.
// zero_row: index of the first of the >= 8 zero rows behind the item's Q rows (k_prepare.h).
#ifndef IPP_GAIN_GROUP_SQ
#define IPP_GAIN_GROUP_SQ 8
#endif
template <int MC, int VEC, int MODE>
__device__ __forceinline__ void stream_rows_sq(const float* __restrict__ cov_src, const int* rowidx, int rows, size_t npad,
                                               int cell0, const float* qrows, int zero_row, float (&acc)[VEC][MC]) {
    constexpr int QS = (MC + 3) & ~3;
    constexpr int G = IPP_GAIN_GROUP_SQ;
    static_assert(G <= 8, "the prologue keeps 8 zero rows behind Q");
    typedef const __attribute__((address_space(4))) float* cfloat_p;
    cfloat_p qc0 = (cfloat_p)(const void*)qrows;
    typedef float rowv __attribute__((ext_vector_type(VEC)));
    for (int kk = 0; kk < rows; kk += G) {
        rowv u[G];
        int qk[G];
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int sidx = kk + i;
            const int ri = __builtin_amdgcn_readfirstlane(rowidx[min(sidx, rows - 1)]);  // rows past the end: last row x zero Q
            u[i] = __builtin_nontemporal_load(reinterpret_cast<const rowv*>(cov_src + (size_t)ri * npad + cell0));
            qk[i] = sidx < rows ? (MODE == IPP_DENSE ? sidx : ri) : zero_row;
        }
        __builtin_amdgcn_sched_barrier(0);  // all G requests leave before the first wait
#pragma unroll
        for (int i = 0; i < G; ++i) {
            cfloat_p qc = qc0 + (size_t)qk[i] * QS;
            float qv[MC];
#pragma unroll
            for (int j = 0; j < MC; ++j) qv[j] = qc[j];
#pragma unroll
            for (int j = 0; j < MC; ++j)
#pragma unroll
                for (int c = 0; c < VEC; ++c) acc[c][j] = fmaf(u[i][c], qv[j], acc[c][j]);
        }
    }
}

template <int MC, int VEC, int MODE>
__global__ __launch_bounds__(kMaxTileThreads, IPP_MINWAVES) void k_gain(View v, int n_items, unsigned flags, int q_chunk,
                                                             int lut_cap, float* __restrict__ reward_out) {
    constexpr int QS = (MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gain[];
    // LDS carve: max(Qs[q_chunk + 2*kPipe][QS], lut[lut_cap]) | Ls[MC*MC] | ys[MC] (padded) | red[16] doubles | rowidx[q_chunk] (dense)
    float* Qs = reinterpret_cast<float*>(smem_gain);
    float* Ls = Qs + max((size_t)(q_chunk + 2 * kPipe) * QS, (size_t)((lut_cap + 3) & ~3));
    float* ys = Ls + ((MC * MC + 3) & ~3);
    double* red = reinterpret_cast<double*>(ys + ((MC + 3) & ~3));
    int* rowidx = reinterpret_cast<int*>(red + 16);

    int item, tile;
    if (!decode_block(blockIdx.x, n_items, v.n_tiles, item, tile)) return;
    const int tid = threadIdx.x;
    const ItemHdr h = uniform_hdr(v.hdr[item]);
    const int m = h.m;
    const int T = blockDim.x;
    // Q is read through the scalar cache, which is not coherent: drop whatever an earlier launch left of this
    // item's block (same address every step) before the first read
    __builtin_amdgcn_s_dcache_inv();

    if (m == 0 || h.status == IPP_STATUS_NOT_PD) {
        if (tid == 0) {
            const double bad = (h.status == IPP_STATUS_NOT_PD) ? (double)NAN : 0.0;
            if (v.n_tiles == 1) reward_out[item] = (float)bad; else v.partial[(size_t)item * v.n_tiles + tile] = bad;
        }
        return;
    }

    // tiles outside the span of the column(s) this step appends hold no part of Wc (window_rows > 0): nothing to do
    if (MODE == IPP_FACTOR && (tile < h.t_lo || tile > h.t_hi)) {
        if (tid == 0) {
            if (v.n_tiles == 1) reward_out[item] = 0.f; else v.partial[(size_t)item * v.n_tiles + tile] = 0.0;
        }
        return;
    }

    const int cell0 = tile * VEC * T + VEC * tid;
    constexpr int LQ = (MC * MC + MC + 3) & ~3;  // the item's scratch block is [L^-1 | y | pad | Q rows | zero rows]
    const float* __restrict__ qg = v.q + (size_t)item * v.q_item + LQ;
    const float* cov_src = v.cov + (size_t)h.env * v.cov_slot;
    float* cov_dst = v.cov + (size_t)h.dst * v.cov_slot;

    for (int i = tid; i < MC * MC; i += T) Ls[i] = v.linv[(size_t)item * MC * MC + i];
    if (tid < MC) ys[tid] = v.yv[(size_t)item * MC + tid];
    int rows = h.rows;
    if (MODE == IPP_DENSE) {
        for (int i = tid; i < rows; i += T) rowidx[i] = (h.yu + i / h.w) * v.W + h.xl + i % h.w;
    } else {
        // ordered compaction of the columns of U that are stored on this tile (all of them when window_rows == 0)
        const int* __restrict__ span = v.colspan + (size_t)h.env * v.rank_cap;
        const int lane = tid & (kWave - 1), wave = tid / kWave, nw = T / kWave;
        int* wcount = reinterpret_cast<int*>(red);  // red[] is free until the epilogue
        int base = 0;
        for (int k0 = 0; k0 < h.rows; k0 += T) {
            const int k = k0 + tid;
            bool on = false;
            if (k < h.rows) {
                const int sp = span[k];
                on = tile >= (sp & 0xffff) && tile <= (sp >> 16);
            }
            const unsigned long long mask = __ballot(on);
            if (lane == 0) wcount[wave] = __popcll(mask);
            __syncthreads();
            int off = base;
            for (int w = 0; w < wave; ++w) off += wcount[w];
            if (on) rowidx[off + __popcll(mask & ((1ull << lane) - 1ull))] = k;
            for (int w = 0; w < nw; ++w) base += wcount[w];
            __syncthreads();
        }
        rows = base;
    }
    __syncthreads();

    float acc[VEC][MC];
#pragma unroll
    for (int c = 0; c < VEC; ++c)
#pragma unroll
        for (int j = 0; j < MC; ++j) acc[c][j] = 0.f;

    // ------------------------------------------------------------------ base term from the analytic prior
    // Wc0[i,:] = sum_b (sum_{f in block b} w_f P0[i, F_f]) L_inv[b,:].  P0 depends on (|drow|, |dcol|) only, so
    // the workgroup tabulates it once in LDS (aliasing the Q staging area) instead of evaluating sqrt/exp
    // for every (cell, footprint cell) pair; grids too large for the table evaluate it directly.
    if (MODE == IPP_FACTOR) {
        const float s3 = (float)(kSqrt3 * v.res) / h.ls;
        const bool use_lut = v.N <= lut_cap;
        float* lut = Qs;
        if (use_lut) {
            for (int i = tid; i < v.N; i += T) {
                const int dr = i / v.W, dc = i - dr * v.W;
                lut[i] = matern_f(dr, dc, s3, h.sv);
            }
            __syncthreads();
        }
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
        if (use_lut) __syncthreads();  // the table is overwritten by Q below
    }

    // ------------------------------------------------------------------ streaming loop
    if (rows > 0)
        stream_rows_sq<MC, VEC, MODE>(cov_src, rowidx, rows, (size_t)v.Npad, cell0, qg, (MODE == IPP_DENSE) ? h.f : h.rank, acc);

    // ------------------------------------------------------------------ epilogue
    float mean_in[VEC], diag_in[VEC];
    load_vec<VEC>(v.mean + (size_t)h.env * v.Npad + cell0, mean_in);
    load_vec<VEC>(v.diag + (size_t)h.env * v.Npad + cell0, diag_in);
    float dred[VEC], dmean[VEC];
    double part = 0.0;
    const bool adaptive = (flags & IPP_ADAPTIVE) != 0;
#pragma unroll
    for (int c = 0; c < VEC; ++c) {
        const bool valid = (cell0 + c) < v.N;
        float w2 = 0.f, dm = 0.f;
        if (h.fallback) {
            // acc = Y = P H^T ; Z = Y S^-1 ; diag(P - Y S^-1 Y^T), x + Y S^-1 v   (mappings.py:204-212)
            for (int j = 0; j < m; ++j) {
                float zj = 0.f;
#pragma unroll
                for (int i = 0; i < MC; ++i) zj = fmaf(acc[c][i], Ls[i * MC + j], zj);
                float aj = 0.f;
#pragma unroll
                for (int i = 0; i < MC; ++i) aj = (i == j) ? acc[c][i] : aj;
                w2 = fmaf(aj, zj, w2);
            }
        } else {
#pragma unroll
            for (int j = 0; j < MC; ++j) w2 = fmaf(acc[c][j], acc[c][j], w2);
        }
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
    part = wave_sum(part);
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    if (lane == 0) red[wave] = part;
    __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int w = 0; w < (T + kWave - 1) / kWave; ++w) tot += red[w];
        if (v.n_tiles == 1)
            reward_out[item] = (float)(tot / (h.cost_d + 1.0));  // rewards.py:31
        else
            v.partial[(size_t)item * v.n_tiles + tile] = tot;
    }

    if (tid == 0) {  // traffic accounting: rows streamed (+ appended) on this tile, mean / diag read (+ write)
        const int valid = max(0, min(v.tile_cells, v.N - tile * v.tile_cells));
        const unsigned long long units = (unsigned long long)(rows + (h.commit ? m + 4 : 2)) * valid;
        atomicAdd(v.counters + (size_t)((item * 5 + tile) & (kCountSlots - 1)) * 16, units);
    }
    if (!h.commit) return;
    float outv[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) outv[c] = diag_in[c] - dred[c];
    store_vec<VEC>(v.diag + (size_t)h.dst * v.Npad + cell0, outv);
    if (!(flags & IPP_COV_ONLY)) {
#pragma unroll
        for (int c = 0; c < VEC; ++c) outv[c] = mean_in[c] + dmean[c];
        store_vec<VEC>(v.mean + (size_t)h.dst * v.Npad + cell0, outv);
    } else if (h.dst != h.env) {
        store_vec<VEC>(v.mean + (size_t)h.dst * v.Npad + cell0, mean_in);
    }
    if (h.dst != h.env) {
        float g[VEC];
        load_vec<VEC>(gt_plane(v, h.env) + cell0, g);
        store_vec<VEC>(gt_plane(v, h.dst) + cell0, g);
        if (tile == 0 && tid == 0) {
            v.prior[2 * h.dst + 0] = v.prior[2 * h.env + 0];
            v.prior[2 * h.dst + 1] = v.prior[2 * h.env + 1];
        }
    }
    if (MODE == IPP_FACTOR) {
        // append the m new columns (rows of the [k][cell] layout): coalesced row writes
#pragma unroll
        for (int j = 0; j < MC; ++j)
            if (j < m) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) outv[c] = acc[c][j];
                store_stream<VEC>(cov_dst + (size_t)(h.rank + j) * v.Npad + cell0, outv);
            }
        if (tile == h.t_lo) {  // one workgroup per item publishes the new rank and the span of the new columns
            if (tid == 0) v.rank[h.dst] = h.rank + m;
            if (tid < m) {
                v.colspan[(size_t)h.dst * v.rank_cap + h.rank + tid] = h.t_lo | (h.t_hi << 16);
                v.colrect[(size_t)h.dst * v.rank_cap + h.rank + tid] = (int)kRectFull;
            }
        }
    } else {
        float* wc = v.wc + (size_t)item * MC * v.Npad + cell0;
#pragma unroll
        for (int j = 0; j < MC; ++j) {
#pragma unroll
            for (int c = 0; c < VEC; ++c) outv[c] = acc[c][j];
            store_vec<VEC>(wc + (size_t)j * v.Npad, outv);
        }
    }
}

// Sum the per-tile reward partials in a fixed order (bit-reproducible regardless of GPU count).
__global__ void k_reward_finalize(View v, int n_items, float* __restrict__ reward_out) {
    const int item = blockIdx.x * blockDim.x + threadIdx.x;
    if (item >= n_items) return;
    double tot = 0.0;
    for (int t = 0; t < v.n_tiles; ++t) tot += v.partial[(size_t)item * v.n_tiles + t];
    reward_out[item] = (float)(tot / (v.hdr[item].cost_d + 1.0));
}

// Dense state: P_dst = P_src - Wc Wc^T (mapping/mappings.py:190), or P - Y S^-1 Y^T on the fallback
// path (:206).  One workgroup = kBandRows rows x one column tile; in place when dst == src.
template <int MC, int VEC>
__global__ __launch_bounds__(kMaxTileThreads, 4) void k_downdate(View v, int n_items, int n_bands) {
    constexpr int QS = (MC + 3) & ~3;
    __shared__ __attribute__((aligned(16))) float wi[(kBandRows + 2 * kPipe) * QS];
    __shared__ float Ss[MC * MC];
    int item, part;
    if (!decode_block(blockIdx.x, n_items, n_bands * v.n_tiles, item, part)) return;
    const ItemHdr h = uniform_hdr(v.hdr[item]);
    if (!h.commit || h.m == 0) return;
    const int band = part / v.n_tiles, tile = part - band * v.n_tiles;
    const int tid = threadIdx.x, T = blockDim.x;
    const int cell0 = tile * VEC * T + VEC * tid;
    const int row0 = band * kBandRows;
    const int nrows = min(kBandRows, v.N - row0);
    const float* wc = v.wc + (size_t)item * MC * v.Npad;

    for (int i = tid; i < (kBandRows + 2 * kPipe) * QS; i += T) {
        const int rr = i / QS, j = i - rr * QS;
        wi[i] = (rr < nrows && j < MC) ? -wc[(size_t)j * v.Npad + row0 + rr] : 0.f;  // negated once here
    }
    if (h.fallback)
        for (int i = tid; i < MC * MC; i += T) Ss[i] = v.linv[(size_t)item * MC * MC + i];

    // this thread's cells of Wc (normal path) or of Z = Y S^-1 (fallback, mappings.py:206: accumulated row by row so
    // that no second m-vector per cell is live -- the unrolled Y -> Z transform used to cost 38 spilled registers)
    float wj[VEC][MC];
    if (h.fallback) {
        __syncthreads();  // Ss staged
#pragma unroll
        for (int c = 0; c < VEC; ++c)
#pragma unroll
            for (int j = 0; j < MC; ++j) wj[c][j] = 0.f;
        for (int i = 0; i < MC; ++i) {
            float t[VEC];
            load_vec<VEC>(wc + (size_t)i * v.Npad + cell0, t);
#pragma unroll
            for (int j = 0; j < MC; ++j) {
                const float sij = Ss[i * MC + j];
#pragma unroll
                for (int c = 0; c < VEC; ++c) wj[c][j] = fmaf(t[c], sij, wj[c][j]);
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < MC; ++j) {
            float t[VEC];
            load_vec<VEC>(wc + (size_t)j * v.Npad + cell0, t);
#pragma unroll
            for (int c = 0; c < VEC; ++c) wj[c][j] = t[c];
        }
    }
    __syncthreads();
    const float* psrc = v.cov + (size_t)h.env * v.cov_slot + (size_t)row0 * v.Npad + cell0;
    float* pdst = v.cov + (size_t)h.dst * v.cov_slot + (size_t)row0 * v.Npad + cell0;
    const int last = nrows - 1;
    // groups of G rows requested together, downdated and written back; nothing loaded is carried over the back edge
    // (a carried ping-pong group compiles to register copies that each wait for their load, see k_gain_wave.h)
    constexpr int G = 8;
    static_assert(G <= 2 * kPipe, "wi holds 2 * kPipe zero rows behind the band");
    for (int rr = 0; rr < nrows; rr += G) {
        float p[G][VEC];
#pragma unroll
        for (int i = 0; i < G; ++i) load_stream<VEC>(psrc + (size_t)min(rr + i, last) * v.Npad, p[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) {
            float a[QS];
#pragma unroll
            for (int t4 = 0; t4 < QS / 4; ++t4) {
                const float4 q4 = *reinterpret_cast<const float4*>(&wi[(rr + i) * QS + 4 * t4]);
                a[4 * t4 + 0] = q4.x; a[4 * t4 + 1] = q4.y; a[4 * t4 + 2] = q4.z; a[4 * t4 + 3] = q4.w;
            }
#pragma unroll
            for (int j = 0; j < MC; ++j)
#pragma unroll
                for (int c = 0; c < VEC; ++c) p[i][c] = fmaf(a[j], wj[c][j], p[i][c]);
            if (rr + i < nrows) store_stream<VEC>(pdst + (size_t)(rr + i) * v.Npad, p[i]);
        }
    }
}

}  // namespace ipp
