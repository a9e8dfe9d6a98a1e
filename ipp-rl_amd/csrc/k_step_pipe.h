// Fused factor-state env step, PIPELINED: a persistent grid of 256-thread workgroups, each a producer / consumer
// pipeline over the items it draws from a device ticket counter.
//
//   wave 0      PRODUCER.  For item k+1 while the consumers stream item k: the whole per-item prologue by one wave
//               (prepare_item_ex<.., WAVE>: header, observation, gather of HT = H_F U[F,:]^T, S, Cholesky, L^-1, y,
//               Q = -HT L^-1), the prior table, measurement-block tables and adaptive-mask bits of the touched tiles,
//               into one of two LDS item buffers; the Q rows go to the item's global scratch block (read back by the
//               consumers through the scalar cache).  Then it publishes the buffer through an LDS sequence number.
//   waves 1..3  CONSUMERS.  gain_tiles over the published item: tiles from an LDS ticket, prior term + stream of the stored
//               columns of U against Q (SGPR operands), fused reward / diag / mean / append.  A consumer that runs out of
//               tiles moves on to the next published item by itself; the buffer returns to the producer when all three
//               have left it.  No workgroup barrier anywhere.
//
// Why (timeline of k_step_factor, one workgroup per item, tools/timeline.py): a workgroup spent 17-19 us in the prologue's
// chain of dependent memory round trips and its slot stood empty for another 10 us (50x50, 4096 items) to 23 us
// (100x100, 32768 items) between its exit and the start of its successor (whole-workgroup dispatch behind an in-order,
// per-XCD queue), against 33-38 us of streaming: about half of the resident waves had requests in flight, and achieved
// bandwidth follows that number.  Here three of four waves stream all the time, the prologue latency is hidden behind the
// previous item's stream, and nothing is dispatched after the launch.
//
// STATUS: correct (tests/test_hip_sharding.py::test_pipelined_step_kernel_matches_default_and_oracle) but NOT faster, and
// therefore opt-in (IPP_PIPE=1).  Measured at 4096 items of 50x50 (tools/timeline.py marks): the single producer wave
// needs ~50 us per item (inputs 5, observation + tables + mask 17, gather of HT 13, m x m algebra + Q rows 16) against
// ~37 us that its three consumers need to stream the item, so the pipeline is producer-bound: 0.357 ms per launch
// against 0.313 ms for k_step_factor.  What was tried on the producer: the m x m algebra in registers (30 -> 16 us),
// INTER_AREA weights one per lane, the producer as an out-of-line function with its own register allocation (the inlined
// kernel spills 60-150 VGPRs; out of line the argument copies through scratch made it slower: 58 us).  What it would
// take: <= 35 us per item, i.e. the next item's inputs prefetched under the current item and fewer registers in the
// producer path (a 7 x 9 lane gather with 112 rows per pass cut the gather 13 -> 11 us but the extra live registers
// cost as much in the table phase: 52 us).  A persistent grid also keeps every CU busy until it ends, so the
// side-stream ground-truth kernels of VecIPPEnv cannot overlap with it (configs[2]: +0.6 ms per step).
#pragma once
#include "ipp_common.h"
#include "k_gain_factor.h"
#include "k_prepare.h"

namespace ipp {

constexpr int kPipeThreads = 256;
constexpr int kPipeConsumers = kPipeThreads / kWave - 1;
constexpr int kTicketSlots = 64;  // ring of launch ticket counters (View::tickets): slot s is zeroed by launch s - 32

struct PipeCtl {
    int seq_ready[2];  // number of items published into buffer p so far
    int released[2];   // consumer waves that have left buffer p so far (kPipeConsumers per item)
    int total;         // -1 while the producer still draws items, else the number of items it published
    int pad[3];
};

// LDS carve.  Per item buffer: a GainLds block without work / small / per-wave areas + the item header.
template <int MC>
struct PipeLds {
    static constexpr int QS = (MC + 3) & ~3;
    __host__ __device__ static size_t buf_bytes(int rank_cap, int lut_floats, int win_tiles) {
        return ((GainLds<MC>::bytes(rank_cap, 0, lut_floats, 0, 0, win_tiles, win_tiles * kWave) + 16 + sizeof(ItemHdr) + 15) & ~(size_t)15);
    }
    __host__ __device__ static size_t ht_floats(int rank_cap) { return (size_t)MC * ((rank_cap + 3) & ~3); }
    __host__ __device__ static size_t bytes(int rank_cap, int lut_floats, int win_tiles) {
        size_t b = sizeof(PipeCtl);
        b += (prep_small_bytes<MC>() + 15) & ~(size_t)15;
        b += (ht_floats(rank_cap) * 4 + 15) & ~(size_t)15;
        b += ((size_t)(kPipeThreads / kWave) * (rank_cap + 8) * 2 + 15) & ~(size_t)15;
        b += 2 * buf_bytes(rank_cap, lut_floats, win_tiles);
        return (b + 15) & ~(size_t)15;
    }
};

template <int MC, int VEC>
__global__ __launch_bounds__(kPipeThreads, IPP_GF_MINWAVES) void k_step_pipe(
    View v, const int* __restrict__ env_ids, int n_items,
    const double* __restrict__ action, const double* __restrict__ prev_action, const float* __restrict__ meas_noise,
    unsigned flags, int lut_rows, int* __restrict__ status_out, float* __restrict__ reward_out, int ticket_slot) {
    constexpr int QS = (MC + 3) & ~3;
    constexpr int LQ = (MC * MC + MC + 3) & ~3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_pp[];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int lutf = lut_rows * v.W;

    // ---- carve
    PipeCtl* ctl = reinterpret_cast<PipeCtl*>(smem_pp);
    unsigned char* small = smem_pp + sizeof(PipeCtl);
    float* ht = reinterpret_cast<float*>(small + ((prep_small_bytes<MC>() + 15) & ~(size_t)15));
    unsigned short* ridx_all = reinterpret_cast<unsigned short*>(reinterpret_cast<unsigned char*>(ht) + ((PipeLds<MC>::ht_floats(v.rank_cap) * 4 + 15) & ~(size_t)15));
    unsigned char* bufs = reinterpret_cast<unsigned char*>(ridx_all) + (((size_t)(kPipeThreads / kWave) * (v.rank_cap + 8) * 2 + 15) & ~(size_t)15);
    const size_t buf_sz = PipeLds<MC>::buf_bytes(v.rank_cap, lutf, v.win_tiles);
    // buffer p: [GainLds block | item index (16 B) | ItemHdr]
    auto buffer = [&](int p, ItemHdr*& hdr, int*& item_slot) -> GainLds<MC> {
        unsigned char* base = bufs + (size_t)p * buf_sz;
        GainLds<MC> g(base, v.rank_cap, 0, lutf, 0, 0, v.win_tiles, v.win_tiles * kWave);
        g.ridx_all = ridx_all;
        hdr = reinterpret_cast<ItemHdr*>(base + buf_sz - sizeof(ItemHdr));
        item_slot = reinterpret_cast<int*>(base + buf_sz - sizeof(ItemHdr) - 16);
        return g;
    };

    if (tid == 0) {
        ctl->seq_ready[0] = ctl->seq_ready[1] = 0;
        ctl->released[0] = ctl->released[1] = 0;
        ctl->total = -1;
        if (blockIdx.x == 0) v.tickets[(ticket_slot + kTicketSlots / 2) & (kTicketSlots - 1)] = 0;  // for a later launch
    }
    __syncthreads();  // (the only workgroup barrier: before the roles split)
    int* ticket = v.tickets + ticket_slot;

    if (wave == 0) {
        // ============================================================ PRODUCER
        int k = 0;  // items published by this workgroup
        for (;;) {
            int item = 0;
            if (lane == 0) item = atomicAdd(ticket, 1);
            item = __builtin_amdgcn_readfirstlane(item);
            if (item >= n_items) break;
            const int p = k & 1;
            // the buffer is free once the consumers have left the item published into it two items ago
            while (__hip_atomic_load(&ctl->released[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < kPipeConsumers * (k >> 1))
                __builtin_amdgcn_s_sleep(2);
            ItemHdr* hbuf;
            int* item_slot;
            const GainLds<MC> g = buffer(p, hbuf, item_slot);
            if (lane == 0) IPP_MARK(item, 0);
            auto mid = [&](const ItemHdr& hh) {
                if (lane == 0) { *g.next_tile = 0; *g.done_waves = 0; g.red[0] = 0.0; g.red[1] = 0.0; }
                fill_block_tables<MC>(hh, g.fb_yx, g.fb_w);
                // adaptive mask of the touched tiles (rewards.py:11: pre-step mean and pre-step diag), one byte per VEC cells
                typedef float cellv __attribute__((ext_vector_type(VEC)));
                const cellv* mean_v = reinterpret_cast<const cellv*>(v.mean + (size_t)hh.env * v.Npad);
                const cellv* diag_v = reinterpret_cast<const cellv*>(v.diag + (size_t)hh.env * v.Npad);
                const bool adaptive = (flags & IPP_ADAPTIVE) != 0;
                constexpr int kMaskPerThread = 8;
                const int q_lo = hh.t_lo * kWave, q_hi = (hh.t_hi + 1) * kWave;
                for (int q0 = q_lo + lane; q0 < q_hi; q0 += kMaskPerThread * kWave) {
                    cellv mu[kMaskPerThread], dg[kMaskPerThread];
#pragma unroll
                    for (int u = 0; u < kMaskPerThread; ++u) {
                        const int q = q0 + u * kWave;
                        if (adaptive && q < q_hi) { mu[u] = mean_v[q]; dg[u] = diag_v[q]; }
                    }
#pragma unroll
                    for (int u = 0; u < kMaskPerThread; ++u) {
                        const int q = q0 + u * kWave;
                        if (q >= q_hi) continue;
                        unsigned bits = 0;
#pragma unroll
                        for (int c = 0; c < VEC; ++c) {
                            const bool in = !adaptive || ((double)mu[u][c] + v.kf * (double)dg[u][c] >= v.thr);
                            bits |= (in ? 1u : 0u) << c;
                        }
                        g.mask4[q - q_lo] = (unsigned char)bits;
                    }
                }
                const float s3 = (float)(kSqrt3 * v.res) / hh.ls;
                for (int i = lane; i < lutf; i += kWave) {
                    const int dr = i / v.W, dc = i - dr * v.W;
                    g.lut[i] = matern_f(dr, dc, s3, hh.sv);
                }
            };
            float* qrows_w = v.q + (size_t)item * v.q_item + LQ;
            ItemHdr* hs;
            if constexpr (MC == 9) {  // prologue up to the gather, then the m x m algebra in registers + Q rows
                hs = prepare_item_ex<MC, IPP_FACTOR, kWave, true, decltype(mid), false, true>(
                    v, item, env_ids, nullptr, action, prev_action, meas_noise, flags, status_out, nullptr, nullptr, nullptr, small,
                    ht, 0, 1, nullptr, g.Ls, nullptr, g.ys, nullptr, g.span_s, mid);
                const ItemHdr hf = uniform_hdr(*hs);
                if (hf.m > 0) solve_wave_fast<MC>(v, hf, item, flags, small, ht, (hf.rank + 3) & ~3, 1, g.Ls, g.ys, qrows_w, status_out);
            } else {
                hs = prepare_item_ex<MC, IPP_FACTOR, kWave, false, decltype(mid), false, true>(
                    v, item, env_ids, nullptr, action, prev_action, meas_noise, flags, status_out, nullptr, nullptr, nullptr, small,
                    ht, 0, 1, qrows_w, g.Ls, nullptr, g.ys, nullptr, g.span_s, mid);
            }
            wave_lds_sync();
            const ItemHdr h = uniform_hdr(*hs);
            if (lane == 0) IPP_MARK(item, 1);
            if (h.m == 0 || h.status == IPP_STATUS_NOT_PD) {  // nothing to stream: the producer settles the item itself
                if (lane == 0) reward_out[item] = (h.status == IPP_STATUS_NOT_PD) ? NAN : 0.f;
                continue;
            }
            if (lane == 0) { *hbuf = *hs; *item_slot = item; }
            // Q rows: vector stores, read back by the consumers through the scalar cache (from L2): every store has to be
            // acknowledged before the buffer is published (k_step_factor.h, same ordering)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(&ctl->seq_ready[p], (k >> 1) + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++k;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(&ctl->total, k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return;
    }

    // ================================================================ CONSUMERS
    for (int k = 0;; ++k) {
        const int p = k & 1;
        for (;;) {
            if (__hip_atomic_load(&ctl->seq_ready[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) > (k >> 1)) break;
            const int tot = __hip_atomic_load(&ctl->total, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (tot >= 0 && k >= tot) return;
            __builtin_amdgcn_s_sleep(2);
        }
        __builtin_amdgcn_s_dcache_inv();  // Q rows of this item come through the (non-coherent) scalar cache
        ItemHdr* hbuf;
        int* item_slot;
        const GainLds<MC> g = buffer(p, hbuf, item_slot);
        const ItemHdr h = uniform_hdr(*hbuf);
        const int item = __builtin_amdgcn_readfirstlane(*item_slot);
        gain_tiles<MC, VEC, sf_pipe<MC, VEC>(), false, true, false, true, false, kPipeConsumers>(
            v, h, item, flags, lut_rows, g, v.q + (size_t)item * v.q_item + LQ, reward_out);
        // this wave is done with the buffer (the last one out of gain_tiles has also written the item's results)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(&ctl->released[p], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

}  // namespace ipp
