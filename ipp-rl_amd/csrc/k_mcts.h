// Device-side tree search for the MCTS-zero planner (planning/mcts_zero/mcts.py of the reference):
// PUCT selection with forced playouts (:280-296), valid-action sets (:148-158), transposition-aware child lookup (the
// reference keys nodes by hash(str(P)): states reached by the same measurements in any order are one node), expansion
// with uniform or network priors and Dirichlet noise at the root (:160-164, 204-233), value backup (:255-265).
//
// The host drivers (planning/mcts_zero/mcts.py, vector_mcts.py) spend 95 % of a configs[4] search in NumPy: 5 s per
// 1024 roots x 256 simulations around 0.1 s of device work.  Here the search tables live in HBM and one WAVE owns one
// root: everything of a root (its node range, its hash table, its device-node range, the simulations in flight) is
// touched by that wave only and in program order, so there are no atomics except the per-level request counters, and
// the search is deterministic.  Per wave of W simulations in flight per root:
//   k_mcts_select   all roots: W descents each (virtual visits keep them apart), recording paths, pending leaves and the
//                   covariance steps of edges traversed for the first time (ONE request list per wave of simulations that IS
//                   the argument arrays of ipp_tree_step: the steps are independent of each other -- a node created in this
//                   wave stays a leaf until the wave's expansion -- so one launch takes them all), their path arguments and
//                   the device paths of the new nodes
//   ipp_tree_step   on the request list (k_tree_patch writes the edge numerators itself, band-tile engines through k_mcts_apply)
//   k_mcts_expand   valid-action sets and priors of the pending leaves (network replies optional)
//   k_mcts_backup   values back along the recorded paths
// Arithmetic is fp64 with contraction off and in the operand order of the NumPy driver (vector_mcts.py), which builds the
// same trees when ties are broken by the lowest action index (tests/test_hip_mcts.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ipp_engine.h"
#include "ipp_common.h"

namespace ipp {

constexpr int kMctsPath = 6;  // == kTreeDepth: device nodes on a path
constexpr unsigned char kNodeExpanded = 1, kNodeStored = 2;

__device__ __forceinline__ double mc_first(double x) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double mc_shfl_xor(double x, int m) { return __shfl_xor(x, m, 64); }
// Wave-wide max / min without LDS permutes: a butterfly of __shfl_xor is six DEPENDENT ds_bpermute round trips per value (two per fp64),
// and a level of the selection reduced five values for the Q range and six for the argmax -- 84 permutes, most of the 2 us that "PUCT +
// argmax" cost a lone wave.  Quads, half rows and rows through DPP moves, the four row results through v_readlane (as wave_sum_dpp);
// max / min do not round, so the result does not depend on the order.  Every lane returns the result.
__device__ __forceinline__ double mc_wave_max(double x) {
    x = fmax(x, dpp_mov_f64<0xB1>(x));
    x = fmax(x, dpp_mov_f64<0x4E>(x));
    x = fmax(x, dpp_mov_f64<0x141>(x));
    x = fmax(x, dpp_mov_f64<0x140>(x));
    return fmax(fmax(bcast_lane(x, 0), bcast_lane(x, 16)), fmax(bcast_lane(x, 32), bcast_lane(x, 48)));
}
__device__ __forceinline__ double mc_wave_min(double x) {
    x = fmin(x, dpp_mov_f64<0xB1>(x));
    x = fmin(x, dpp_mov_f64<0x4E>(x));
    x = fmin(x, dpp_mov_f64<0x141>(x));
    x = fmin(x, dpp_mov_f64<0x140>(x));
    return fmin(fmin(bcast_lane(x, 0), bcast_lane(x, 16)), fmin(bcast_lane(x, 32), bcast_lane(x, 48)));
}
__device__ __forceinline__ int mc_wave_min_i32(int x) {
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x141, 0xf, 0xf, true));
    x = min(x, __builtin_amdgcn_update_dpp(0, x, 0x140, 0xf, 0xf, true));
    return min(min(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)), min(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}
__device__ __forceinline__ double mc_readlane(double x, int lane_s) {  // lane_s wave-uniform
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, lane_s), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), lane_s);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// splitmix64: tie-break draws and the Dirichlet noise (counter-based: a draw depends on (seed, root, simulation, ...) only)
__device__ __forceinline__ uint64_t mc_mix(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ double mc_u01(uint64_t h) { return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

// actions.py:8-41 (Euclidean distance or the trapezoidal flight time), operand order of BatchedMCTS.row_cost
__device__ __forceinline__ double mc_distance(const double* a, const double* p) {
#pragma clang fp contract(off)
    const double d0 = a[0] - p[0], d1 = a[1] - p[1], d2 = a[2] - p[2];
    return sqrt((d0 * d0 + d1 * d1) + d2 * d2);
}
__device__ __forceinline__ double mc_cost(const ipp_mcts_tables& m, const double* a, const double* p) {
#pragma clang fp contract(off)
    const double dist = mc_distance(a, p);
    if (!m.use_flight_time) return dist;
    const double ramp = fmin(0.5 * dist, (m.vmax * m.vmax) / (2 * m.amax));
    return (dist - 2 * ramp) / m.vmax + 2 * sqrt(2 * ramp / m.amax);
}

// (timing build -DIPP_MCTS_CLOCKS=1: lane 0 stamps the sections of a level with the 100 MHz wall clock -- every stamp waits for the
// loads before it, so the sections do not overlap as they do in the product build; sums printed when the engine is destroyed)
#ifndef IPP_MCTS_CLOCKS
#define IPP_MCTS_CLOCKS 0
#endif
#if IPP_MCTS_CLOCKS
__device__ unsigned long long g_mclk[16];
#define MC_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long n_ = wall_clock64(); mclk_[k] += n_ - mt_; mt_ = n_; } while (0)
#else
#define MC_STAMP(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------------------------- selection
// One wave per root, W descents one after the other.  A descent is a chain of dependent memory round trips (node
// record -> edge rows -> the chosen edge -> child), so the kernel's time is the number of round trips per level:
// NE > 0 keeps the node's edge rows in registers (kmax <= 64 NE entries, all loads of a level's rows in one round trip,
// the chosen edge's fields in a second, the child's record in a third); NE = 0 is the general form for wider rows.
template <int NE>
__global__ __launch_bounds__(256) void k_mcts_select(ipp_mcts_tables m, const int32_t* __restrict__ root_env,
                                                     const double* __restrict__ prev0, const double* __restrict__ budget0,
                                                     int depth0, int sim0, int W, uint64_t seed) {
#pragma clang fp contract(off)
    constexpr int NR = NE > 0 ? NE : 1;
    const int j = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (j >= m.roots) return;
    const int base = j * m.nodes_per_root;
    const bool virt = W > 1;
    uint64_t* hk = m.h_keys + (size_t)j * m.table_size;
    int32_t* hv = m.h_vals + (size_t)j * m.table_size;
#if IPP_MCTS_CLOCKS
    unsigned long long mclk_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mt_ = wall_clock64(), mlev_ = 0;
#endif
    for (int w = 0; w < W; ++w) {
        int cur = base, plen = 0, leafnode = -1;
        double prev[3] = {prev0[3 * j], prev0[3 * j + 1], prev0[3 * j + 2]};
        double budget = budget0[j];
        // record of the current node (the next level's is loaded at the end of a level, next to the chosen edge)
        unsigned char fl = m.n_flags[cur];
        int K = m.n_k[cur];
        double ns = m.n_ns[cur];
        MC_STAMP(0);
        for (int d = depth0; d <= m.horizon; ++d) {
            if (!(budget > 0)) break;  // mcts.py:175-176
            if (!(fl & kNodeExpanded)) {
                // ---- leaf: evaluated (once per wave of simulations) by k_mcts_expand
                leafnode = cur;
                if (lane == 0) {
                    const int cnt = m.pend_count[j];
                    bool seen = false;
                    for (int s = 0; s < cnt; ++s) seen |= m.pend_node[j * m.wave + s] == cur;
                    if (!seen) {
                        const int s = j * m.wave + cnt;
                        m.pend_node[s] = cur;
                        m.pend_depth[s] = d;
                        m.pend_sim[s] = sim0 + w;
                        m.pend_prev[3 * s] = prev[0]; m.pend_prev[3 * s + 1] = prev[1]; m.pend_prev[3 * s + 2] = prev[2];
                        m.pend_budget[s] = budget;
                        m.pend_count[j] = cnt + 1;
                    }
                }
                break;
            }
            // ---- PUCT over the node's K valid actions (mcts.py:280-296)
            const size_t row = (size_t)cur * m.kmax;
            double rq[NR], rn[NR], rp[NR];
            int ri[NR], rc[NR];
            if (NE > 0) {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int k = min(lane + 64 * i, m.kmax - 1);
                    rq[i] = m.t_qsa[row + k]; rn[i] = m.t_nsa[row + k]; rp[i] = m.t_ps[row + k]; ri[i] = m.t_idx[row + k];
                    rc[i] = m.t_child[row + k];  // (with the rows: the chosen edge's child is known one round trip earlier)
                }
            }
            const int pv = lane < kMctsPath ? m.n_devpath[(size_t)kMctsPath * cur + lane] : -1;  // (this node's device path: lanes 0..5)
            // the factors of the prior that depend on Ns only: from the host's table when it reaches (requested with the rows)
            double pc_ld, sq_ld;
            {
                const long long nsi = (long long)ns;
                if (m.puct_c && nsi >= 0 && nsi < m.ns_table_n && (double)nsi == ns) { pc_ld = m.puct_c[nsi]; sq_ld = m.sqrt_ns1[nsi]; }
                else { pc_ld = m.puct_init + log((ns + m.puct_base + 1) / m.puct_base); sq_ld = sqrt(ns + 1); }
            }
            double lo = INFINITY, hi = -INFINITY;
            int nz = 0;
            if (NE > 0) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    if (lane + 64 * i < K) { lo = fmin(lo, rq[i]); hi = fmax(hi, rq[i]); nz |= (rq[i] != 0.0) ? 1 : 0; }
            } else {
                for (int k = lane; k < K; k += 64) {
                    const double q = m.t_qsa[row + k];
                    lo = fmin(lo, q);
                    hi = fmax(hi, q);
                    nz |= (q != 0.0) ? 1 : 0;
                }
            }
            lo = mc_wave_min(lo);
            hi = mc_wave_max(hi);
            nz = __ballot(nz != 0) != 0ull ? 1 : 0;
            MC_STAMP(1);
            if (K < m.num_actions) { lo = fmin(lo, 0.0); hi = fmax(hi, 0.0); }  // the zeros of the invalid actions take part (mcts.py:267-278)
            // (wave-uniform facts as scalars: the compiler then branches instead of evaluating every form of qn -- two fp64 divisions
            // per edge -- and selecting)
            const bool allzero = __builtin_amdgcn_readfirstlane(nz) == 0;
            const bool flat = __builtin_amdgcn_readfirstlane((int)(lo == hi)) != 0;
            const double pc = pc_ld, sq = sq_ld;  // (requested with the rows)
            const bool force = d == 0;
            double best = -INFINITY, best_u = -1.0, best_nsa = 0.0;
            int best_k = 0x7fffffff, best_a = -1, best_c = -2;  // (best_c = -2: not loaded with the rows)
            auto consider = [&](int k, double q, double nsa, double ps, int ai, int ci) {
                double qn = q;
                if (!allzero) { if (flat) qn = q / hi; else qn = (q - lo) / (hi - lo); }
                double uct = qn + pc * (ps * (sq / (1 + nsa)));
                if (force) {
                    double nfp = ceil(sqrt(m.fpf * ps * ns));
                    if (nsa == 0) nfp = 0;
                    if (nsa < nfp) uct = INFINITY;
                }
                const double u = m.tie_break ? mc_u01(mc_mix(seed ^ mc_mix(((uint64_t)(uint32_t)(cur + m.root_base * m.nodes_per_root) << 32) | (uint32_t)(k + 1)) ^ ((uint64_t)(sim0 + w) << 20))) : 0.0;
                if (uct > best || (uct == best && (m.tie_break ? u > best_u : k < best_k))) { best = uct; best_k = k; best_u = u; best_nsa = nsa; best_a = ai; best_c = ci; }
            };
            if (NE > 0) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    if (lane + 64 * i < K) consider(lane + 64 * i, rq[i], rn[i], rp[i], ri[i], rc[i]);
            } else {
                for (int k = lane; k < K; k += 64) consider(k, m.t_qsa[row + k], m.t_nsa[row + k], m.t_ps[row + k], m.t_idx[row + k], -2);
            }
            // the winner over the lanes: largest uct, then (random tie-break) largest draw, then lowest k -- a total order, so the result
            // is the butterfly's.  ONE max-reduction of uct and a ballot name it unless lanes tie (forced playouts: uct = inf on several
            // edges), the other fields come from its lane (k = lane + 64 i: the lane of an edge is k & 63) by v_readlane.
            {
                const double bmax = mc_wave_max(best);
                unsigned long long tie = __ballot(best == bmax);
                int win = (int)__builtin_ctzll(tie);
                if (tie & (tie - 1ull)) {  // (wave-uniform)
                    bool in = best == bmax;
                    if (m.tie_break) {
                        const double umax = mc_wave_max(in ? best_u : -1.0);
                        in = in && best_u == umax;
                    }
                    win = mc_wave_min_i32(in ? best_k : 0x7fffffff) & 63;
                }
                win = __builtin_amdgcn_readfirstlane(win);
                best_nsa = mc_readlane(best_nsa, win);
                best_k = __builtin_amdgcn_readlane(best_k, win);
                best_a = __builtin_amdgcn_readlane(best_a, win);
                best_c = __builtin_amdgcn_readlane(best_c, win);
            }
            MC_STAMP(2);
            const int k = best_k, a_idx = best_a;  // (K >= 1 for an expanded node)
            // ---- the chosen edge: everything it needs in one round trip
            const double action[3] = {m.actions[3 * (size_t)a_idx], m.actions[3 * (size_t)a_idx + 1], m.actions[3 * (size_t)a_idx + 2]};
            int child = (NE > 0) ? best_c : m.t_child[row + k];
            const double num = m.t_num[row + k];
            const uint64_t zk = m.zkey[a_idx], hcur = m.n_hash[cur];
            // (an existing child's record in the same round trip as the edge's fields)
            const bool child_known = child >= 0;
            const int cpre = child_known ? child : cur;
            const unsigned char cfl_pre = m.n_flags[cpre];
            const int cK_pre = m.n_k[cpre];
            const double cns_pre = m.n_ns[cpre];
            const double cost = mc_cost(m, action, prev);
            MC_STAMP(3);
            if (child < 0) {  // (wave-uniform) transposition lookup: the same measurements in any order are one node
                if (lane == 0) {
                    uint64_t key = hcur + zk;
                    if (key == 0) key = 1;
                    unsigned slot = (unsigned)(mc_mix(key) & (uint64_t)(m.table_size - 1));
                    for (;;) {
                        const uint64_t kk = hk[slot];
                        if (kk == key) { child = hv[slot]; break; }
                        if (kk == 0) {
                            const int cnt = m.root_count[j];
                            if (cnt >= m.nodes_per_root) { m.err[0] = 1; child = base; break; }  // node range exhausted
                            child = base + cnt;
                            m.root_count[j] = cnt + 1;
                            hk[slot] = key;
                            hv[slot] = child;
                            m.n_hash[child] = key;
                            break;
                        }
                        slot = (slot + 1) & (unsigned)(m.table_size - 1);
                    }
                    m.t_child[row + k] = child;
                }
                child = __builtin_amdgcn_readfirstlane(child);
            }
            MC_STAMP(4);
            // ---- the child's record (the next level's node) next to this level's bookkeeping
            const unsigned char cfl = child_known ? cfl_pre : m.n_flags[child];
            const int cK = child_known ? cK_pre : m.n_k[child];
            const double cns = child_known ? cns_pre : m.n_ns[child];
            int pth[kMctsPath];
#pragma unroll
            for (int s = 0; s < kMctsPath; ++s) pth[s] = __builtin_amdgcn_readlane(pv, s);
            MC_STAMP(5);
            if (lane == 0) {
                if (isnan(num)) {  // first traversal of the edge: one device step
                    m.t_num[row + k] = INFINITY;
                    int newdev = -1;
                    if (d + 1 <= m.horizon && !(cfl & kNodeStored)) {
                        const int dc = m.dev_count[j];
                        if (dc >= m.dev_per_root) m.err[1] = 1;  // device-node range exhausted: the state is not kept
                        else {
                            newdev = m.dev_base + j * m.dev_per_root + dc;
                            m.dev_count[j] = dc + 1;
                            m.n_flags[child] = cfl | kNodeStored;
                        }
                    }
                    // ONE request list per wave of simulations: the requested steps do not depend on each other (a node that gets its
                    // device state in this wave is a leaf until the wave's expansion, so no descent of the wave goes below it)
                    const size_t r = (size_t)atomicAdd(m.rq_count, 1);
                    m.rq_root[r] = root_env[j];
                    m.rq_parent[r] = cur;
                    m.rq_k[r] = k;
                    m.rq_child[r] = child;
                    m.rq_newdev[r] = newdev;
                    m.rq_cost[r] = cost;
                    m.rq_prev[3 * r] = prev[0]; m.rq_prev[3 * r + 1] = prev[1]; m.rq_prev[3 * r + 2] = prev[2];
                    m.rq_action[3 * r] = action[0]; m.rq_action[3 * r + 1] = action[1]; m.rq_action[3 * r + 2] = action[2];
                    // the step's path argument = this node's device path; a stored child's = that + its new device node (the node was
                    // expanded in an earlier wave of simulations, so its path is final; nobody reads the child's before the next wave)
                    int depth = 0;
#pragma unroll
                    for (int s = 0; s < kMctsPath; ++s) {
                        m.ts_paths[kMctsPath * r + s] = pth[s];
                        depth += pth[s] >= 0 ? 1 : 0;
                    }
                    if (newdev >= 0) {
#pragma unroll
                        for (int s = 0; s < kMctsPath; ++s) m.n_devpath[(size_t)kMctsPath * child + s] = (s == depth) ? newdev : pth[s];
                    }
                }
                const size_t ps = ((size_t)w * m.roots + j) * m.max_depth + plen;
                m.p_node[ps] = cur;
                m.p_k[ps] = k;
                m.p_cost[ps] = cost;
                if (virt) {
                    m.t_nsa[row + k] = best_nsa + 1;
                    m.n_ns[cur] = ns + 1;
                }
            }
            MC_STAMP(6);
#if IPP_MCTS_CLOCKS
            mlev_ += 1;
#endif
            plen += 1;
            budget -= cost;
            prev[0] = action[0]; prev[1] = action[1]; prev[2] = action[2];
            cur = child;
            fl = cfl; K = cK; ns = cns;  // (the expanded bit, K and Ns of the child are not written by this level)
        }
        if (lane == 0) {
            m.p_len[w * m.roots + j] = plen;
            m.leaf[w * m.roots + j] = leafnode;
        }
        // lane 0's stores (virtual visits, children, flags) before the next descent's loads: the SAME wave reads them, through the
        // same L1, so workgroup scope is enough (wait for the stores; no cache action) -- __threadfence() is an agent-scope
        // release, which writes the L2 back: 11 of the 43 us of a descent (lane-0 clocks of the -DIPP_MCTS_CLOCKS build).
        // INVARIANT this rests on: every table a root touches (n_*, t_*, the hash table, root_count: all indexed by the root j) is
        // private to ONE wave for the whole launch.  A table shared between roots or a second wave per root needs the agent-scope
        // fence back: -DIPP_MCTS_AGENT_FENCE=1 restores it (A/B, debugging).
#if defined(IPP_MCTS_AGENT_FENCE) && IPP_MCTS_AGENT_FENCE
        __threadfence();
#else
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#endif
        MC_STAMP(7);
    }
#if IPP_MCTS_CLOCKS
    if (lane == 0) { for (int q = 0; q < 8; ++q) atomicAdd(&g_mclk[q], mclk_[q]); atomicAdd(&g_mclk[8], mlev_); atomicAdd(&g_mclk[9], (unsigned long long)W); }
#endif
}

// Results of a level's ipp_tree_step for engines whose tree kernels do not write them themselves (band-tile nodes; k_tree_patch
// does: TreeEdgeOut): the edge's masked trace reduction (reward (cost + 1), rewards.py:31 undone: the cost depends on the path that
// led to the node, the reduction does not).
__global__ void k_mcts_apply(ipp_mcts_tables m, int first, int n) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t r = (size_t)first + i;
    if (m.ts_status[r] != 0) m.err[2] = m.ts_status[r];
    m.t_num[(size_t)m.rq_parent[r] * m.kmax + m.rq_k[r]] = (double)m.ts_reward[r] * (m.rq_cost[r] + 1.0);
}

// ---------------------------------------------------------------------------------------------------- expansion
// Marsaglia-Tsang gamma(shape, 1) from a counter-based stream (shape < 1 through the u^(1/shape) boost)
__device__ inline double mc_gamma(double shape, uint64_t key) {
    double boost = 1.0;
    uint64_t c = 0;
    if (shape < 1.0) {
        boost = pow(mc_u01(mc_mix(key ^ 0xA5A5A5A5ull)), 1.0 / shape);
        shape += 1.0;
    }
    const double dd = shape - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * dd);
    for (int it = 0; it < 64; ++it) {
        const double u1 = mc_u01(mc_mix(key + (++c))), u2 = mc_u01(mc_mix(key + (++c))), u3 = mc_u01(mc_mix(key + (++c)));
        const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);  // Box-Muller
        const double t = 1.0 + cc * x;
        if (t <= 0.0) continue;
        const double v = t * t * t;
        if (log(u3) < 0.5 * x * x + dd - dd * v + dd * log(v)) return boost * dd * v;
    }
    return boost * dd;
}

// One wave per pending leaf (slot s of root j): the valid-action set in ascending action index (mcts.py:148-158: every
// action farther than 0, within the remaining budget and closer than max_valid_action_distance; only the cells within
// that distance are looked at), priors (uniform, or the network's on the valid set), Dirichlet noise for the root of
// the first simulation, node flags and value.
// prior: [roots * wave][kmax] network priors on the leaf's valid set (slot order) or NULL (uniform, mcts.py:204);
// value: [roots * wave] or NULL (value_const).
__global__ __launch_bounds__(256) void k_mcts_expand(ipp_mcts_tables m, const double* __restrict__ prior,
                                                     const double* __restrict__ value, double value_const, int sets_only,
                                                     double alpha, double eps, uint64_t seed) {
#pragma clang fp contract(off)
    const int g = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (g >= m.roots * m.wave) return;
    const int j = g / m.wave, s = g - j * m.wave;
    if (s >= m.pend_count[j]) return;
    const int nd = m.pend_node[g];
    const size_t row = (size_t)nd * m.kmax;
    const double pos[3] = {m.pend_prev[3 * g], m.pend_prev[3 * g + 1], m.pend_prev[3 * g + 2]};
    const double budget = m.pend_budget[g];
    int K = 0;
    if (sets_only || !(m.n_flags[nd] & 4)) {
        const int px = (int)floor(pos[0] / m.res), py = (int)floor(pos[1] / m.res);
        const int ncell = m.grid_w * m.grid_h;
        const int n_cand = m.n_levels * m.n_off;
        for (int c0 = 0; c0 < n_cand; c0 += 64) {
            const int c = c0 + lane;
            bool ok = false;
            int idx = -1;
            if (c < n_cand) {
                const int lv = c / m.n_off, o = c - lv * m.n_off;
                const int cx = px + m.off_x[o], cy = py + m.off_y[o];
                if (cx >= 0 && cx < m.grid_w && cy >= 0 && cy < m.grid_h) {
                    idx = lv * ncell + m.cell_action[cx * m.grid_h + cy];
                    const double dist = mc_distance(m.actions + 3 * (size_t)idx, pos);
                    ok = dist > 0 && dist <= budget && dist < m.max_dist;
                }
            }
            const unsigned long long bal = __ballot(ok);
            if (ok) {
                const int at = K + __popcll(bal & ((1ull << lane) - 1ull));
                if (at < m.kmax) m.t_idx[row + at] = idx;
            }
            K += __popcll(bal);
        }
        if (K > m.kmax) { if (lane == 0) m.err[3] = 1; K = m.kmax; }
        for (int k = K + lane; k < m.kmax; k += 64) m.t_idx[row + k] = -1;
        if (lane == 0) m.n_k[nd] = K;
    } else {
        K = m.n_k[nd];
    }
    if (sets_only) {  // (the host asks the network with the sets, then calls again)
        if (lane == 0) m.n_flags[nd] |= 4;
        return;
    }
    if (K == 0) {  // mcts.py:201-202: stays a leaf, value 0 (its set is computed again when it is reached again)
        if (lane == 0) m.n_flags[nd] &= (unsigned char)~4;
        return;
    }
    // ---- priors on the valid set
    const bool noise = m.pend_depth[g] == 0 && m.pend_sim[g] == 0;
    double total = 0.0;
    if (prior) {
        for (int k = lane; k < K; k += 64) total += prior[(size_t)g * m.kmax + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) total += mc_shfl_xor(total, o);
    }
    const double uni = m.uniform_ps[K];  // (1/A) / sum of K copies of 1/A, summed like NumPy does (host table)
    double gsum = 0.0, rest = 0.0;
    const uint64_t nkey = mc_mix(seed ^ ((uint64_t)(uint32_t)(j + m.root_base) << 24));  // (the root's number in the whole search: ipp_mcts_tables.root_base)
    if (noise) {
        // Dirichlet(alpha) over all A actions, looked at on the K valid ones: independent Gamma(alpha) draws for those, one
        // Gamma((A - K) alpha) draw for the total of the rest (aggregation property); the reference normalises the noisy
        // vector over ALL actions, the mass that lands on invalid actions stays there (mcts.py:160-164, 222-225)
        for (int k = lane; k < K; k += 64) gsum += mc_gamma(alpha, nkey + ((uint64_t)(k + 1) << 32));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gsum += mc_shfl_xor(gsum, o);
        rest = (m.num_actions > K) ? mc_first(mc_gamma(alpha * (m.num_actions - K), nkey)) : 0.0;
    }
    for (int k = lane; k < m.kmax; k += 64) {
        double ps = 0.0;
        if (k < K) {
            if (!noise) {
                ps = prior ? ((total > 0) ? prior[(size_t)g * m.kmax + k] / total : 1.0 / K) : uni;
            } else {
                const double p0 = prior ? prior[(size_t)g * m.kmax + k] : 1.0 / m.num_actions;
                const double psum = prior ? total : (double)K / m.num_actions;
                const double gk = mc_gamma(alpha, nkey + ((uint64_t)(k + 1) << 32));
                ps = ((1 - eps) * p0 + eps * gk / (gsum + rest)) / ((1 - eps) * psum + eps);
            }
        }
        m.t_ps[row + k] = ps;
        m.t_nsa[row + k] = 0.0;
        m.t_qsa[row + k] = 0.0;
        m.t_num[row + k] = NAN;
        m.t_child[row + k] = -1;
    }
    if (lane == 0) {
        m.n_ns[nd] = 0.0;
        m.n_value[nd] = value ? value[g] : value_const;
        m.n_flags[nd] = (m.n_flags[nd] & ~4) | kNodeExpanded;
    }
}

// ---------------------------------------------------------------------------------------------------- backup
// The W recorded descents of a root, deepest step first (mcts.py:255-265; virtual visits taken back first).
// k_mcts_backup_serial: one thread per root walks them one after the other -- every step a chain of dependent round trips (66 us
// per wave of 8 simulations at configs[4]).  k_mcts_backup (W x max_depth <= 64): one WAVE per root, one lane per recorded step in
// processing order (lane = w D + pos, step = plen_w - 1 - pos): every lane loads its step and its edge at once; the values
// val = reward + gamma value' are a suffix scan along each descent (they do not depend on the tables); the updates of an edge that
// several descents went through are applied descent by descent on a copy in LDS held at the edge's first lane -- the same operations
// on the same operands in the same order as the serial walk, so the tables come out bit-identical (tests/test_hip_mcts.py).
// Both clear the per-wave counters (pending leaves, requests per level) for the next wave of simulations.
__global__ void k_mcts_backup_serial(ipp_mcts_tables m, int W) {
#pragma clang fp contract(off)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) m.rq_count[0] = 0;
    if (j >= m.roots) return;
    m.pend_count[j] = 0;
    const bool virt = W > 1;
    for (int w = 0; w < W; ++w) {
        const int lf = m.leaf[w * m.roots + j];
        double value = lf >= 0 ? m.n_value[lf] : 0.0;
        const int plen = m.p_len[w * m.roots + j];
        for (int stp = plen - 1; stp >= 0; --stp) {
            const size_t ps = ((size_t)w * m.roots + j) * m.max_depth + stp;
            const int node = m.p_node[ps], k = m.p_k[ps];
            const double cost = m.p_cost[ps];
            const size_t e = (size_t)node * m.kmax + k;
            double nsa = m.t_nsa[e];
            if (virt) {
                nsa -= 1;
                m.n_ns[node] -= 1;
            }
            const double reward = m.t_num[e] / (cost + 1.0);  // rewards.py:31
            const double val = reward + m.gamma * value;
            m.t_qsa[e] = (nsa > 0) ? (nsa * m.t_qsa[e] + val) / (nsa + 1) : val;
            m.t_nsa[e] = nsa + 1;
            m.n_ns[node] += 1;
            value = val;
        }
    }
}

__global__ __launch_bounds__(64) void k_mcts_backup(ipp_mcts_tables m, int W) {
#pragma clang fp contract(off)
    __shared__ double st_nsa[64], st_q[64];
    const int j = blockIdx.x, lane = threadIdx.x;
    const int D = m.max_depth;
    if (j == 0 && lane == 0) m.rq_count[0] = 0;
    if (lane == 0) m.pend_count[j] = 0;
    const bool virt = W > 1;
    const int w = lane / D, pos = lane - w * D;
    const bool in_w = w < W;
    const int plen = in_w ? m.p_len[w * m.roots + j] : 0;
    const int lf = in_w ? m.leaf[w * m.roots + j] : -1;
    const bool valid = pos < plen;
    const int stp = plen - 1 - pos;
    int node = -1, k = -1;
    double cost = 0.0;
    if (valid) {
        const size_t ps = ((size_t)w * m.roots + j) * D + stp;
        node = m.p_node[ps]; k = m.p_k[ps]; cost = m.p_cost[ps];
    }
    double carry = lf >= 0 ? m.n_value[lf] : 0.0;  // (the leaf's value, on every lane of the descent)
    const size_t e = valid ? (size_t)node * m.kmax + k : 0;
    double nsa0 = 0.0, q0 = 0.0, num = 0.0, nns = 0.0;
    if (valid) { nsa0 = m.t_nsa[e]; q0 = m.t_qsa[e]; num = m.t_num[e]; if (!virt) nns = m.n_ns[node]; }
    // ---- values along each descent: position 0 is the deepest step
    const double reward = num / (cost + 1.0);  // rewards.py:31
    double val = 0.0;
    for (int p = 0; p < D; ++p) {
        if (pos == p) val = reward + m.gamma * carry;
        const int src = min(w * D + p, 63);
        const double vp = __shfl(val, src, 64);
        const int okp = __shfl((int)valid, src, 64);
        if (okp) carry = vp;
    }
    // ---- the first lane (in processing order) that holds the same edge: edges sit at one depth of the tree, so only the lanes of
    // the same step index in the earlier descents can match
    int first = lane;
    for (int wp = 0; wp < W; ++wp) {
        const int plen_p = __shfl(plen, min(wp * D, 63), 64);
        const int src = wp * D + (plen_p - 1 - stp);
        const bool cand = valid && wp < w && stp < plen_p && stp >= 0;
        const int s2 = cand ? src : lane;
        const int node_p = __shfl(node, s2, 64), k_p = __shfl(k, s2, 64);
        if (cand && first == lane && node_p == node && k_p == k) first = src;
    }
    st_nsa[lane] = nsa0;
    st_q[lane] = q0;
    __syncthreads();
    // ---- the updates, descent by descent (no two steps of one descent share an edge)
    for (int wp = 0; wp < W; ++wp) {
        if (valid && w == wp) {
            double nsa = st_nsa[first];
            const double q = st_q[first];
            if (virt) nsa -= 1;
            st_q[first] = (nsa > 0) ? (nsa * q + val) / (nsa + 1) : val;
            st_nsa[first] = nsa + 1;
        }
        __syncthreads();
    }
    if (valid && first == lane) { m.t_qsa[e] = st_q[lane]; m.t_nsa[e] = st_nsa[lane]; }
    if (valid && !virt) m.n_ns[node] = nns + 1;  // (W == 1: one descent, every node once; with virtual visits: - 1 + 1, unchanged)
}

// ---------------------------------------------------------------------------------------------------- policy read-out
// One wave per root: the search policy from the root's visit counts (mcts.py:83-143 for temperature > 0).  Training time
// (deploy_time == 0): forced playouts taken back (:109-131) -- among the most visited actions one is kept (tie_u[j] picks it
// like rng.choice picks among the ties; NULL: the first), every other action gives back one forced playout at a time while its
// PUCT score with the reduced count stays below the kept action's, and single remaining visits are dropped; then
// visits^(1/T) / sum.  Same arithmetic and operand order as the NumPy read-out (device_mcts.py::_policies_rows); every action's
// take-back loop is independent of the others', so the lanes run theirs side by side.
// policy [R][kmax] (0 on the padding), idx_out [R][kmax] = the root's valid action indices (optional), ok [R] = 1 where the root
// was expanded and kept visits (the reference returns None otherwise).
__global__ __launch_bounds__(256) void k_mcts_policy(ipp_mcts_tables m, const double* __restrict__ tie_u, double inv_temperature, int deploy_time,
                                                     double* __restrict__ policy, int32_t* __restrict__ idx_out, int32_t* __restrict__ ok) {
#pragma clang fp contract(off)
    const int j = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (j >= m.roots) return;
    const int node = j * m.nodes_per_root, kmax = m.kmax;
    const size_t row = (size_t)node * kmax;
    double* pj = policy + (size_t)j * kmax;
    const int K = m.n_k[node];
    const double ns = m.n_ns[node];
    const bool ok_root = (m.n_flags[node] & kNodeExpanded) && K > 0;
    if (idx_out)
        for (int k = lane; k < kmax; k += 64) idx_out[(size_t)j * kmax + k] = m.t_idx[row + k];
    if (!ok_root) {
        for (int k = lane; k < kmax; k += 64) pj[k] = 0.0;
        if (lane == 0) ok[j] = 0;
        return;
    }
    auto visits_of = [&](int k) { return (m.t_idx[row + k] >= 0) ? m.t_nsa[row + k] : 0.0; };
    double max_puct = -INFINITY, lo = INFINITY, hi = -INFINITY, pc = 0.0;
    const double sq = sqrt(ns + 1);
    bool allzero = true;
    int best = -1;
    if (!deploy_time) {
        double vmax = -INFINITY;
        int nz = 0;
        for (int k = lane; k < kmax; k += 64) {
            vmax = fmax(vmax, visits_of(k));
            if (m.t_idx[row + k] >= 0) {
                const double q = m.t_qsa[row + k];
                lo = fmin(lo, q); hi = fmax(hi, q); nz |= (q != 0.0) ? 1 : 0;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            vmax = fmax(vmax, mc_shfl_xor(vmax, o));
            lo = fmin(lo, mc_shfl_xor(lo, o));
            hi = fmax(hi, mc_shfl_xor(hi, o));
            nz |= __shfl_xor(nz, o, 64);
        }
        if (K < m.num_actions) { lo = fmin(lo, 0.0); hi = fmax(hi, 0.0); }  // (mcts.py:267-278, as in k_mcts_select)
        allzero = nz == 0;
        pc = m.puct_init + log((ns + m.puct_base + 1) / m.puct_base);
        // the kept action: the pick-th (ascending action order) of the most visited
        int n_ties = 0;
        for (int k0 = 0; k0 < kmax; k0 += 64) {
            const int k = k0 + lane;
            n_ties += __popcll(__ballot(k < kmax && k < K && visits_of(k) == vmax));
        }
        if (n_ties > 0 && (vmax > 0 || m.num_actions == K)) {
            long long pick = tie_u ? (long long)(tie_u[j] * (double)n_ties) : 0;
            pick = pick < n_ties - 1 ? pick : n_ties - 1;
            int seen = 0;
            for (int k0 = 0; k0 < kmax && best < 0; k0 += 64) {
                const int k = k0 + lane;
                unsigned long long b = __ballot(k < kmax && k < K && visits_of(k) == vmax);
                const int c = __popcll(b);
                if (pick < seen + c) {
                    for (int r = (int)pick - seen; r > 0; --r) b &= b - 1;  // drop the r lowest ties
                    best = k0 + (int)__ffsll((long long)b) - 1;
                }
                seen += c;
            }
            const double q = m.t_qsa[row + best], nsa = m.t_nsa[row + best], ps = m.t_ps[row + best];
            const double qn = allzero ? q : ((lo == hi) ? q / hi : (q - lo) / (hi - lo));
            max_puct = (m.t_idx[row + best] >= 0) ? qn + pc * (ps * (sq / (1 + nsa))) : -INFINITY;
        }
    }
    double sum_vt = 0.0, tot = 0.0;
    for (int k = lane; k < kmax; k += 64) {
        const bool valid = m.t_idx[row + k] >= 0;
        double visits = valid ? m.t_nsa[row + k] : 0.0;
        if (!deploy_time) {
            if (valid && k != best) {
                const double ps = m.t_ps[row + k], q = m.t_qsa[row + k];
                double left = (visits == 0) ? 0.0 : ceil(sqrt(m.fpf * ps * ns));
                const double qn = allzero ? q : ((lo == hi) ? q / hi : (q - lo) / (hi - lo));
                while (left > 0) {
                    const double prior = pc * (ps * (sq / (1 + (visits - 1))));
                    if (qn + prior >= max_puct) break;  // (the playout is not taken back: the action leaves the loop)
                    visits -= 1;
                    left -= 1;
                }
            }
            if (visits == 1) visits = 0;
        }
        const double vt = (inv_temperature == 1.0) ? visits : pow(visits, inv_temperature);
        pj[k] = vt;
        sum_vt += vt;
        tot += visits;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sum_vt += mc_shfl_xor(sum_vt, o); tot += mc_shfl_xor(tot, o); }
    for (int k = lane; k < kmax; k += 64) pj[k] = (tot > 0) ? pj[k] / sum_vt : 0.0;  // (each lane re-reads what it wrote)
    if (lane == 0) ok[j] = tot > 0 ? 1 : 0;
}

}  // namespace ipp
