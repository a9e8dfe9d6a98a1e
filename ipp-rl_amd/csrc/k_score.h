// Candidate-action scoring from ONE state: reward of A actions, nothing written.
// Replaces the per-candidate simulate_prediction_step calls of greedy_search / the rollout policy
// (planning/common/optimization.py:33-104, planning/mcts_mission.py:232-246).
//
// Every candidate's reward is  sum_{i in mask} |Wc_i|^2 / (cost + 1)  with  Wc = P[:,F] H^T L^-1,  S = L^T L.
// The numerator is  tr(L^-T H (P M P)[F,F] H^T L^-1) = tr(S^-1 H G[F,F] H^T)  with  G = P M P,  M = diag(mask):
// G depends only on the state and the mask, not on the candidate, and a candidate needs just the f x f block of
// G on its footprint.  Footprints are small rectangles, so only the entries G[i][j] with |row_i - row_j| <= Dy,
// |col_i - col_j| <= Dx are ever read: a band of (Dy+1)(2Dx+1) values per cell.  Instead of streaming the state
// once per candidate (25 000 x 1.2 MB for a 50x50 grid with 10 altitude levels) the state is read once:
//   k_score_hdr      footprint / cost / noise of every candidate, largest footprint extent (Dy, Dx)
//   k_score_densify  factor state: P = P0 - U U^T into scratch (dense state: P is read in place)
//   k_score_band     G band in fp64: workgroup = (grid row y, dy), LDS tiles of the two P row blocks over k
//   k_score_eval     one wave per candidate: S = H P_FF H^T + R, T = H G_FF H^T, Cholesky, tr(S^-1 T) / (cost + 1)
#pragma once
#include "ipp_common.h"

namespace ipp {

constexpr int kScoreDCap = 9;                                      // footprints up to 10 x 10 cells
constexpr int kScoreBandCap = (kScoreDCap + 1) * (2 * kScoreDCap + 1);  // doubles per cell
constexpr int kScoreSplit = 4;                                     // k-range splits of the band kernel (partials summed in order)

struct ScoreHdr {
    int xl, yu, w, h;
    int rf, m, f, status;
    double cost, nv;
};

struct ScoreView {
    ScoreHdr* hdr;      // [max_batch]
    int* extent;        // [2] max (h - 1, w - 1) over the valid candidates
    float* mask;        // [Npad] 1 / 0
    float* P;           // [N][Npad] (scratch for factor engines, the env slot for dense engines)
    double* G;          // [kScoreSplit][N][kScoreBandCap] partial sums over k ranges; entry (dy, dx) at dy * (2 Dx + 1) + dx + Dx, dy >= 0
};

struct PrevAction { double p[3]; };

// ---------------------------------------------------------------- candidate headers (fp64 like NumPy)
template <int MC>
__global__ __launch_bounds__(256) void k_score_hdr(View v, ScoreView sv, int env, const double* __restrict__ actions, int A,
                                                   PrevAction prev, unsigned flags, float* __restrict__ reward,
                                                   int* __restrict__ status_out, const float* __restrict__ diag_src) {
    constexpr int FC = 4 * MC;
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    // adaptive mask of the state (planning/common/rewards.py:8-12), one pass by the first blocks
    for (int i = a; i < v.Npad; i += gridDim.x * blockDim.x) {
        float mk = 0.f;
        if (i < v.N)
            mk = (!(flags & IPP_ADAPTIVE) ||
                  ((double)v.mean[(size_t)env * v.Npad + i] + v.kf * (double)(diag_src ? diag_src[i] : v.diag[(size_t)env * v.Npad + i]) >= v.thr)) ? 1.f : 0.f;
        sv.mask[i] = mk;
    }
    if (a >= A) return;
    const double ax = actions[3 * a + 0], ay = actions[3 * a + 1], az = actions[3 * a + 2];
    ScoreHdr h;
    bool ok = isfinite(ax) && isfinite(ay) && isfinite(az);
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
    h.xl = xl; h.yu = yu; h.w = xr - xl + 1; h.h = yd - yu + 1;
    h.rf = (az > v.rf_alt) ? 2 : 1;                                               // cameras.py:125
    const int nx = (h.w - 1) / h.rf + 1, ny = (h.h - 1) / h.rf + 1;               // sensor_models.py:57
    h.m = nx * ny;
    h.f = h.w * h.h;
    h.nv = v.coeff_a * (1.0 - exp(-v.coeff_b * az));                              // sensor_models.py:30
    const double dx = ax - prev.p[0], dy = ay - prev.p[1], dz = az - prev.p[2];
    const double dist = sqrt(dx * dx + dy * dy + dz * dz);                        // actions.py:15-16
    double cost = dist;
    if (flags & IPP_USE_FLIGHT_TIME) {                                            // actions.py:32-41
        const double d_acc = fmin(dist * 0.5, v.vmax * v.vmax / (2 * v.amax));
        cost = (dist - 2 * d_acc) / v.vmax + 2 * sqrt(2 * d_acc / v.amax);
    }
    h.cost = cost;
    h.status = IPP_STATUS_OK;
    if (!ok || h.m > MC || h.f > FC || h.h - 1 > kScoreDCap || h.w - 1 > kScoreDCap) h.status = IPP_STATUS_BAD_FOOTPRINT;
    if (h.status == IPP_STATUS_OK) {
        atomicMax(&sv.extent[0], h.h - 1);
        atomicMax(&sv.extent[1], h.w - 1);
    } else {
        reward[a] = 0.f;
    }
    if (status_out) status_out[a] = h.status;
    sv.hdr[a] = h;
}

// ---------------------------------------------------------------- factor state -> dense P (fp32), 64 x 64 tiles
// P[i][j] = P0(i, j) - sum_k U[k][i] U[k][j] over the columns stored on both cells' tiles.
// CHAIN: the state of a tree node (k_tree.h): the root env's columns followed by the column blocks of the path.
struct ScorePath { int depth; int ids[kTreeDepth]; };

template <bool CHAIN>
__global__ __launch_bounds__(256) void k_score_densify(View v, ScoreView sv, int env, ScorePath path, const float* __restrict__ node_cov,
                                                       const int* __restrict__ node_meta, int win_cells) {
    constexpr int TB = 64, KC = 32;
    __shared__ float ui[KC][TB + 1], uj[KC][TB + 1];
    const int i0 = blockIdx.y * TB, j0 = blockIdx.x * TB;
    const int tid = threadIdx.x, ti = tid / 16, tj = tid % 16;  // thread owns rows ti*4.., cols tj*4..
    const float* U = v.cov + (size_t)env * v.cov_slot;
    int r = v.rank[env];
    const int* span = v.colspan + (size_t)env * v.rank_cap;
    ChainCols cc;
    if (CHAIN) {
        cc.root = U; cc.root_spans = span; cc.root_rects = v.colrect + (size_t)env * v.rank_cap; cc.r_root = r; cc.depth = 0; cc.npad = (size_t)v.Npad; cc.nstride = (size_t)win_cells;
#pragma unroll
        for (int j = 0; j < kTreeDepth; ++j) { cc.node[j] = U; cc.off[j] = 0x7fffffff; cc.nspan[j] = 0; cc.nrect[j] = kRectFull; }
#pragma unroll
        for (int j = 0; j < kTreeDepth; ++j)
            if (j < path.depth) {
                const int id = path.ids[j];
                cc.nspan[j] = node_meta[kNodeMeta * id + 1];
                cc.nrect[j] = (unsigned)node_meta[kNodeMeta * id + 4];
                cc.node[j] = node_cov + (size_t)id * v.meas_cap * win_cells - (size_t)(cc.nspan[j] & 0xffff) * v.tile_cells;
                cc.off[j] = r;
                r += node_meta[kNodeMeta * id];
            }
        cc.depth = path.depth;
    }
    const int tile_i = i0 / v.tile_cells, tile_j = j0 / v.tile_cells;  // 64 divides tile_cells: a block sits on one tile
    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.f;
    for (int k0 = 0; k0 < r; k0 += KC) {
        __syncthreads();
        for (int idx = tid; idx < KC * TB; idx += 256) {
            const int kk = idx / TB, c = idx - kk * TB;
            const int k = k0 + kk;
            float a = 0.f, b = 0.f;
            if (k < r && v.patch) {
                // patch layout (k_step_patch.h / k_tree_patch.h): column k is the patch of its rectangle -- the root env's in its slot,
                // a path node's in node_cov[id][j]
                unsigned rc = (unsigned)v.colrect[(size_t)env * v.rank_cap + min(k, v.rank_cap - 1)];
                const float* pk = U + (size_t)k * v.pstride;
                if (CHAIN) {
#pragma unroll
                    for (int d = 0; d < kTreeDepth; ++d)
                        if (d < path.depth && k >= cc.off[d]) {
                            rc = cc.nrect[d];
                            pk = node_cov + ((size_t)path.ids[d] * v.meas_cap + (k - cc.off[d])) * v.pstride;
                        }
                }
                const int r0 = rc & 0xff, c0 = (rc >> 16) & 0xff;
                const int ic = min(i0 + c, v.N - 1), jc = min(j0 + c, v.N - 1);
                const int ri = ic / v.W, ci = ic - ri * v.W, rj = jc / v.W, cj = jc - rj * v.W;
                if (i0 + c < v.N && rect_has(rc, ri, ci)) a = pk[(ri - r0) * v.pw + (ci - c0)];
                if (j0 + c < v.N && rect_has(rc, rj, cj)) b = pk[(rj - r0) * v.pw + (cj - c0)];
            } else if (k < r) {
                const int sp = CHAIN ? cc.span(k) : span[k], lo = sp & 0xffff, hi = sp >> 16;
                const float* rowk = CHAIN ? cc.row(k) : U + (size_t)k * v.Npad;
                unsigned rc = kRectFull;  // (View::rect_meta: nothing is stored outside a column's rectangle)
                if (v.rect_meta) rc = CHAIN ? cc.rect(k) : (unsigned)v.colrect[(size_t)env * v.rank_cap + k];
                const int ic = min(i0 + c, v.N - 1), jc = min(j0 + c, v.N - 1);
                if (tile_i >= lo && tile_i <= hi && i0 + c < v.N && rect_has(rc, ic / v.W, ic % v.W)) a = rowk[i0 + c];
                if (tile_j >= lo && tile_j <= hi && j0 + c < v.N && rect_has(rc, jc / v.W, jc % v.W)) b = rowk[j0 + c];
            }
            ui[kk][c] = a;
            uj[kk][c] = b;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < KC; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) av[a] = ui[kk][ti * 4 + a];
#pragma unroll
            for (int b = 0; b < 4; ++b) bv[b] = uj[kk][tj * 4 + b];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = fmaf(av[a], bv[b], acc[a][b]);
        }
    }
    const double svv = v.prior[2 * env + 0], ls = v.prior[2 * env + 1];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int i = i0 + ti * 4 + a;
        if (i >= v.N) continue;
        const int ri = i / v.W, ci = i - ri * v.W;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int j = j0 + tj * 4 + b;
            if (j >= v.Npad) continue;
            float out = 0.f;
            if (j < v.N) {
                const int rj = j / v.W, cj = j - rj * v.W;
                out = (float)(matern_d(ri - rj, ci - cj, v.res, svv, ls) - (double)acc[a][b]);
            }
            sv.P[(size_t)i * v.Npad + j] = out;
        }
    }
}

// ---------------------------------------------------------------- G band: G[i][j] = sum_k mask_k P[i][k] P[j][k]
// Workgroup (y, dy, z): cells i of grid row y against cells j of grid row y + dy, |col_i - col_j| <= Dx, over the
// z-th quarter of the k range, fp64.  LDS: mask-weighted P rows of the i block and P rows of the j block, converted
// to fp64 once, KC columns k at a time.  The kScoreSplit partial bands are added in order by k_score_eval.
__global__ __launch_bounds__(256) void k_score_band(View v, ScoreView sv, int kc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_band[];
    const int W = v.W;
    const int ld = kc + 1;
    double* pi = reinterpret_cast<double*>(smem_band);  // [W][kc + 1]
    double* pj = pi + (size_t)W * ld;                    // [W][kc + 1]
    const int Dy = sv.extent[0], Dx = sv.extent[1];
    const int y = blockIdx.x, dy = blockIdx.y, z = blockIdx.z;
    if (dy > Dy || y + dy >= v.H) return;
    const int nb = 2 * Dx + 1;
    const int n_out = W * nb;                          // (ci, dx) pairs
    const int tid = threadIdx.x;
    const int k_per = ((v.N + kScoreSplit - 1) / kScoreSplit + kc - 1) / kc * kc;
    const int k_lo = z * k_per, k_hi = min(v.N, k_lo + k_per);
    double* Gz = sv.G + (size_t)z * v.N * kScoreBandCap;
    constexpr int OPT = 8;                             // outputs per thread and pass
    const float* Pi = sv.P + (size_t)(y * W) * v.Npad;
    const float* Pj = sv.P + (size_t)((y + dy) * W) * v.Npad;
    for (int base = 0; base < n_out; base += 256 * OPT) {  // one pass unless W (2 Dx + 1) > 2048
        double acc[OPT];
        int oci[OPT], ocj[OPT];
#pragma unroll
        for (int o = 0; o < OPT; ++o) {
            const int idx = base + tid + o * 256;
            acc[o] = 0.0;
            oci[o] = -1; ocj[o] = 0;
            if (idx < n_out) {
                const int ci = idx / nb, dx = idx - ci * nb - Dx;
                if (ci + dx >= 0 && ci + dx < W) { oci[o] = ci; ocj[o] = ci + dx; }
            }
        }
        for (int k0 = k_lo; k0 < k_hi; k0 += kc) {
            __syncthreads();
            for (int idx = tid; idx < W * kc; idx += 256) {
                const int c = idx / kc, kk = idx - c * kc;
                const int k = k0 + kk;
                const bool in = k < k_hi;
                pi[c * ld + kk] = in ? (double)(sv.mask[k] * Pi[(size_t)c * v.Npad + k]) : 0.0;
                pj[c * ld + kk] = in ? (double)Pj[(size_t)c * v.Npad + k] : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int o = 0; o < OPT; ++o) {
                if (oci[o] < 0) continue;
                const double* a = pi + oci[o] * ld;
                const double* b = pj + ocj[o] * ld;
                double s = acc[o];
#pragma unroll 8
                for (int kk = 0; kk < kc; ++kk) s = fma(a[kk], b[kk], s);
                acc[o] = s;
            }
        }
#pragma unroll
        for (int o = 0; o < OPT; ++o) {
            if (oci[o] < 0) continue;
            const int i = y * W + oci[o];
            Gz[(size_t)i * kScoreBandCap + dy * nb + (ocj[o] - oci[o]) + Dx] = acc[o];
        }
    }
}

// ---------------------------------------------------------------- per-candidate evaluation, one wave each
template <int MC>
__global__ __launch_bounds__(256) void k_score_eval(View v, ScoreView sv, int A, float* __restrict__ reward) {
    constexpr int LD = MC + 1, WPB = 4;
    __shared__ double Ss[WPB][MC * LD], Ts[WPB][MC * LD], Ls[WPB][MC * LD], Xs[WPB][MC * LD];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int a = blockIdx.x * WPB + wave;
    if (a >= A) return;
    const ScoreHdr h = sv.hdr[a];
    if (h.status != IPP_STATUS_OK) return;  // reward already 0
    const int Dx = sv.extent[1], nb = 2 * Dx + 1;
    const int m = h.m, nx = (h.w - 1) / h.rf + 1;
    double* S = Ss[wave]; double* T = Ts[wave]; double* L = Ls[wave]; double* X = Xs[wave];
    const double R = (double)(h.rf * h.rf * h.rf) * h.nv;  // sensor_models.py:36

    // S[p][q] = sum_{a in blk p} sum_{b in blk q} w_p w_q P[a][b] (+ R), T likewise from the G band; pairs p <= q
    const int npairs = m * (m + 1) / 2;
    for (int pr = lane; pr < npairs; pr += kWave) {
        int q = (int)((sqrtf(8.0f * pr + 1.0f) - 1.0f) * 0.5f);
        while (q * (q + 1) / 2 > pr) --q;
        while ((q + 1) * (q + 2) / 2 <= pr) ++q;
        const int p = pr - q * (q + 1) / 2;
        const Block bp = block_of(p, nx, h.rf, h.w, h.h), bq = block_of(q, nx, h.rf, h.w, h.h);
        double s = 0.0, t = 0.0;
        for (int ia = 0; ia < bp.count(); ++ia) {
            const int ya = h.yu + bp.y0 + ia / bp.bw, xa = h.xl + bp.x0 + ia % bp.bw;
            const int ca = ya * v.W + xa;
            for (int ib = 0; ib < bq.count(); ++ib) {
                const int yb = h.yu + bq.y0 + ib / bq.bw, xb = h.xl + bq.x0 + ib % bq.bw;
                const int cb = yb * v.W + xb;
                s += (double)sv.P[(size_t)ca * v.Npad + cb];
                // band entry of the pair with the smaller row first (G is symmetric)
                const bool swap = (yb < ya);
                const int ci = swap ? cb : ca, ddy = swap ? ya - yb : yb - ya, ddx = swap ? xa - xb : xb - xa;
                const size_t gi = (size_t)ci * kScoreBandCap + ddy * nb + ddx + Dx;
#pragma unroll
                for (int zz = 0; zz < kScoreSplit; ++zz) t += sv.G[(size_t)zz * v.N * kScoreBandCap + gi];
            }
        }
        const double wpq = bp.weight * bq.weight;
        s *= wpq; t *= wpq;
        if (p == q) s += R;
        S[p * LD + q] = s; S[q * LD + p] = s;
        T[p * LD + q] = t; T[q * LD + p] = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // Cholesky S = C C^T (wave-synchronous), then X = S^-1 T column by column, reward = tr(X) / (cost + 1)
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
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
        if (lane > c && lane < m) {
            double sacc = S[lane * LD + c];
            for (int k = 0; k < c; ++k) sacc -= L[lane * LD + k] * L[c * LD + k];
            L[lane * LD + c] = sacc / L[c * LD + c];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    double tr = 0.0;
    if (pd && lane < m) {
        const int j = lane;  // column j of T: C z = T[:,j], C^T x = z
        for (int i = 0; i < m; ++i) {
            double s = T[i * LD + j];
            for (int k = 0; k < i; ++k) s -= L[i * LD + k] * X[k * LD + j];
            X[i * LD + j] = s / L[i * LD + i];
        }
        for (int i = m - 1; i >= 0; --i) {
            double s = X[i * LD + j];
            for (int k = i + 1; k < m; ++k) s -= L[k * LD + i] * X[k * LD + j];
            X[i * LD + j] = s / L[i * LD + i];
        }
        tr = X[j * LD + j];
    }
    tr = wave_sum(tr);
    if (lane == 0) reward[a] = pd ? (float)(tr / (h.cost + 1.0)) : NAN;  // rewards.py:31
}

}  // namespace ipp
