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

namespace ipp {

// One axis of OpenCV's INTER_AREA table (computeResizeAreaTab, opencv 4.5.2 resize.cpp): weights are
// evaluated in double and stored as float.  Returns the tap count (<= cap).
__device__ inline int area_taps(int src, int dst, int d, int* idx, double* wt, int cap) {
    const double scale = (double)src / (double)dst;
    const double a = d * scale, b = a + scale;
    const double cell = fmin(scale, (double)src - a);
    int s1 = (int)ceil(a);
    int s2 = min((int)floor(b), src - 1);
    s1 = min(s1, s2);
    int n = 0;
    if (s1 - a > 1e-3 && n < cap) { idx[n] = s1 - 1; wt[n++] = (double)(float)((s1 - a) / cell); }
    for (int s = s1; s < s2 && n < cap; ++s) { idx[n] = s; wt[n++] = (double)(float)(1.0 / cell); }
    if (b - s2 > 1e-3 && n < cap) { idx[n] = s2; wt[n++] = (double)(float)(fmin(fmin(b - s2, 1.0), cell) / cell); }
    return n;
}

template <int MC, int MODE>
__global__ __launch_bounds__(kPrepThreads) void k_prepare(View v, const int* __restrict__ env_ids,
                                                          const int* __restrict__ dst_ids, int n_items,
                                                          const double* __restrict__ action,
                                                          const double* __restrict__ prev_action,
                                                          const float* __restrict__ meas_noise, unsigned flags,
                                                          int* __restrict__ status_out, float* __restrict__ obs_out,
                                                          int* __restrict__ obs_m, int* __restrict__ obs_shape) {
    constexpr int FC = 4 * MC;
    constexpr int LD = MC + 1;  // padded leading dimension of the small fp64 matrices
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // ---- LDS carve (all offsets multiples of 16 B)
    double* S = reinterpret_cast<double*>(smem);          // [MC][LD]
    double* L = S + MC * LD;                              // [MC][LD]   lower Cholesky factor / work
    double* Li = L + MC * LD;                             // [MC][LD]   upper-triangular inverse (or S^-1)
    double* zz = Li + MC * LD;                            // [MC] observation
    double* vv = zz + MC;                                 // [MC] innovation
    double* yy = vv + MC;                                 // [MC]
    double* sub = yy + MC;                                // [FC] ground-truth crop
    ItemHdr* hs = reinterpret_cast<ItemHdr*>(sub + FC);
    int* okflag = reinterpret_cast<int*>(hs + 1);          // [4]
    float* big = reinterpret_cast<float*>(okflag + 4);     // factor: HT[MC][ht_ld]   dense: PFF[FC][FC+1]

    const int item = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wave = tid / kWave;
    constexpr int NW = kPrepThreads / kWave;
    if (item >= n_items) return;

    // ------------------------------------------------------------------ header (thread 0, fp64 like NumPy)
    if (tid == 0) {
        ItemHdr h;
        h.env = env_ids ? env_ids[item] : item;
        h.dst = dst_ids ? dst_ids[item] : h.env;
        h.status = IPP_STATUS_OK;
        h.fallback = 0;
        h.commit = (flags & IPP_PREDICT_ONLY) ? 0 : 1;
        h.pad0 = h.pad1 = 0;
        const double ax = action[3 * item + 0], ay = action[3 * item + 1], az = action[3 * item + 2];
        const double px = prev_action[3 * item + 0], py = prev_action[3 * item + 1], pz = prev_action[3 * item + 2];
        bool ok = isfinite(ax) && isfinite(ay) && isfinite(az) && h.env >= 0 && h.env < v.cap && h.dst >= 0 &&
                  h.dst < v.cap;
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
        h.nx = (h.w - 1) / h.rf + 1;                                                  // sensor_models.py:57
        h.ny = (h.h - 1) / h.rf + 1;
        h.m = h.nx * h.ny;                                                            // mappings.py:125-126
        h.f = h.w * h.h;
        h.nv_d = v.coeff_a * (1.0 - exp(-v.coeff_b * az));                            // sensor_models.py:30
        h.nv = (float)h.nv_d;
        const double dx = ax - px, dy = ay - py, dz = az - pz;
        const double dist = sqrt(dx * dx + dy * dy + dz * dz);                        // actions.py:15-16
        double cost = dist;
        if (flags & IPP_USE_FLIGHT_TIME) {                                            // actions.py:32-41
            const double d_acc = fmin(dist * 0.5, v.vmax * v.vmax / (2 * v.amax));
            cost = (dist - 2 * d_acc) / v.vmax + 2 * sqrt(2 * d_acc / v.amax);
        }
        h.cost_d = cost;
        h.cost = (float)cost;
        h.rank = 0;
        h.sv = h.ls = 0.f;
        if (ok) {
            h.rank = (MODE == IPP_FACTOR) ? v.rank[h.env] : 0;
            h.sv = (float)v.prior[2 * h.env + 0];
            h.ls = (float)v.prior[2 * h.env + 1];
        }
        if (!ok || h.m > MC || h.f > FC) h.status = IPP_STATUS_BAD_FOOTPRINT;
        if (h.status == IPP_STATUS_OK && h.rf > 1 && !(flags & IPP_COV_ONLY)) {
            // area resampler is only restated for shrinking scales
            const int ocols = (h.h + h.rf - 1) / h.rf, orows = (h.w + h.rf - 1) / h.rf;
            if (h.w < ocols || h.h < orows) h.status = IPP_STATUS_BAD_FOOTPRINT;
        }
        if (MODE == IPP_FACTOR && h.status == IPP_STATUS_OK && h.commit && h.rank + h.m > v.rank_cap) {
            h.status = IPP_STATUS_RANK_FULL;
            h.commit = 0;
        }
        h.rows = (MODE == IPP_FACTOR) ? h.rank : h.f;
        if (h.status == IPP_STATUS_BAD_FOOTPRINT) { h.m = 0; h.f = 0; h.rows = 0; h.commit = 0; }
        *hs = h;
        okflag[0] = 1;
    }
    __syncthreads();
    const ItemHdr h = *hs;
    const int m = h.m, f = h.f, r = h.rank;
    float* linv_out = v.linv + (size_t)item * MC * MC;
    float* y_out = v.yv + (size_t)item * MC;
    double* dbg = v.dbg + (size_t)item * (2 * MC * MC + 2 * MC);
    float* q_out = v.q + (size_t)item * v.q_rows * v.q_stride;

    if (m == 0) {  // bad footprint: nothing to stream
        if (tid == 0) {
            v.hdr[item] = h;
            if (status_out) status_out[item] = h.status;
            if (obs_m) obs_m[item] = 0;
        }
        for (int i = tid; i < MC * MC; i += kPrepThreads) linv_out[i] = 0.f;
        for (int i = tid; i < MC; i += kPrepThreads) y_out[i] = 0.f;
        return;
    }

    const double sv = v.prior[2 * h.env + 0], ls = v.prior[2 * h.env + 1];
    const float* mean_env = v.mean + (size_t)h.env * v.Npad;
    const float* gt_env = v.gt + (size_t)h.env * v.Npad;
    const float* cov_env = v.cov + (size_t)h.env * v.cov_slot;
    const double R = (double)(h.rf * h.rf * h.rf) * h.nv_d;  // sensor_models.py:36
    const bool cov_only = (flags & IPP_COV_ONLY) != 0;

    // ------------------------------------------------------------------ observation + innovation
    if (!cov_only) {
        for (int i = tid; i < f; i += kPrepThreads) {
            const int ly = i / h.w, lx = i - ly * h.w;
            sub[i] = (double)gt_env[(h.yu + ly) * v.W + h.xl + lx];  // simulations/__init__.py:24-25
        }
        __syncthreads();
        if (tid < m) {
            double val;
            if (h.rf == 1) {
                val = sub[tid];
            } else {
                // cv2.resize(sub, dsize=(ceil(h/rf), ceil(w/rf))) -> width=ceil(h/rf), height=ceil(w/rf)
                const int ocols = (h.h + h.rf - 1) / h.rf, orows = (h.w + h.rf - 1) / h.rf;
                const int orow = tid / ocols, ocol = tid - orow * ocols;
                int ix[12], iy[12];
                double wx[12], wy[12];
                const int nxt = area_taps(h.w, ocols, ocol, ix, wx, 12);
                const int nyt = area_taps(h.h, orows, orow, iy, wy, 12);
                val = 0.0;
                for (int a = 0; a < nyt; ++a)
                    for (int b = 0; b < nxt; ++b) val += sub[iy[a] * h.w + ix[b]] * wx[b] * wy[a];
                (void)orows;
            }
            const double eps = meas_noise ? (double)meas_noise[(size_t)item * MC + tid] : 0.0;
            if (flags & IPP_GIVEN_OBSERVATION)
                val = eps;  // caller supplies z (update_grid_map(pos, z), mappings.py:114-121)
            else
                val = fmin(fmax(val + h.nv_d * eps, 0.0), 1.0);  // sensor_manipulations.py:56-57 (variance used as std)
            zz[tid] = val;
            const Block b = block_of(tid, h.nx, h.rf, h.w, h.h);
            double hx = 0.0;
            for (int a = 0; a < b.count(); ++a) {
                const int ly = b.y0 + a / b.bw, lx = b.x0 + a % b.bw;
                hx += b.weight * (double)mean_env[(h.yu + ly) * v.W + h.xl + lx];
            }
            vv[tid] = val - hx;  // mappings.py:195
        }
    } else if (tid < MC) {
        zz[tid] = 0.0;
        vv[tid] = 0.0;
    }

    if (obs_out) {  // ipp_observe: observation only
        __syncthreads();
        if (tid < MC) obs_out[(size_t)item * MC + tid] = (tid < m) ? (float)zz[tid] : 0.f;
        if (tid == 0) {
            obs_m[item] = m;
            if (obs_shape) {
                // reference shapes: rf=1 -> (h, w); rf=2 -> cv2 dsize transposition -> (ceil(w/rf), ceil(h/rf))
                obs_shape[2 * item + 0] = (h.rf == 1) ? h.h : (h.w + h.rf - 1) / h.rf;
                obs_shape[2 * item + 1] = (h.rf == 1) ? h.w : (h.h + h.rf - 1) / h.rf;
            }
            if (status_out) status_out[item] = h.status;
        }
        return;
    }

    // ------------------------------------------------------------------ gather the state rows of the footprint
    int ht_ld = 0;
    if (MODE == IPP_FACTOR) {
        // HT[i][k] = sum_{cells of block i} w * U[k][cell]   (m x r)
        ht_ld = (r + 3) & ~3;
        for (int idx = tid; idx < r * m; idx += kPrepThreads) {
            const int k = idx / m, i = idx - k * m;
            const Block b = block_of(i, h.nx, h.rf, h.w, h.h);
            const float* row = cov_env + (size_t)k * v.Npad;
            float s = 0.f;
            for (int a = 0; a < b.count(); ++a) {
                const int ly = b.y0 + a / b.bw, lx = b.x0 + a % b.bw;
                s += row[(h.yu + ly) * v.W + h.xl + lx];
            }
            big[i * ht_ld + k] = s * (float)b.weight;
        }
    } else {
        // PFF[a][b] = P[F_a][F_b]   (f x f)
        for (int idx = tid; idx < f * f; idx += kPrepThreads) {
            const int a = idx / f, b = idx - a * f;
            const int ca = (h.yu + a / h.w) * v.W + h.xl + a % h.w;
            const int cb = (h.yu + b / h.w) * v.W + h.xl + b % h.w;
            big[a * (FC + 1) + b] = cov_env[(size_t)ca * v.Npad + cb];
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ S = H P_FF H^T + R  (mappings.py:182-183)
    const int npairs = m * (m + 1) / 2;
    for (int p = wave; p < npairs; p += NW) {
        int j = (int)((sqrt(8.0 * p + 1.0) - 1.0) * 0.5);
        while (j * (j + 1) / 2 > p) --j;
        while ((j + 1) * (j + 2) / 2 <= p) ++j;
        const int i = p - j * (j + 1) / 2;  // i <= j
        const Block bi = block_of(i, h.nx, h.rf, h.w, h.h), bj = block_of(j, h.nx, h.rf, h.w, h.h);
        double acc = 0.0;
        if (lane < bi.count() * bj.count()) {
            const int a = lane / bj.count(), b = lane - a * bj.count();
            const int lya = bi.y0 + a / bi.bw, lxa = bi.x0 + a % bi.bw;
            const int lyb = bj.y0 + b / bj.bw, lxb = bj.x0 + b % bj.bw;
            if (MODE == IPP_FACTOR)
                acc = bi.weight * bj.weight * matern_d(lya - lyb, lxa - lxb, v.res, sv, ls);
            else
                acc = bi.weight * bj.weight * (double)big[(lya * h.w + lxa) * (FC + 1) + lyb * h.w + lxb];
        }
        if (MODE == IPP_FACTOR) {
            const float* hi = big + i * ht_ld;
            const float* hj = big + j * ht_ld;
            for (int k = lane; k < r; k += kWave) acc -= (double)hi[k] * (double)hj[k];
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            if (i == j) acc += R;
            S[i * LD + j] = acc;
            S[j * LD + i] = acc;
        }
    }
    __syncthreads();

    // ------------------------------------------------------------------ Cholesky S = C C^T (C lower), fp64
    // np.linalg.cholesky(S) returns C; the reference uses L = C^T (upper).  mappings.py:185
    for (int c = 0; c < m; ++c) {
        if (tid == 0) {
            double d = S[c * LD + c];
            for (int k = 0; k < c; ++k) d -= L[c * LD + k] * L[c * LD + k];
            if (!(d > 0.0)) okflag[0] = 0;
            L[c * LD + c] = sqrt(d);
        }
        __syncthreads();
        if (okflag[0] == 0) break;
        if (tid > c && tid < m) {
            double s = S[tid * LD + c];
            for (int k = 0; k < c; ++k) s -= L[tid * LD + k] * L[c * LD + k];
            L[tid * LD + c] = s / L[c * LD + c];
        }
        __syncthreads();
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
        __syncthreads();
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
        __syncthreads();
        if (tid < m) {  // y = S_inv v
            double s = 0.0;
            for (int i = 0; i < m; ++i) s += Li[tid * LD + i] * vv[i];
            yy[tid] = s;
        }
    } else {
        status = IPP_STATUS_NOT_PD;  // factor form cannot hold an indefinite update
    }
    __syncthreads();

    // ------------------------------------------------------------------ outputs for the streaming kernels
    const bool dead = (MODE == IPP_FACTOR) && !pd;
    for (int idx = tid; idx < MC * MC; idx += kPrepThreads) {
        const int i = idx / MC, j = idx - i * MC;
        const double val = (!dead && i < m && j < m) ? Li[i * LD + j] : 0.0;
        linv_out[idx] = (float)val;
        dbg[MC * MC + idx] = val;
        dbg[idx] = (i < m && j < m) ? S[i * LD + j] : 0.0;
    }
    for (int i = tid; i < MC; i += kPrepThreads) {
        const double yval = (!dead && !cov_only && i < m) ? yy[i] : 0.0;
        y_out[i] = (float)yval;
        dbg[2 * MC * MC + i] = (i < m) ? zz[i] : 0.0;
        dbg[2 * MC * MC + MC + i] = yval;
    }
    const int QS = v.q_stride;
    if (MODE == IPP_FACTOR) {
        // Q[k][j] = sum_{i<=j} HT[i][k] L_inv[i][j]:  U Q = U U[F,:]^T H_F^T L^-1
        for (int idx = tid; idx < r * QS; idx += kPrepThreads) {
            const int k = idx / QS, j = idx - k * QS;
            double s = 0.0;
            if (!dead && j < m)
                for (int i = 0; i <= j; ++i) s += (double)big[i * ht_ld + k] * Li[i * LD + j];
            q_out[idx] = (float)(-s);  // stored negated: the stream accumulates acc += row * Q
        }
    } else {
        // Q[fi][j] = w_f L_inv[blk(f)][j]  (normal)   or  w_f [blk(f) == j]  (fallback: stream PH^T)
        for (int idx = tid; idx < f * QS; idx += kPrepThreads) {
            const int fi = idx / QS, j = idx - fi * QS;
            const int ly = fi / h.w, lx = fi - ly * h.w;
            const int bi = (ly / h.rf) * h.nx + lx / h.rf;
            const Block b = block_of(bi, h.nx, h.rf, h.w, h.h);
            double s = 0.0;
            if (j < m) s = fallback ? ((bi == j) ? b.weight : 0.0) : b.weight * Li[bi * LD + j];
            q_out[idx] = (float)s;
        }
    }
    if (tid == 0) {
        ItemHdr ho = h;
        ho.status = status;
        ho.fallback = fallback;
        if (dead) { ho.commit = 0; ho.rows = 0; }
        v.hdr[item] = ho;
        if (status_out) status_out[item] = status;
    }
}

}  // namespace ipp
