// Gaussian random field by a half-spectrum DFT along the rows and k-dependent circular convolutions along the
// columns (simulations/ground_truths.py:14-33: field = Re ifft2(fft2(white) * amp), min-max normalised).
//
// amp is real and even for even n (fft_indices, ground_truths.py:7-11), the white noise is real, so with
//   A[k][x] = sum_y w[y][x] e^{-2 pi i k y / n}                       k = 0 .. n/2   (rows n-k are conjugates)
//   D[k][x] = sum_x' A[k][x'] g_k[(x - x') mod n]                     g_k = c_k/n * IDFT_x(amp[k][.]) real, even
//   f[y][x] = sum_k Re(D[k][x] e^{+2 pi i k y / n})                   c_0 = c_{n/2} = 1, else 2
// the field costs 3 n^3 fp64 FMAs per env instead of the n^4 of the direct circular convolution (k_grf_conv):
// 17x fewer at n = 50, 33x at n = 100.  The constant 1/n of the row inverse is dropped: the field is min-max
// normalised right after.  cs[j] = (cos, sin)(2 pi j / n) and g come from the host in fp64.
//
// One workgroup per env (256 threads up to n = 100, 1024 threads up to n = 256).  LDS: the white noise (fp32; read
// from global memory instead when it would not fit, n > 100), cs, and one chunk of KC spectrum rows (A, D complex
// fp64, g real) -- 25 KB at n = 50 -- so several fields run per CU beside the step kernel.
// Stage 3 keeps the outputs in registers: thread (x, yg) owns rows y = yg, yg + R, ... of column x, so D[k][x] is
// read once per k and the phase (k y mod n) advances by y per k without a multiply.  The normalisation is fused
// (workgroup min / max), the field goes straight to the env slot or the caller's buffer.
#pragma once
#include "ipp_common.h"

namespace ipp {

__host__ __device__ inline size_t grf_dft_lds_bytes(int n, int kc, bool w_in_lds) {
    return (w_in_lds ? (size_t)n * n * 4 : 0) + (size_t)n * 16 + (size_t)kc * n * (16 + 16 + 8) + 2 * 16 * 8;
}

// Stages 1 and 2 for one chunk of kn spectrum rows, XB adjacent columns per thread: the twiddle of stage 1 and the
// spectrum value of stage 2 are read once for XB outputs (an LDS read feeds 2 XB FMAs instead of 2), and the index
// arithmetic is shared.  XB divides n.  Ends with the barrier that publishes D.
//   stage 1: A[k][x] = sum_y w[y][x] (cos - i sin)(2 pi k y / n)
//   stage 2: D[k][x] = sum_x' A[k][x'] g_k[(x - x') mod n]
template <int XB, int NT>
__device__ __forceinline__ void grf_stage12(const int n, const int k0, const int kn, const float* __restrict__ wsrc,
                                            const double2* __restrict__ cs_s, const double* __restrict__ gs,
                                            double2* __restrict__ A, double2* __restrict__ D, const int tid) {
    typedef float wvec __attribute__((ext_vector_type(XB)));
    const int nxb = n / XB;
    for (int idx = tid; idx < kn * nxb; idx += NT) {
        const int kk = idx / nxb, x0 = (idx - kk * nxb) * XB, k = k0 + kk;
        double re[XB], im[XB];
#pragma unroll
        for (int b = 0; b < XB; ++b) { re[b] = 0.0; im[b] = 0.0; }
        int p = 0;
        for (int y = 0; y < n; ++y) {
            const wvec w = *reinterpret_cast<const wvec*>(wsrc + y * n + x0);  // XB | n: aligned
            const double2 c = cs_s[p];
#pragma unroll
            for (int b = 0; b < XB; ++b) {
                const double wv = (double)w[b];
                re[b] = fma(wv, c.x, re[b]);
                im[b] = fma(-wv, c.y, im[b]);
            }
            p += k;
            p -= (p >= n) ? n : 0;
        }
#pragma unroll
        for (int b = 0; b < XB; ++b) A[kk * n + x0 + b] = make_double2(re[b], im[b]);
    }
    __syncthreads();
    for (int idx = tid; idx < kn * nxb; idx += NT) {
        const int kk = idx / nxb, x0 = (idx - kk * nxb) * XB;
        const double2* __restrict__ arow = A + (size_t)kk * n;
        const double* __restrict__ grow = gs + (size_t)kk * n;
        double re[XB], im[XB];
#pragma unroll
        for (int b = 0; b < XB; ++b) { re[b] = 0.0; im[b] = 0.0; }
        // output b at x' = xp0 + s needs g[(x0 + b - xp0 - s) mod n] = gw[b - s + XB - 1], gw[j] = g[(base + j) mod n]
        int base = x0 - (XB - 1);
        base += (base < 0) ? n : 0;
        for (int xp0 = 0; xp0 < n; xp0 += XB) {
            double gw[2 * XB - 1];
#pragma unroll
            for (int j = 0; j < 2 * XB - 1; ++j) {
                int gi = base + j;
                gi -= (gi >= n) ? n : 0;
                gw[j] = grow[gi];
            }
#pragma unroll
            for (int sft = 0; sft < XB; ++sft) {
                const double2 a = arow[xp0 + sft];
#pragma unroll
                for (int b = 0; b < XB; ++b) {
                    re[b] = fma(a.x, gw[b - sft + XB - 1], re[b]);
                    im[b] = fma(a.y, gw[b - sft + XB - 1], im[b]);
                }
            }
            base -= XB;
            base += (base < 0) ? n : 0;
        }
#pragma unroll
        for (int b = 0; b < XB; ++b) D[kk * n + x0 + b] = make_double2(re[b], im[b]);
    }
    __syncthreads();
}

// Stage-3 ownership: thread (cg, yg) owns the XB3 columns cg XB3 .. cg XB3 + XB3 - 1 of the rows yg, yg + R, ...
// (R = NT / (n / XB3) row groups, at most OPT rows per thread: OPT >= ceil(n / R)), all of them in registers across
// the chunks of the spectrum: per k a thread reads XB3 spectrum values and one twiddle per row for 2 XB3 FMAs each.
// NT: workgroup size; WLDS: white noise staged in LDS.  XB3 divides n.
template <int OPT, int XB3, int NT, bool WLDS>
__global__ __launch_bounds__(NT) void k_grf_dft(View v, const int* __restrict__ env_ids, int n_items,
                                                         const float* __restrict__ white, const double2* __restrict__ cs,
                                                         const double* __restrict__ g, int kc,
                                                         float* __restrict__ gt_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_dft[];
    const int n = v.W, N = v.N;
    const int item = blockIdx.x;
    if (item >= n_items) return;
    const int env = gt_out ? 0 : (env_ids ? env_ids[item] : item);
    if (env < 0 || env >= v.cap) return;
    double2* cs_s = reinterpret_cast<double2*>(smem_dft);
    double2* A = cs_s + n;
    double2* D = A + (size_t)kc * n;
    double* gs = reinterpret_cast<double*>(D + (size_t)kc * n);
    constexpr int kGrfThreads = NT;
    float* ws = reinterpret_cast<float*>(gs + (size_t)kc * n);
    double* red = reinterpret_cast<double*>(ws + (WLDS ? N : 0));  // [2 * waves] min / max per wave (N is even)
    const int tid = threadIdx.x;
    const float* __restrict__ wn = white + (size_t)item * N;
    if (WLDS)
        for (int i = tid; i < N; i += kGrfThreads) ws[i] = wn[i];
    const float* __restrict__ wsrc = WLDS ? ws : wn;
    for (int i = tid; i < n; i += kGrfThreads) cs_s[i] = cs[i];

    const int ncg = n / XB3;
    const int R = kGrfThreads / ncg > 0 ? kGrfThreads / ncg : 1;
    const int cg = tid % ncg, yg = tid / ncg, x3 = cg * XB3;
    const bool own = tid < R * ncg;
    double acc[OPT][XB3];
#pragma unroll
    for (int j = 0; j < OPT; ++j)
#pragma unroll
        for (int b = 0; b < XB3; ++b) acc[j][b] = 0.0;
    const int stepR = R % n;  // (row y advances by R per j: its phase k y mod n advances by k R mod n)

    const int n_k = n / 2 + 1;
    for (int k0 = 0; k0 < n_k; k0 += kc) {
        const int kn = min(kc, n_k - k0);
        __syncthreads();  // previous chunk fully consumed (and ws / cs_s visible on the first pass)
        for (int i = tid; i < kn * n; i += kGrfThreads) gs[i] = g[(size_t)k0 * n + i];
        // ---- stages 1 and 2, XB columns per thread (4 when 4 | n, else 2: n is even; 2 for the 1024-thread
        // workgroups, whose 128-VGPR budget also holds the fp64 accumulators of stage 3)
        if (NT <= 256 && (n & 3) == 0) grf_stage12<4, NT>(n, k0, kn, wsrc, cs_s, gs, A, D, tid);
        else grf_stage12<2, NT>(n, k0, kn, wsrc, cs_s, gs, A, D, tid);
        // ---- stage 3: f[y][x] += Re(D[k][x] (cos + i sin)(2 pi k y / n)).  Row y = yg + j R of this thread has
        // phase (k yg + j (k R)) mod n: two running values per k instead of a table of OPT phases
        if (own) {
            for (int kk = 0; kk < kn; ++kk) {
                const int k = k0 + kk;
                double2 d[XB3];
#pragma unroll
                for (int b = 0; b < XB3; ++b) d[b] = D[(size_t)kk * n + x3 + b];
                int p = (int)(((long)k * yg) % n);
                const int dp = (int)(((long)k * stepR) % n);
#pragma unroll
                for (int j = 0; j < OPT; ++j) {
                    if (yg + j * R < n) {
                        const double2 c = cs_s[p];
#pragma unroll
                        for (int b = 0; b < XB3; ++b) acc[j][b] = fma(d[b].x, c.x, fma(-d[b].y, c.y, acc[j][b]));
                    }
                    p += dp;
                    p -= (p >= n) ? n : 0;
                }
            }
        }
    }

    // ---- min-max normalisation to [0, 1] (ground_truths.py:31), fp64 like the reference
    double lo = INFINITY, hi = -INFINITY;
    if (own) {
#pragma unroll
        for (int j = 0; j < OPT; ++j)
            if (yg + j * R < n) {
#pragma unroll
                for (int b = 0; b < XB3; ++b) { lo = fmin(lo, acc[j][b]); hi = fmax(hi, acc[j][b]); }
            }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, off));
        hi = fmax(hi, __shfl_xor(hi, off));
    }
    constexpr int NW = NT / kWave;
    if ((tid & (kWave - 1)) == 0) { red[tid / kWave] = lo; red[NW + tid / kWave] = hi; }
    __syncthreads();
    double dlo = red[0], dhi = red[NW];
#pragma unroll
    for (int w = 1; w < NW; ++w) { dlo = fmin(dlo, red[w]); dhi = fmax(dhi, red[NW + w]); }
    const double span = dhi - dlo;
    typedef float outv __attribute__((ext_vector_type(XB3)));
    float* gt = gt_out ? gt_out + (size_t)item * N : gt_plane(v, env);
    if (own) {
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int y = yg + j * R;
            if (y < n) {
                outv o;
#pragma unroll
                for (int b = 0; b < XB3; ++b) o[b] = (float)((acc[j][b] - dlo) / span);
                *reinterpret_cast<outv*>(gt + y * n + x3) = o;  // XB3 | n and the field starts 16-byte aligned
            }
        }
    }
    if (!gt_out)
        for (int i = N + tid; i < v.Npad; i += kGrfThreads) gt[i] = 0.f;
}

}  // namespace ipp
