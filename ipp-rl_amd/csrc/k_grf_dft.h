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

// OPT: rows of one column a thread owns in stage 3, ceil(n / (NT / n)) <= OPT;  NT: workgroup size;
// WLDS: white noise staged in LDS
template <int OPT, int NT, bool WLDS>
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

    // stage-3 ownership: column x, rows yg + j R
    const int R = kGrfThreads / n > 0 ? kGrfThreads / n : 1;
    const int x3 = tid % n, yg = tid / n;
    const bool own = tid < R * n;
    double acc[OPT];
#pragma unroll
    for (int j = 0; j < OPT; ++j) acc[j] = 0.0;
    const int stepR = R % n;  // (row y advances by R per j: its phase k y mod n advances by k R mod n)

    const int n_k = n / 2 + 1;
    for (int k0 = 0; k0 < n_k; k0 += kc) {
        const int kn = min(kc, n_k - k0);
        __syncthreads();  // previous chunk fully consumed (and ws / cs_s visible on the first pass)
        for (int i = tid; i < kn * n; i += kGrfThreads) gs[i] = g[(size_t)k0 * n + i];
        // ---- stage 1: A[k][x] = sum_y w[y][x] (cos - i sin)(2 pi k y / n)
        for (int idx = tid; idx < kn * n; idx += kGrfThreads) {
            const int kk = idx / n, x = idx - kk * n, k = k0 + kk;
            double re = 0.0, im = 0.0;
            int p = 0;
            for (int y = 0; y < n; ++y) {
                const double wv = (double)wsrc[y * n + x];
                const double2 c = cs_s[p];
                re = fma(wv, c.x, re);
                im = fma(-wv, c.y, im);
                p += k;
                p -= (p >= n) ? n : 0;
            }
            A[idx] = make_double2(re, im);
        }
        __syncthreads();
        // ---- stage 2: D[k][x] = sum_x' A[k][x'] g_k[(x - x') mod n]
        for (int idx = tid; idx < kn * n; idx += kGrfThreads) {
            const int kk = idx / n, x = idx - kk * n;
            const double2* __restrict__ arow = A + (size_t)kk * n;
            const double* __restrict__ grow = gs + (size_t)kk * n;
            double re = 0.0, im = 0.0;
            int d = x;
            for (int xp = 0; xp < n; ++xp) {
                const double2 a = arow[xp];
                const double gg = grow[d];
                re = fma(a.x, gg, re);
                im = fma(a.y, gg, im);
                d = (d == 0) ? n - 1 : d - 1;
            }
            D[idx] = make_double2(re, im);
        }
        __syncthreads();
        // ---- stage 3: f[y][x] += Re(D[k][x] (cos + i sin)(2 pi k y / n)).  Row y = yg + j R of this thread has
        // phase (k yg + j (k R)) mod n: two running values per k instead of a table of OPT phases
        if (own) {
            for (int kk = 0; kk < kn; ++kk) {
                const int k = k0 + kk;
                const double2 d = D[(size_t)kk * n + x3];
                int p = (int)(((long)k * yg) % n);
                const int dp = (int)(((long)k * stepR) % n);
#pragma unroll
                for (int j = 0; j < OPT; ++j) {
                    if (yg + j * R < n) {
                        const double2 c = cs_s[p];
                        acc[j] = fma(d.x, c.x, fma(-d.y, c.y, acc[j]));
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
            if (yg + j * R < n) { lo = fmin(lo, acc[j]); hi = fmax(hi, acc[j]); }
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
    if (gt_out) {
        float* gt = gt_out + (size_t)item * N;
        if (own) {
#pragma unroll
            for (int j = 0; j < OPT; ++j) {
                const int y = yg + j * R;
                if (y < n) gt[y * n + x3] = (float)((acc[j] - dlo) / span);
            }
        }
        return;
    }
    float* gt = v.gt + (size_t)env * v.Npad;
    if (own) {
#pragma unroll
        for (int j = 0; j < OPT; ++j) {
            const int y = yg + j * R;
            if (y < n) gt[y * n + x3] = (float)((acc[j] - dlo) / span);
        }
    }
    for (int i = N + tid; i < v.Npad; i += kGrfThreads) gt[i] = 0.f;
}

}  // namespace ipp
