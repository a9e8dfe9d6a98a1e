// Gaussian random field through FAST Hartley transforms (simulations/ground_truths.py:14-33: field = Re ifft2(fft2(white) * amp),
// min-max normalised) for n = 50 and n = 100 -- the grids of BASELINE configs[1] .. [3].
//
// Same identity as k_grf_hartley.h: the white noise is real and the amplitude is real and even in each spectral index, so with the
// separable Hartley transform T = H (x) H (H[j][k] = cos + sin of 2 pi j k / n, H H = n I)
//      field = H (amp .* (H w H)) H          (constants dropped: the field is min-max normalised right after)
// -- but the four multiplications by H are not GEMMs (8 n^3 flops per field: 2048 fields of 100x100 per step of configs[2] took
// 0.47 ms, as long as the step kernel, on 101 KB of LDS and 168 VGPRs that evicted the step's workgroups).  Here every multiplication
// is n one-dimensional Hartley transforms computed as n / 2 COMPLEX FFTs of two real vectors packed as real and imaginary part
//      z = x + i y,  Z = FFT(z):   DHT_x[j] = a + c - b + d,   DHT_y[j] = b + d + a - c      (Z[j] = a + i b, Z[n - j] = c + i d)
// and a length-n FFT is two steps of small FFTs held in registers (n = N1 x 10, N1 = 5 or 10: the four-step scheme):
//   A  thread (pair f, column c < 10):  FFT-N1 over the elements c + 10 r, times the twiddles w_n^(c k), written back in place
//   B  thread (pair f, row k < N1):     FFT-10 over the elements 10 k + c, output q is X[k + N1 q]
//   C  thread (pair f, ...):            the Hartley combination of the elements j and n - j above (and the amplitude, once)
// ~0.7 MFLOP per field instead of 10, all in place on ONE fp64 array in LDS (n = 100: 100 x 100, exactly 80 KB: two fields per CU; n = 50:
// 50 x 51, 20 KB), twiddles from the (cos, sin) table of k_grf_dft.h, ~90 VGPRs.
// Checked against numpy.fft in fp64 to 7e-16 (tools/probes/grf_fft_model.py is this kernel line by line in NumPy).
#pragma once
#include "ipp_common.h"
#include "k_misc.h"

namespace ipp {

// n x LD fp64 array + min / max scratch.  n = 100: LD = n, exactly 80 KB = half of a CU's LDS, so that TWO fields are resident per CU (the
// kernel is bound by LDS round trips between its sixteen barriers: a second workgroup fills them); n = 50: LD = n + 1 (seven per CU either way).
__host__ __device__ inline int grf_fft_ld(int n) { return n == 100 ? n : n + 1; }
__host__ __device__ inline size_t grf_fft_lds_bytes(int n, int real_bytes = 8) {
    return n == 100 ? (size_t)10 * 1024 * real_bytes : ((size_t)n * (n + 1) + 2 * 16) * real_bytes + 64;
}

// forward DFT of 5 complex numbers (re / im arrays, in place); T = double, or float (IPP_GRF_FP32: see k_grf_fft)
template <typename T>
__device__ __forceinline__ void fft5(T (&re)[5], T (&im)[5]) {
    constexpr T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;  // cos 72, cos 144 degrees
    constexpr T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;   // sin 72, sin 144 degrees
    const T t1r = re[1] + re[4], t1i = im[1] + im[4], t2r = re[2] + re[3], t2i = im[2] + im[3];
    const T t3r = re[1] - re[4], t3i = im[1] - im[4], t4r = re[2] - re[3], t4i = im[2] - im[3];
    const T m1r = re[0] + c1 * t1r + c2 * t2r, m1i = im[0] + c1 * t1i + c2 * t2i;
    const T m2r = re[0] + c2 * t1r + c1 * t2r, m2i = im[0] + c2 * t1i + c1 * t2i;
    const T u1r = s1 * t3r + s2 * t4r, u1i = s1 * t3i + s2 * t4i;
    const T u2r = s2 * t3r - s1 * t4r, u2i = s2 * t3i - s1 * t4i;
    re[0] += t1r + t2r; im[0] += t1i + t2i;
    // X1 = m1 - i u1, X4 = m1 + i u1, X2 = m2 - i u2, X3 = m2 + i u2     (-i (ur + i ui) = ui - i ur)
    re[1] = m1r + u1i; im[1] = m1i - u1r;
    re[4] = m1r - u1i; im[4] = m1i + u1r;
    re[2] = m2r + u2i; im[2] = m2i - u2r;
    re[3] = m2r - u2i; im[3] = m2i + u2r;
}
// forward DFT of 10 complex numbers: two FFT-5 on the even / odd elements + the radix-2 butterflies
template <typename T>
__device__ __forceinline__ void fft10(T (&re)[10], T (&im)[10]) {
    T er[5] = {re[0], re[2], re[4], re[6], re[8]}, ei[5] = {im[0], im[2], im[4], im[6], im[8]};
    T qr[5] = {re[1], re[3], re[5], re[7], re[9]}, qi[5] = {im[1], im[3], im[5], im[7], im[9]};
    fft5(er, ei);
    fft5(qr, qi);
    // w_10^k = exp(-2 pi i k / 10), k = 0 .. 4
    constexpr T wr[5] = {(T)1.0, (T)0.80901699437494742410, (T)0.30901699437494742410, (T)-0.30901699437494742410, (T)-0.80901699437494742410};
    constexpr T wi[5] = {(T)0.0, (T)-0.58778525229247312917, (T)-0.95105651629515357212, (T)-0.95105651629515357212, (T)-0.58778525229247312917};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const T tr = wr[k] * qr[k] - wi[k] * qi[k], ti = wr[k] * qi[k] + wi[k] * qr[k];
        re[k] = er[k] + tr; im[k] = ei[k] + ti;
        re[k + 5] = er[k] - tr; im[k + 5] = ei[k] - ti;
    }
}
template <int N, typename T>
__device__ __forceinline__ void fft_small(T (&re)[N], T (&im)[N]) {
    if constexpr (N == 5) fft5(re, im); else fft10(re, im);
}

// One Hartley pass: X <- DHT along `axis` of every vector (axis 1: the rows of X, axis 0: its columns), times amp on the way out when
// amp != nullptr.  element j of vector v sits at X[v * sv + j * sj].
template <int N1, int NT, typename T>
__device__ __forceinline__ void grf_fft_pass(T* X, int sv, int sj, const double2* __restrict__ tw, const double* __restrict__ amp, int amp_ld, int tid) {
    constexpr int N2 = 10, n = N1 * N2;
    const int f = tid / 10, c = tid - 10 * f;  // pair of vectors, position inside the group of ten threads
    const bool on = f < n / 2;
    T* x0 = X + (size_t)(2 * f) * sv;  // real part: vector 2 f, imaginary part: vector 2 f + 1
    T* x1 = x0 + sv;
    // ---- A: FFT-N1 over the elements c + 10 r, twiddles w_n^(c k)
    if (on) {
        T re[N1], im[N1];
#pragma unroll
        for (int r = 0; r < N1; ++r) { re[r] = x0[(c + N2 * r) * sj]; im[r] = x1[(c + N2 * r) * sj]; }
        fft_small<N1, T>(re, im);
#pragma unroll
        for (int k = 0; k < N1; ++k) {
            const double2 t = tw[c * k];  // (cos, sin)(2 pi c k / n), c k < n: w_n^(c k) = cos - i sin
            const T wr = (T)t.x, wi = (T)-t.y;
            x0[(c + N2 * k) * sj] = re[k] * wr - im[k] * wi;
            x1[(c + N2 * k) * sj] = re[k] * wi + im[k] * wr;
        }
    }
    __syncthreads();
    // ---- B: FFT-10 over the elements 10 k + c (threads k < N1), output q is X[k + N1 q]
    {
        T re[N2], im[N2];
        const bool onb = on && c < N1;
        if (onb) {
#pragma unroll
            for (int q = 0; q < N2; ++q) { re[q] = x0[(N2 * c + q) * sj]; im[q] = x1[(N2 * c + q) * sj]; }
            fft10(re, im);
        }
        __syncthreads();  // (every row is in registers before the first scattered store)
        if (onb) {
#pragma unroll
            for (int q = 0; q < N2; ++q) { x0[(c + N1 * q) * sj] = re[q]; x1[(c + N1 * q) * sj] = im[q]; }
        }
    }
    __syncthreads();
    // ---- C: the two Hartley transforms out of the packed spectrum, pairs (j, n - j), j = 0 .. n / 2
    if (on) {
        for (int j = c; j <= n / 2; j += 10) {
            const int jj = (j == 0) ? 0 : n - j;
            const T a = x0[j * sj], b = x1[j * sj], cc = x0[jj * sj], d = x1[jj * sj];
            T xj = a + cc - b + d, yj = b + d + a - cc, xjj = cc + a - d + b, yjj = d + b + cc - a;
            if (amp) {  // amp is even in both indices and symmetric: amp[v][j] whichever the axis
                const T a0j = (T)amp[(size_t)(2 * f) * amp_ld + j], a1j = (T)amp[(size_t)(2 * f + 1) * amp_ld + j];
                const T a0jj = (T)amp[(size_t)(2 * f) * amp_ld + jj], a1jj = (T)amp[(size_t)(2 * f + 1) * amp_ld + jj];
                xj *= a0j; yj *= a1j; xjj *= a0jj; yjj *= a1jj;
            }
            x0[j * sj] = xj; x1[j * sj] = yj;
            x0[jj * sj] = xjj; x1[jj * sj] = yjj;
        }
    }
    __syncthreads();
}

// Where the white noise of a field comes from when it is drawn in the kernel (white == nullptr): the numbers ipp_fill_normal_rows would
// have written for row id (row_ids ? row_ids[item] : item) + row_offset of a row_len = N fill -- the fill kernel's 4 N bytes per field
// written and read back (160 MB per step of configs[2], 13 % of its GPU time) do not exist then.
struct GrfNoise {
    const int* row_ids;
    long long row_offset;
    uint64_t seed, subseq;
    // fields of several episodes in one launch (ipp_generate_grf_groups): field i belongs to group i / group_rows and draws from
    // subseq + group_subseq[group] (by value: no upload in front of the launch); group_rows == 0: one group.  A NEGATIVE row id skips the field.
    int group_rows;
    long long group_subseq[16];
    int to_alt;  // gt_out == NULL: field i goes to the ALTERNATE ground-truth plane of env row_ids[i] (staged for its next episode; the reset flips)
};

// One workgroup per field.  white [n_items][N] float standard normals (or nullptr: GrfNoise); amp [n][amp_ld] doubles (the table of
// k_grf_hartley.h: zero padded, leading dimension amp_ld); result into the env slots (gt_out == nullptr) or gt_out [n_items][N].
// T = float (default): the transforms in fp32 -- inputs and output are fp32 anyway and the field is min-max normalised: 6e-7 of the
// normalised field against numpy's fp64 path (bar 1e-5, tests/test_hip_big_grids.py), half the LDS (four 100x100 fields per CU) and
// plain-rate arithmetic (T = double -- the reference's precision end to end -- measured no closer to the bar and is not instantiated).
template <int N1, typename T>
__global__ __launch_bounds__(N1 == 10 ? 512 : 256) void k_grf_fft(View v, const int* __restrict__ env_ids, int n_items, const float* __restrict__ white,
                                                                  const double* __restrict__ amp, int amp_ld, float* __restrict__ gt_out, GrfNoise gn,
                                                                  const double2* __restrict__ tw) {
    constexpr int n = N1 * 10, LD = (N1 == 10) ? n : n + 1, NT = (N1 == 10) ? 512 : 256, NW = NT / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gf[];
    const int item = blockIdx.x;
    if (item >= n_items) return;
    if (!white && gn.row_ids && gn.row_ids[item] < 0) return;  // (padding row of a staged block)
    const int env = gt_out ? 0 : (gn.to_alt ? gn.row_ids[item] : (env_ids ? env_ids[item] : item));
    if (env < 0 || env >= v.cap) return;
    T* X = reinterpret_cast<T*>(smem_gf);
    // min / max per wave: behind the array (n = 50) or, at n = 100 where the array fills the workgroup's 80 KB, in its first row once the
    // field has been reduced into registers (see below)
    T* red = (N1 == 10) ? X : X + (size_t)n * LD;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N = n * n;
    {
        if (white) {
            const float2* __restrict__ w2 = reinterpret_cast<const float2*>(white + (size_t)item * N);
            for (int i = tid; i < N / 2; i += NT) {
                const float2 wv = w2[i];
                const int y = i / (n / 2), x = 2 * (i - y * (n / 2));
                X[y * LD + x] = (T)wv.x;
                X[y * LD + x + 1] = (T)wv.y;
            }
        } else {
            const uint64_t rid = (uint64_t)((gn.row_ids ? (long long)gn.row_ids[item] : (long long)item) + gn.row_offset);
            const uint64_t subseq = gn.subseq + (gn.group_rows > 0 ? (uint64_t)gn.group_subseq[min(item / gn.group_rows, 15)] : 0ull);
            constexpr int qpr = (N + 3) / 4;  // counters per row: element e of the row is normal e & 3 of counter rid * qpr + (e >> 2)
            for (int qc = tid; qc < qpr; qc += NT) {
                float nrm[4];
                philox_normal4(rid * (uint64_t)qpr + (uint64_t)qc, subseq, gn.seed, nrm);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int e = 4 * qc + h;
                    if (e < N) { const int y = e / n; X[y * LD + (e - y * n)] = (T)nrm[h]; }
                }
            }
        }
    }
    __syncthreads();
    grf_fft_pass<N1, NT, T>(X, LD, 1, tw, nullptr, 0, tid);      // rows:    w H
    grf_fft_pass<N1, NT, T>(X, 1, LD, tw, amp, amp_ld, tid);     // columns: amp .* (H w H)
    grf_fft_pass<N1, NT, T>(X, LD, 1, tw, nullptr, 0, tid);
    grf_fft_pass<N1, NT, T>(X, 1, LD, tw, nullptr, 0, tid);      // field (x constants)
    // ---- min-max normalisation to [0, 1] (ground_truths.py:31), fp64 like the reference; the field moves into registers first (the
    // cross-wave exchange of the extrema reuses the array's first row)
    constexpr int PER = (n * n + NT - 1) / NT;
    T val[PER];
    T lo = (T)INFINITY, hi = (T)-INFINITY;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = tid + q * NT;
        const int y = min(i, N - 1) / n, x = min(i, N - 1) - y * n;
        val[q] = X[y * LD + x];
        if (i < N) { lo = val[q] < lo ? val[q] : lo; hi = val[q] > hi ? val[q] : hi; }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        const T ol = __shfl_xor(lo, off), oh = __shfl_xor(hi, off);
        lo = ol < lo ? ol : lo; hi = oh > hi ? oh : hi;
    }
    __syncthreads();  // (every thread holds its cells)
    if (lane == 0) { red[wave] = lo; red[16 + wave] = hi; }
    __syncthreads();
    double dlo = (double)red[0], dhi = (double)red[16];
#pragma unroll
    for (int w = 1; w < NW; ++w) { dlo = fmin(dlo, (double)red[w]); dhi = fmax(dhi, (double)red[16 + w]); }
    // (x - lo) * (1 / span) in fp64, rounded to fp32: the quotient's fp32 rounding except where the fp64 value sits within an fp64 ulp
    // of an fp32 rounding boundary (1 cell in 10^8); 10^4 fp64 divisions per field were a third of the kernel's time
    const double inv_span = 1.0 / (dhi - dlo);
    float* gt = gt_out ? gt_out + (size_t)item * N : (gn.to_alt ? v.gt + (size_t)gt_alt_slot(v, env) * v.Npad : gt_plane(v, env));
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int i = tid + q * NT;
        if (i < N) gt[i] = (float)(((double)val[q] - dlo) * inv_span);
    }
    if (!gt_out)
        for (int i = N + tid; i < v.Npad; i += NT) gt[i] = 0.f;
    if (!gt_out && gn.to_alt && tid == 0) v.gt_slot[v.cap + env] = 1;  // the env's next folded reset may flip (stream order: the reset's launch waits for this one)
}

}  // namespace ipp
