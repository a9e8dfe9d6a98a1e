// NN input state plane: the N x N covariance of one env slot, rows and columns outside the
// adaptive mask zeroed, min-max normalised to [0, 1] -- planning/common/features.py:91-101 and :74-81
// (generate_input_feature_planes / min_max_normalize), produced on the device from the slot's state so that the
// 4 N^2-byte plane never crosses PCIe as an fp64 matrix.  The mask uses the mean the caller passes
// (adaptive_info["mean"], the CURRENT map mean, also for older states of the history) and the slot's own diag(P).
// Factor slots are densified first (k_score_densify); constant planes (position, budget) are host-side fills.
#pragma once
#include "ipp_common.h"

namespace ipp {

__device__ __forceinline__ int plane_key(float x) {  // order-preserving float -> int
    const int b = __float_as_int(x);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__device__ __forceinline__ float plane_unkey(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

// mask_i = mean_i + k * P_ii >= thr (planning/common/rewards.py:8-12); all ones without IPP_ADAPTIVE
__global__ void k_plane_mask(View v, int env, const float* __restrict__ mean_in, unsigned flags, float* __restrict__ mask,
                             int* __restrict__ mm) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { mm[0] = 0x7fffffff; mm[1] = (int)0x80000000; }
    if (i >= v.Npad) return;
    float mk = 0.f;
    if (i < v.N) {
        const double mu = mean_in ? (double)mean_in[i] : (double)v.mean[(size_t)env * v.Npad + i];
        mk = (!(flags & IPP_ADAPTIVE) || (mu + v.kf * (double)v.diag[(size_t)env * v.Npad + i] >= v.thr)) ? 1.f : 0.f;
    }
    mask[i] = mk;
}

// extrema of the masked matrix (zeros of masked rows / columns included, like the reference's in-place masking)
__global__ __launch_bounds__(256) void k_plane_minmax(View v, const float* __restrict__ P, const float* __restrict__ mask,
                                                      int* __restrict__ mm) {
    __shared__ float smin[4], smax[4];
    float lo = INFINITY, hi = -INFINITY;
    for (int i = blockIdx.x; i < v.N; i += gridDim.x) {
        const float mi = mask[i];
        const float* row = P + (size_t)i * v.Npad;
        for (int j = threadIdx.x; j < v.N; j += blockDim.x) {
            const float x = (mi != 0.f && mask[j] != 0.f) ? row[j] : 0.f;
            lo = fminf(lo, x);
            hi = fmaxf(hi, x);
        }
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        lo = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
        hi = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        atomicMin(&mm[0], plane_key(lo));
        atomicMax(&mm[1], plane_key(hi));
    }
}

// out[i][j] = (x - min) / (max - min), or x / max when min == max (features.py:74-81)
__global__ __launch_bounds__(256) void k_plane_write(View v, const float* __restrict__ P, const float* __restrict__ mask,
                                                     const int* __restrict__ mm, float* __restrict__ out) {
    const double lo = (double)plane_unkey(mm[0]), hi = (double)plane_unkey(mm[1]);
    const bool flat = (lo == hi);
    for (int i = blockIdx.x; i < v.N; i += gridDim.x) {
        const float mi = mask[i];
        const float* row = P + (size_t)i * v.Npad;
        for (int j = threadIdx.x; j < v.N; j += blockDim.x) {
            const double x = (mi != 0.f && mask[j] != 0.f) ? (double)row[j] : 0.0;
            out[(size_t)i * v.N + j] = (float)(flat ? x / hi : (x - lo) / (hi - lo));
        }
    }
}

}  // namespace ipp
