// Microbenchmark: what HBM read bandwidth does the k_gain access pattern allow, with no arithmetic?
//   flat   : one contiguous region, grid-stride float4 reads (the guide's "float4 copy" style ceiling, read-only)
//   stream : workgroup (env, tile) reads R rows of Npad floats from its env slab (stride Npad), float4 per lane,
//            exactly the row/tile geometry of k_gain; per-env R optionally varies like the staggered episode mix.
// Build: hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k_flat(const float4* __restrict__ p, size_t n4, float* out) {
    float4 a = make_float4(0, 0, 0, 0);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = p[i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

template <int PIPE, bool NT>
__global__ void k_stream(const float* __restrict__ base, size_t slot, int npad, int tiles, const int* __restrict__ rows, float* out) {
    const int b = blockIdx.x;
    const int xcd = b & 7, s = b >> 3;
    const int env = (s / tiles) * 8 + xcd, tile = s % tiles;
    const int R = rows[env];
    const float* p = base + (size_t)env * slot + (size_t)tile * 4 * blockDim.x + 4 * threadIdx.x;
    float4 a = make_float4(0, 0, 0, 0);
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int k = 0; k < R; k += PIPE) {
        f4 v[PIPE];
#pragma unroll
        for (int i = 0; i < PIPE; ++i) {
            const f4* q = reinterpret_cast<const f4*>(p + (size_t)min(k + i, R - 1) * npad);
            v[i] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < PIPE; ++i) { a.x += v[i][0]; a.y += v[i][1]; a.z += v[i][2]; a.w += v[i][3]; }
    }
    if (a.x + a.y + a.z + a.w == 123.456f) out[0] = a.x;
}

// stream + the arithmetic of k_gain: 36 FMAs per float4 against 9 broadcast values (registers or LDS)
template <int PIPE, bool NT, bool LDSQ>
__global__ __launch_bounds__(640, 4) void k_stream_fma(const float* __restrict__ base, size_t slot, int npad, int tiles, const int* __restrict__ rows, float* out) {
    __shared__ __attribute__((aligned(16))) float Qs[368 * 12];
    const int b = blockIdx.x;
    const int xcd = b & 7, s = b >> 3;
    const int env = (s / tiles) * 8 + xcd, tile = s % tiles;
    const int R = rows[env];
    for (int i = threadIdx.x; i < 368 * 12; i += blockDim.x) Qs[i] = 1e-3f * (i % 7);
    __syncthreads();
    const float* p = base + (size_t)env * slot + (size_t)tile * 4 * blockDim.x + 4 * threadIdx.x;
    float acc[4][9];
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 9; ++j) acc[c][j] = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int k = 0; k < R; k += PIPE) {
        f4 v[PIPE];
#pragma unroll
        for (int i = 0; i < PIPE; ++i) {
            const f4* q = reinterpret_cast<const f4*>(p + (size_t)min(k + i, R - 1) * npad);
            v[i] = NT ? __builtin_nontemporal_load(q) : *q;
        }
#pragma unroll
        for (int i = 0; i < PIPE; ++i) {
            float qv[12];
            if (LDSQ) {
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float4 q4 = *reinterpret_cast<const float4*>(&Qs[(k + i) * 12 + 4 * t]);
                    qv[4 * t] = q4.x; qv[4 * t + 1] = q4.y; qv[4 * t + 2] = q4.z; qv[4 * t + 3] = q4.w;
                }
            } else {
#pragma unroll
                for (int t = 0; t < 12; ++t) qv[t] = 1e-3f * t + (float)k;
            }
#pragma unroll
            for (int j = 0; j < 9; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c][j] = fmaf(v[i][c], qv[j], acc[c][j]);
        }
    }
    float t = 0;
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 9; ++j) t += acc[c][j];
    if (t == 123.456f) out[0] = t;
}

// one wave per item: the wave walks `ntile` 256-cell tiles of its env, streaming `R` rows per tile with 36 FMAs per
// float4 against 12 values that arrive through SCALAR loads (one Q row per streamed row), 4 waves / SIMD.
template <int PIPE, bool SCALARQ>
__global__ __launch_bounds__(64, 4) void k_wave_item(const float* __restrict__ base, size_t slot, int npad, int ntile,
                                                     const int* __restrict__ rows, const float* __restrict__ q, float* out) {
    __shared__ __attribute__((aligned(16))) float Qs[368 * 12];
    const int env = blockIdx.x, lane = threadIdx.x;
    const int R = rows[env];
    const float* __restrict__ qrow = q + (size_t)env * 368 * 12;
    if (!SCALARQ) { for (int i = lane; i < 368 * 12; i += 64) Qs[i] = qrow[i]; __syncthreads(); }
    float tot = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    for (int t = 0; t < ntile; ++t) {
        const float* p = base + (size_t)env * slot + (size_t)((t * 3) % 10) * 256 + 4 * lane;
        float acc[4][9];
        for (int c = 0; c < 4; ++c) for (int j = 0; j < 9; ++j) acc[c][j] = 0.f;
        for (int k = 0; k < R; k += PIPE) {
            f4 v[PIPE];
#pragma unroll
            for (int i = 0; i < PIPE; ++i) v[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + (size_t)min(k + i, R - 1) * npad));
#pragma unroll
            for (int i = 0; i < PIPE; ++i) {
                const int kk = min(k + i, R - 1);
                float qv[12];
#pragma unroll
                for (int j = 0; j < 12; ++j) qv[j] = SCALARQ ? qrow[kk * 12 + j] : Qs[kk * 12 + j];
#pragma unroll
                for (int j = 0; j < 9; ++j)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[c][j] = fmaf(v[i][c], qv[j], acc[c][j]);
            }
        }
        for (int c = 0; c < 4; ++c) for (int j = 0; j < 9; ++j) tot += acc[c][j];
    }
    if (tot == 123.456f) out[0] = tot;
}

int main() {
    const int B = 4096, npad = 2560, rcap = 360;
    const size_t slot = (size_t)rcap * npad;
    const size_t total = (size_t)B * slot;
    float* d; float* out; int* rows;
    CK(hipMalloc(&d, total * 4)); CK(hipMalloc(&out, 4)); CK(hipMalloc(&rows, B * 4));
    CK(hipMemset(d, 0, total * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto launch, double bytes, const char* name) {
        launch(); CK(hipDeviceSynchronize());
        float best = 1e30f, sum = 0;
        for (int it = 0; it < 5; ++it) {
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; sum += ms;
        }
        printf("%-44s %8.3f ms (best) %8.1f GB/s   avg %8.1f GB/s\n", name, best, bytes / best / 1e6, bytes / (sum / 5) / 1e6);
    };
    time_it([&] { hipLaunchKernelGGL(k_flat, dim3(256 * 8), dim3(256), 0, 0, (const float4*)d, total / 4, out); }, total * 4.0, "flat read 15 GB, 2048x256");
    time_it([&] { hipLaunchKernelGGL(k_flat, dim3(256 * 32), dim3(256), 0, 0, (const float4*)d, total / 4, out); }, total * 4.0, "flat read 15 GB, 8192x256");
    {
        float* q; CK(hipMalloc(&q, (size_t)B * 368 * 12 * 4)); CK(hipMemset(q, 0, (size_t)B * 368 * 12 * 4));
        std::vector<int> h(B);
        double rs = 0;
        for (int e = 0; e < B; ++e) { h[e] = (int)(0.6 * 5.9 * ((e % 40) + 0.5)) + 1; rs += h[e]; }
        CK(hipMemcpy(rows, h.data(), B * 4, hipMemcpyHostToDevice));
        const int ntile = 6;
        const double bytes = rs * ntile * 256 * 4.0;
        time_it([&] { hipLaunchKernelGGL((k_wave_item<2, true>), dim3(B), dim3(64), 0, 0, d, slot, npad, ntile, rows, q, out); }, bytes, "wave-per-item scalarQ pipe2 (6 tiles x 0.6R rows)");
        time_it([&] { hipLaunchKernelGGL((k_wave_item<4, true>), dim3(B), dim3(64), 0, 0, d, slot, npad, ntile, rows, q, out); }, bytes, "wave-per-item scalarQ pipe4");
        time_it([&] { hipLaunchKernelGGL((k_wave_item<8, true>), dim3(B), dim3(64), 0, 0, d, slot, npad, ntile, rows, q, out); }, bytes, "wave-per-item scalarQ pipe8");
        time_it([&] { hipLaunchKernelGGL((k_wave_item<4, false>), dim3(B), dim3(64), 0, 0, d, slot, npad, ntile, rows, q, out); }, bytes, "wave-per-item ldsQ(17KB) pipe4");
    }
    for (int mode = 0; mode < 2; ++mode) {
        std::vector<int> h(B);
        double rs = 0;
        for (int e = 0; e < B; ++e) { h[e] = mode == 0 ? 176 : (int)(5.9 * ((e % 40) + 0.5)) + 1; rs += h[e]; }
        CK(hipMemcpy(rows, h.data(), B * 4, hipMemcpyHostToDevice));
        const double bytes = rs * 2500 * 4.0;  // useful cells only, like the engine's accounting
        for (int T : {128, 320, 640}) {
            const int tiles = 640 / T;
            char name[128];
            snprintf(name, sizeof name, "stream %s T=%d pipe4", mode ? "staggered-R" : "uniform-R176", T);
            time_it([&] { hipLaunchKernelGGL((k_stream<4, false>), dim3(B * tiles), dim3(T), 0, 0, d, slot, npad, tiles, rows, out); }, bytes, name);
            snprintf(name, sizeof name, "stream %s T=%d pipe4 nt", mode ? "staggered-R" : "uniform-R176", T);
            time_it([&] { hipLaunchKernelGGL((k_stream<4, true>), dim3(B * tiles), dim3(T), 0, 0, d, slot, npad, tiles, rows, out); }, bytes, name);
            snprintf(name, sizeof name, "stream+36fma(reg) %s T=%d pipe4", mode ? "staggered-R" : "uniform-R176", T);
            time_it([&] { hipLaunchKernelGGL((k_stream_fma<4, false, false>), dim3(B * tiles), dim3(T), 0, 0, d, slot, npad, tiles, rows, out); }, bytes, name);
            snprintf(name, sizeof name, "stream+36fma(lds) %s T=%d pipe4", mode ? "staggered-R" : "uniform-R176", T);
            time_it([&] { hipLaunchKernelGGL((k_stream_fma<4, false, true>), dim3(B * tiles), dim3(T), 0, 0, d, slot, npad, tiles, rows, out); }, bytes, name);
            snprintf(name, sizeof name, "stream+36fma(lds) %s T=%d pipe4 nt", mode ? "staggered-R" : "uniform-R176", T);
            time_it([&] { hipLaunchKernelGGL((k_stream_fma<4, true, true>), dim3(B * tiles), dim3(T), 0, 0, d, slot, npad, tiles, rows, out); }, bytes, name);
        }
    }
    return 0;
}
