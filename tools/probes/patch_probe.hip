// Microbenchmark behind the column-patch layout: the row stream of one env step
// with no arithmetic, same bytes, same occupancy (2-wave workgroups, 20 KB of LDS each -> 8 per CU, 4 waves per SIMD),
// 4096 items x 6 units x R stored rows per unit, against four address patterns:
//   band     rows of Npad floats (50x50 grid, row-major band tiles): a unit = 5 grid-row segments of 26 cells at a stride of
//            50 cells, 8 B per lane through buffer loads (masked lanes out of range) -- the k_step_factor<RECT> pattern
//   patch    columns as compact 25x26 patches (656 floats): a unit = 512 consecutive bytes of the column, 8 B per lane
//   patch16  the same bytes, 16 B per lane: 1 KiB per wave instruction, half the instructions (units of 256 cells)
//   colmajor patch layout, but a wave walks ALL units of a column before the next column (one DRAM page at a time)
// Reports GB/s of requested bytes; run under rocprofv3 --pmc FETCH_SIZE for the fetched bytes of each pattern (the
// FETCH_SIZE calibration on this kernel's own access pattern).
// Build: hipcc --offload-arch=gfx950 -O3 patch_probe.hip -o patch_probe ; run on the GPU box: ./patch_probe [items] [rows]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int kRankCap = 360, kNpad = 2560, kPatch = 656, kUnits = 6, kGroup = 16;

template <int MODE>  // 0 band, 1 patch, 2 patch16, 3 colmajor
__global__ __launch_bounds__(128, 4) void k_probe(const float* __restrict__ base, int n_items, int rows, int rank, float* out) {
    extern __shared__ float lds[];
    const int item = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (item >= n_items) return;
    const size_t slot = (size_t)kRankCap * (MODE == 0 ? kNpad : kPatch);
    const float* env = base + (size_t)item * slot;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    if (MODE == 3) {
        // column outer, unit inner: wave w takes the columns w, w + 2, ... and reads the whole patch of each (6 x 512 B)
        for (int k0 = wave; k0 < rows; k0 += 2 * 4) {
            float2 v[4][kUnits];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = ((k0 + 2 * i) * 7 + item) % rank;
                const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(env + (size_t)k * kPatch), 0, kPatch * 4, 0x00020000);
#pragma unroll
                for (int u = 0; u < kUnits; ++u) {
                    const unsigned off = (unsigned)(u * 128 + 2 * lane) * 4u;
                    v[i][u] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, (k0 + 2 * i < rows) ? off : 0xffffffffu, 0, 2));
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < kUnits; ++u) { acc0 += v[i][u].x; acc1 += v[i][u].y; }
        }
    } else {
        constexpr int UN = (MODE == 2) ? kUnits / 2 : kUnits;
        for (int u = wave; u < UN; u += 2) {
            unsigned off;
            if (MODE == 0) {  // 26-cell segments of 5 grid rows: lane -> (row = lane / 13, pair = lane % 13), lanes 65.. masked
                const int rr = lane / 13, cc = lane - rr * 13;
                off = (unsigned)(((u * 5 + rr) * 50 + 12 + 2 * cc) * 4);
                if (rr >= 5) off = 0xffffffffu;
            } else if (MODE == 1) {
                off = (unsigned)(u * 128 + 2 * lane) * 4u;
            } else {
                off = (unsigned)(u * 256 + 4 * lane) * 4u;
            }
            if (MODE != 0 && off + (MODE == 2 ? 16 : 8) > kPatch * 4) off = 0xffffffffu;
            for (int k0 = 0; k0 < rows; k0 += kGroup) {
                float4 v[kGroup];
#pragma unroll
                for (int i = 0; i < kGroup; ++i) {
                    const int k = ((k0 + i) * 7 + item) % rank;
                    const size_t stride = (MODE == 0) ? kNpad : kPatch;
                    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(env + (size_t)k * stride), 0, (int)stride * 4, 0x00020000);
                    const unsigned o = (k0 + i < rows) ? off : 0xffffffffu;
                    if (MODE == 2) v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 2));
                    else { const float2 t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 2)); v[i] = make_float4(t.x, t.y, 0.f, 0.f); }
                }
#pragma unroll
                for (int i = 0; i < kGroup; ++i) { acc0 += v[i].x; acc1 += v[i].y; acc2 += v[i].z; acc3 += v[i].w; }
            }
        }
    }
    if (acc0 + acc1 + acc2 + acc3 == 123.456f) out[0] = acc0 + lds[0];
}

template <int MODE>
void run(const char* name, const float* base, int n_items, int rows, int rank, float* out, double bytes) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 20480));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_probe<MODE>, dim3(n_items), dim3(128), 20480, 0, base, n_items, rows, rank, out);
    CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0.f;
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_probe<MODE>, dim3(n_items), dim3(128), 20480, 0, base, n_items, rows, rank, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best; tot += ms;
    }
    printf("%-9s %7.4f ms avg %7.4f ms best  requested %.1f MB -> %7.1f GB/s (best %7.1f)\n", name, tot / reps, best, bytes / 1e6, bytes / (tot / reps * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    const int n_items = argc > 1 ? atoi(argv[1]) : 4096, rows = argc > 2 ? atoi(argv[2]) : 32, rank = argc > 3 ? atoi(argv[3]) : 117;
    const size_t floats = (size_t)n_items * kRankCap * kNpad;
    float *base, *out;
    CK(hipMalloc(&base, floats * 4));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(base, 0, floats * 4));
    // bytes requested per launch: band 65 lanes .. -> 5 x 13 = 65 > 64: lanes 0..63 cover 4 rows + 12 pairs: count the lanes
    const double lanes_band = 64.0, per_row = 8.0;
    const double band = (double)n_items * kUnits * rows * lanes_band * per_row;
    const double patch = (double)n_items * rows * (kPatch * 4.0);  // every column's whole patch once (656 floats; the last unit is partial)
    printf("items %d, rows per unit %d of rank %d, units %d\n", n_items, rows, rank, kUnits);
    run<0>("band", base, n_items, rows, rank, out, band);
    run<1>("patch", base, n_items, rows, rank, out, patch);
    run<2>("patch16", base, n_items, rows, rank, out, patch);
    run<3>("colmajor", base, n_items, rows, rank, out, patch);
    return 0;
}
