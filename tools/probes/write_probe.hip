// Calibration of rocprofv3's WRITE_SIZE (KiB) on gfx950 against known byte counts, in the three write patterns of the
// fused step kernel (MI355X_MICROARCH.md, HBM section: "WRITE_SIZE uncalibrated: calibrate on a known byte count in
// your own access pattern"):
//   k_store      plain global_store_dwordx4, 16 B per lane, 1 KiB per wave instruction (new rows of U, temporal)
//   k_store_nt   the same with the non-temporal hint the kernel uses for the appended rows
//   k_atomic     one no-return float atomic add per cell, lane i touching cells 4 i + c for c = 0..3 (four instructions of 64
//                lanes at a 16-byte stride): the kernel's diag / mean update
//   k_atomic_seq one atomic per cell with consecutive lanes on consecutive cells (what a transposed epilogue would do)
// Each kernel touches every byte of a buffer of `mb` MiB exactly once (larger than the 256 MiB Infinity Cache).
// Build: hipcc --offload-arch=gfx950 -O3 write_probe.hip -o write_probe
// Run:   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out -- ./write_probe 1024      (then tools/pmc_summary-style mean per kernel)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_store(float* p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 v = {1.f, 2.f, 3.f, (float)i};
        *reinterpret_cast<f4*>(p + 4 * i) = v;
    }
}
__global__ void k_store_nt(float* p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 v = {1.f, 2.f, 3.f, (float)i};
        __builtin_nontemporal_store(v, reinterpret_cast<f4*>(p + 4 * i));
    }
}
__global__ void k_atomic(float* p, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride)
#pragma unroll
        for (int c = 0; c < 4; ++c) unsafeAtomicAdd(p + 4 * i + c, 1.0f);
}
__global__ void k_atomic_seq(float* p, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) unsafeAtomicAdd(p + i, 1.0f);
}

int main(int argc, char** argv) {
    const size_t mb = argc > 1 ? atol(argv[1]) : 1024;
    const size_t n = mb * 1024 * 1024 / 4, n4 = n / 4;
    float* p;
    CK(hipMalloc(&p, n * 4));
    CK(hipMemset(p, 0, n * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int grid = 256 * 8, block = 256;
    const char* names[4] = {"k_store", "k_store_nt", "k_atomic", "k_atomic_seq"};
    for (int k = 0; k < 4; ++k)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            if (k == 0) hipLaunchKernelGGL(k_store, dim3(grid), dim3(block), 0, 0, p, n4);
            if (k == 1) hipLaunchKernelGGL(k_store_nt, dim3(grid), dim3(block), 0, 0, p, n4);
            if (k == 2) hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(block), 0, 0, p, n4);
            if (k == 3) hipLaunchKernelGGL(k_atomic_seq, dim3(grid), dim3(block), 0, 0, p, n);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 2) printf("%-13s %zu MiB written once: %.3f ms = %.0f GB/s\n", names[k], mb, ms, n * 4 / ms / 1e6);
        }
    CK(hipFree(p));
    return 0;
}
