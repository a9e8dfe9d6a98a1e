// Does s_setprio let one wave of a SIMD run at its lone-wave pace while four others compete for the issue slots?
// W waves per SIMD run the same stream of 16 independent DPP FMAs + 4 scalar instructions per iteration; in mode 1 the first wave of
// every SIMD raises its priority to 3.  Prints the elapsed clocks of the favoured wave and of the others (profiles/r04_prio_probe.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, int mode, unsigned long long* cyc, int blocks_per_cu_round) {
    const int lane = threadIdx.x & 63;
    float a0 = in[lane], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float u = in[64 + lane], q = in[128 + (lane & 15)];
    // the first block that lands on a CU is the favoured one: blocks are dispatched round-robin, block b < 256 is the first of its CU
    // (mode 2: the LAST block of a CU -- the youngest waves of their SIMDs -- is the favoured one: does the priority beat the age order?)
    const bool fav = (mode == 1 && (int)blockIdx.x < blocks_per_cu_round) || (mode == 2 && (int)blockIdx.x >= 4 * blocks_per_cu_round);
    if (fav) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                     "v_fmac_f32_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q), "v"(u));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lane == 0) cyc[gw] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, W = 5, iters = 4000;
    float *in, *out; unsigned long long* cyc;
    (void)hipMalloc(&in, 4096); (void)hipMalloc(&out, (size_t)cus * W * 256 * 4); (void)hipMalloc(&cyc, (size_t)cus * W * 4 * 8);
    std::vector<float> h(1024, 1e-3f); (void)hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(cus * W), dim3(256), 0, 0, out, in, iters, mode, cyc, cus);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> c((size_t)cus * W * 4);
        (void)hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
        double fav = 0, rest = 0; int nf = 0, nr = 0;
        for (int b = 0; b < cus * W; ++b) for (int w = 0; w < 4; ++w) { if (b < cus) { fav += c[b * 4 + w]; ++nf; } else { rest += c[b * 4 + w]; ++nr; } }
        printf("mode %d (%s): first block of a CU %.0f clocks per wave (%.2f per instruction), the other four %.0f (%.2f)\n", mode,
               mode == 2 ? "LAST block at priority 3" : (mode ? "first block at priority 3" : "equal priority"), fav / nf, fav / nf / (16.0 * iters), rest / nr, rest / nr / (16.0 * iters));
        for (int rnd = 0; rnd < W; ++rnd) {
            double sum = 0; for (int b = rnd * cus; b < (rnd + 1) * cus; ++b) for (int w = 0; w < 4; ++w) sum += c[b * 4 + w];
            printf("   block round %d: %.2f clocks per instruction\n", rnd, sum / (cus * 4.0) / (16.0 * iters));
        }
    }
    return 0;
}
