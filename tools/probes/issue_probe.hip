// Issue cost of the vector instructions the step kernels are made of, on gfx950: cycles per wave-instruction for ONE wave per SIMD
// (issue interval of a stream of independent instructions of one kind) and the aggregate per-SIMD rate with W waves per SIMD.
// Cost model behind the instruction cuts of round 4 (profiles/r04_issue_probe.txt).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/issue_probe.hip -o tools/probes/issue_probe && ./tools/probes/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

enum Op { FMAC, FMAC_DPP, PK_FMA_V, PK_FMA_S, FMA64, READLANE, PK_MIN_U16, CNDMASK, SAD, DS_READ_BC, MOV_DPP, FMAC_SGPR, N_OPS };
static const char* kNames[N_OPS] = {"v_fmac_f32 (vgpr)", "v_fmac_f32_dpp row_newbcast", "v_pk_fma_f32 (vgpr pair)", "v_pk_fma_f32 (sgpr coefficient, op_sel)",
                                    "v_fma_f64", "v_readlane_b32", "v_pk_min_u16", "v_cndmask_b32", "v_sad_u32", "ds_read_b32 (broadcast address)",
                                    "v_mov_b32_dpp row_newbcast", "v_fmac_f32 (sgpr coefficient)"};

template <int OP>
__global__ __launch_bounds__(256) void k_issue(float* out, const float* in, int iters, unsigned long long* cyc) {
    __shared__ float lds[64];
    const int lane = threadIdx.x & 63;
    if (threadIdx.x < 64) lds[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
    float a0 = in[lane], a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float u = in[64 + lane], q = in[128 + (lane & 15)];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    f2 uu = {u, u + 1.f}, qq = {q, q};
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, du = u, dq = q;
    const float qs = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, q)));
    int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    unsigned x0 = lane, x1 = lane + 1, x2 = lane + 2, x3 = lane + 3, x4 = lane + 4, x5 = lane + 5, x6 = lane + 6, x7 = lane + 7;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)lds;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long m0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (OP == FMAC) {
            REP4(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q), "v"(u));)
        } else if (OP == FMAC_SGPR) {
            REP4(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                              "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(qs), "v"(u));)
        } else if (OP == FMAC_DPP) {
            REP4(asm volatile("v_fmac_f32_dpp %0, %8, %9 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f32_dpp %2, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f32_dpp %4, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                              "v_fmac_f32_dpp %6, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q), "v"(u));)
        } else if (OP == PK_FMA_V) {
            REP4(asm volatile("v_pk_fma_f32 %0, %8, %9, %0\n v_pk_fma_f32 %1, %8, %9, %1\n v_pk_fma_f32 %2, %8, %9, %2\n v_pk_fma_f32 %3, %8, %9, %3\n"
                              "v_pk_fma_f32 %4, %8, %9, %4\n v_pk_fma_f32 %5, %8, %9, %5\n v_pk_fma_f32 %6, %8, %9, %6\n v_pk_fma_f32 %7, %8, %9, %7\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(uu), "v"(qq));)
        } else if (OP == PK_FMA_S) {
            // coefficient pair in SGPRs; op_sel picks the low (or high) dword for BOTH halves of the product
            const f2 qsp = {qs, qs + 1.f};
            REP4(asm volatile("v_pk_fma_f32 %0, %8, %9, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %8, %9, %1 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                              "v_pk_fma_f32 %2, %8, %9, %2 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %3, %8, %9, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                              "v_pk_fma_f32 %4, %8, %9, %4 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %5, %8, %9, %5 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                              "v_pk_fma_f32 %6, %8, %9, %6 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %7, %8, %9, %7 op_sel:[0,1,0] op_sel_hi:[1,1,1]\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(uu), "s"(qsp));)
        } else if (OP == FMA64) {
            REP4(asm volatile("v_fma_f64 %0, %8, %9, %0\n v_fma_f64 %1, %8, %9, %1\n v_fma_f64 %2, %8, %9, %2\n v_fma_f64 %3, %8, %9, %3\n"
                              "v_fma_f64 %4, %8, %9, %4\n v_fma_f64 %5, %8, %9, %5\n v_fma_f64 %6, %8, %9, %6\n v_fma_f64 %7, %8, %9, %7\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(du), "v"(dq));)
        } else if (OP == READLANE) {
            REP4(asm volatile("v_readlane_b32 %0, %4, 1\n v_readlane_b32 %1, %4, 2\n v_readlane_b32 %2, %4, 3\n v_readlane_b32 %3, %4, 4\n"
                              "v_readlane_b32 %0, %4, 5\n v_readlane_b32 %1, %4, 6\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 8\n"
                              : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(x0));)
        } else if (OP == PK_MIN_U16) {
            REP4(asm volatile("v_pk_min_u16 %0, %0, %8\n v_pk_min_u16 %1, %1, %8\n v_pk_min_u16 %2, %2, %8\n v_pk_min_u16 %3, %3, %8\n"
                              "v_pk_min_u16 %4, %4, %8\n v_pk_min_u16 %5, %5, %8\n v_pk_min_u16 %6, %6, %8\n v_pk_min_u16 %7, %7, %8\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(lane));)
        } else if (OP == CNDMASK) {
            REP4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(lane) : "vcc");)
        } else if (OP == SAD) {
            REP4(asm volatile("v_sad_u32 %0, %0, %8, %8\n v_sad_u32 %1, %1, %8, %8\n v_sad_u32 %2, %2, %8, %8\n v_sad_u32 %3, %3, %8, %8\n"
                              "v_sad_u32 %4, %4, %8, %8\n v_sad_u32 %5, %5, %8, %8\n v_sad_u32 %6, %6, %8, %8\n v_sad_u32 %7, %7, %8, %8\n"
                              : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(lane));)
        } else if (OP == DS_READ_BC) {
            REP4(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
                              "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n s_waitcnt lgkmcnt(0)\n"
                              : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7) : "v"(lbase));)
        } else if (OP == MOV_DPP) {
            REP4(asm volatile("v_mov_b32_dpp %0, %8 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(q));)
        }
    }
    const unsigned long long m1 = __builtin_amdgcn_s_memtime();
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p1[1] + p2[0] + p3[1] + p4[0] + p5[1] + p6[0] + p7[1] +
              (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(s0 + s1 + s2 + s3) + (float)(x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7);
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lane == 0) { cyc[2 * gw] = m1 - m0; cyc[2 * gw + 1] = t1 - t0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int OP>
void run(float* out, const float* in, unsigned long long* cyc, int cus) {
    const int iters = 2000, per_iter = 32;
    printf("%-42s", kNames[OP]);
    for (int W : {1, 2, 4, 5, 8}) {
        const int blocks = cus * W;  // 256-thread blocks: one wave per SIMD each
        hipEvent_t a, b;
        hipEventCreate(&a); hipEventCreate(&b);
        hipLaunchKernelGGL(k_issue<OP>, dim3(blocks), dim3(256), 0, 0, out, in, 10, cyc);
        hipDeviceSynchronize();
        hipEventRecord(a);
        hipLaunchKernelGGL(k_issue<OP>, dim3(blocks), dim3(256), 0, 0, out, in, iters, cyc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0.f;
        hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(2 * blocks * 4);
        hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sm = 0, sc = 0;
        for (int i = 0; i < blocks * 4; ++i) { sm += h[2 * i]; sc += h[2 * i + 1]; }
        sm /= blocks * 4; sc /= blocks * 4;
        const double n = (double)iters * per_iter;
        // per-wave interval (memtime ticks per instruction of ONE wave) and per-SIMD aggregate (ns per instruction per SIMD)
        printf("  W=%d: %.2f tick/inst/wave (%.2f clk) | %.3f ns/inst/SIMD", W, sm / n, sc / n, ms * 1e6 / (n * W));
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clock %d kHz, wall clock rate %d kHz\n", p.name, p.multiProcessorCount, p.clockRate, p.clockInstructionRate);
    float *in, *out;
    unsigned long long* cyc;
    hipMalloc(&in, 4096);
    hipMalloc(&out, (size_t)256 * 8 * 256 * 4 + 4096);
    hipMalloc(&cyc, (size_t)256 * 8 * 4 * 16 + 4096);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 1e-3f * (i % 17);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    const int cus = p.multiProcessorCount;
    run<FMAC>(out, in, cyc, cus);
    run<FMAC_SGPR>(out, in, cyc, cus);
    run<FMAC_DPP>(out, in, cyc, cus);
    run<PK_FMA_V>(out, in, cyc, cus);
    run<PK_FMA_S>(out, in, cyc, cus);
    run<FMA64>(out, in, cyc, cus);
    run<READLANE>(out, in, cyc, cus);
    run<PK_MIN_U16>(out, in, cyc, cus);
    run<CNDMASK>(out, in, cyc, cus);
    run<SAD>(out, in, cyc, cus);
    run<MOV_DPP>(out, in, cyc, cus);
    run<DS_READ_BC>(out, in, cyc, cus);
    return 0;
}
