#include <hip/hip_runtime.h>
template <int J> __device__ __forceinline__ void fmac_bc(float& acc, float q, float u) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(q), "v"(u), "n"(J));
}
__global__ void k(const float* __restrict__ u, const float* __restrict__ q, float* out) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    lds[lane] = q[lane]; lds[64 + lane] = q[64 + lane];
    __syncthreads();
    float qa = lds[lane & 15], qb = lds[16 + (lane & 15)];
    float2 ua = reinterpret_cast<const float2*>(u)[lane], ub = reinterpret_cast<const float2*>(u)[64 + lane];
    float acc[2][9];
    for (int c = 0; c < 2; ++c) for (int j = 0; j < 9; ++j) acc[c][j] = 0.f;
#define ROW(uu, qq) \
    fmac_bc<0>(acc[0][0], qq, uu.x); fmac_bc<0>(acc[1][0], qq, uu.y); \
    fmac_bc<1>(acc[0][1], qq, uu.x); fmac_bc<1>(acc[1][1], qq, uu.y); \
    fmac_bc<8>(acc[0][8], qq, uu.x); fmac_bc<8>(acc[1][8], qq, uu.y);
    ROW(ua, qa) ROW(ub, qb)
    float t = 0; for (int c = 0; c < 2; ++c) for (int j = 0; j < 9; ++j) t += acc[c][j];
    out[lane] = t;
}
int main() {
    float *u, *q, *o; hipMalloc(&u, 1024); hipMalloc(&q, 1024); hipMalloc(&o, 256);
    float hu[256], hq[128], ho[64];
    for (int i = 0; i < 256; ++i) hu[i] = 0.01f * i; for (int i = 0; i < 128; ++i) hq[i] = 1.0f + i;
    hipMemcpy(u, hu, 1024, hipMemcpyHostToDevice); hipMemcpy(q, hq, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, u, q, o);
    hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        float ref = 0;
        for (int c = 0; c < 2; ++c) for (int j : {0, 1, 8}) ref += hq[j] * hu[2 * l + c] + hq[16 + j] * hu[128 + 2 * l + c];
        if (fabsf(ref - ho[l]) > 1e-4f * fabsf(ref)) { ++bad; if (bad < 4) printf("lane %d: %f vs %f\n", l, ho[l], ref); }
    }
    printf("bad %d\n", bad);
}
