// micro-benchmark: issue rate of v_fmac_f32 (plain) vs v_fmac_f32_dpp row_newbcast on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.001f + i;
    float c = out[threadIdx.x & 15], y = out[16 + (threadIdx.x & 63)];
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define F(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(c), "v"(y));
            REP16(F)
#undef F
        } else if (MODE == 1) {
#define F(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:" #i " row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(c), "v"(y));
            REP16(F)
#undef F
        } else if (MODE == 2) {
#define F(i) asm volatile("v_fmac_f32_dpp %0, %1, %2 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(c), "v"(y));
            REP16(F)
#undef F
        } else {
#define F(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&acc[(i)&14]) : "v"(*(double*)&acc[0]), "v"(*(double*)&acc[2]));
            REP16(F)
#undef F
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[64 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 1024 * 8); hipMemset(out, 0, 4096 * 4);
    const int iters = 1000;
    for (int waves = 1; waves <= 8; waves *= 2) {
        unsigned long long h[4];
        for (int m = 0; m < 4; ++m) {
            if (m == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves * 4), 0, 0, out, cyc, iters);
            if (m == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves * 4), 0, 0, out, cyc, iters);
            if (m == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64 * waves * 4), 0, 0, out, cyc, iters);
            if (m == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64 * waves * 4), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
            hipMemcpy(&h[m], cyc, 8, hipMemcpyDeviceToHost);
        }
        if (waves * 4 > 16) break;
        printf("waves/SIMD=%d  cycles per instr per wave: plain %.2f  dpp_newbcast %.2f  dpp_quadperm %.2f  pk_fma %.2f\n", waves,
               h[0] / (16.0 * iters), h[1] / (16.0 * iters), h[2] / (16.0 * iters), h[3] / (16.0 * iters));
    }
    return 0;
}
