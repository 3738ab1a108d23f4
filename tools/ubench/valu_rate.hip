// issue cost (cycles per wave64 instruction, one wave per SIMD) of the VALU forms the kernel's index/epilogue code uses (gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
#define BENCH(NAME, ASM)                                                                          \
    __global__ void NAME(unsigned* out, unsigned long long* cyc, int iters) {                     \
        unsigned a = threadIdx.x + 1, b = threadIdx.x * 3 + 7, c = 5, d = 9;                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                               \
        for (int it = 0; it < iters; ++it) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 16; ++u) {                                      \
                asm volatile(ASM : "+v"(a) : "v"(b), "v"(c));                                     \
                asm volatile(ASM : "+v"(b) : "v"(c), "v"(d));                                     \
                asm volatile(ASM : "+v"(c) : "v"(d), "v"(a));                                     \
                asm volatile(ASM : "+v"(d) : "v"(a), "v"(b));                                     \
            }                                                                                     \
        }                                                                                         \
        const unsigned long long t1 = __builtin_readcyclecounter();                               \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                          \
        out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;                                       \
    }
BENCH(k_mul_lo, "v_mul_lo_u32 %0, %1, %2")
BENCH(k_mul_u24, "v_mul_u32_u24 %0, %1, %2")
BENCH(k_mad_u24, "v_mad_u32_u24 %0, %1, %2, %0")
BENCH(k_add, "v_add_u32 %0, %1, %2")
BENCH(k_lshl_add, "v_lshl_add_u32 %0, %1, 2, %2")
BENCH(k_fma, "v_fma_f32 %0, %1, %2, %0")
BENCH(k_med3, "v_med3_f32 %0, %1, %2, %0")
BENCH(k_cndmask, "v_cndmask_b32 %0, %1, %2, vcc")
BENCH(k_cmp, "v_cmp_le_f32 vcc, %1, %2")
BENCH(k_exp, "v_exp_f32 %0, %1")
BENCH(k_readlane_like, "v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf")
template <class K> void run(const char* name, K k, unsigned* d, unsigned long long* c, int per) {
    const int iters = 2000;
    k<<<1024, 64>>>(d, c, iters); hipDeviceSynchronize();
    k<<<1024, 64>>>(d, c, iters); hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 1024; ++i) s += h[i];
    printf("%-22s %.2f cycles/instr\n", name, s / 1024 / iters / 64.0);
}
int main() {
    unsigned* d; unsigned long long* c;
    hipMalloc(&d, 1024 * 64 * 4); hipMalloc(&c, 1024 * 8);
    run("v_mul_lo_u32", k_mul_lo, d, c, 1); run("v_mul_u32_u24", k_mul_u24, d, c, 1); run("v_mad_u32_u24", k_mad_u24, d, c, 1);
    run("v_add_u32", k_add, d, c, 1); run("v_lshl_add_u32", k_lshl_add, d, c, 1);
    run("v_fma_f32", k_fma, d, c, 1); run("v_med3_f32", k_med3, d, c, 1); run("v_cndmask_b32", k_cndmask, d, c, 1);
    run("v_cmp_le_f32", k_cmp, d, c, 1); run("v_exp_f32", k_exp, d, c, 1); run("v_mov_dpp newbcast", k_readlane_like, d, c, 1);
    return 0;
}
