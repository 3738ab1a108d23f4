// Same probe as coissue.hip with the MFMA stream on the bf16 matrix path (v_mfma_f32_16x16x32_bf16, gfx950): does a VALU
// wave on the same SIMD keep its rate next to it, and what is the instruction's issue period?
// hipcc --offload-arch=gfx950 -O3 coissue_bf16.hip -o coissue_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int kind>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int mode) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool mf = w < 4;
    if (mf ? !(mode & 1) : !(mode & 2)) return;
    const unsigned long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    if (mf) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(l * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
        f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (kind == 0) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
                } else {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(s + 1.f, 2.f, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(s + 1.f, 2.f, c1, 0, 0, 0);
                }
            }
        }
        s = c0[0] + c1[1];
    } else {
        float x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, m = 1.0001f, ad = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(m), "v"(ad));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(m), "v"(ad));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(m), "v"(ad));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(m), "v"(ad));
            }
        }
        s = x0 + x1 + x2 + x3;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 256 * 8 * 8);
    const int iters = 2000;
    const char* names[] = {"v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32"};
    for (int kind = 0; kind < 2; ++kind) {
        double res[4][2] = {};
        for (int mode = 1; mode <= 3; ++mode) {
            hipMemset(c, 0, 256 * 8 * 8);
            if (kind == 0) k<0><<<256, 512>>>(d, c, iters, mode); else k<1><<<256, 512>>>(d, c, iters, mode);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(256 * 8);
            hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
            double a = 0, b = 0;
            for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[i * 8 + w] / (256.0 * 4);
            res[mode][0] = a; res[mode][1] = b;
        }
        const double per = res[1][0] / (iters * 16.0);
        printf("%-26s: %.1f cycles / MFMA alone (%.0f FLOP/cycle/SIMD) | 128-FMA VALU wave alone %.0f cyc | together: mfma x%.2f, valu x%.2f (valu loses %.1f cycles per MFMA issued)\n",
               names[kind], per, (kind == 0 ? 16384.0 : 2048.0) / per, res[2][1] / iters, res[3][0] / res[1][0], res[3][1] / res[2][1],
               (res[3][1] - res[2][1]) / (res[3][1] / per));
    }
    return 0;
}
