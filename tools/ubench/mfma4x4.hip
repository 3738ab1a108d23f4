// layout + rate probe of v_mfma_f32_4x4x1_16b_f32 (gfx950).  hipcc --offload-arch=gfx950 -O3 mfma4x4.hip -o mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void layout(float* out) {
    const int l = threadIdx.x;
    // A_b[i] = 100*b + i + 1 ; B_b[j] = 10*(j+1)  -> D_b[i][j] = (100 b + i + 1) * 10 (j+1)
    const float a = 100.f * (l >> 2) + (l & 3) + 1.f, b = 10.f * ((l & 3) + 1);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
template <int NACC>
__global__ void rate(float* out, int iters) {
    const int l = threadIdx.x;
    float a = l * 0.001f, b = 1.0f + l * 1e-6f;
    f32x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + l] = s;
}
template <int NACC>
__global__ void rate16(float* out, int iters) {
    const int l = threadIdx.x;
    float a = l * 0.001f, b = 1.0f + l * 1e-6f;
    f32x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + l] = s;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    layout<<<1, 64>>>(d);
    std::vector<float> h(256); hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const float exp = (100.f * (l >> 2) + r + 1.f) * 10.f * ((l & 3) + 1);   // D_b[i = r][j = l & 3]
        if (h[l * 4 + r] != exp) { if (bad < 8) printf("lane %d reg %d: got %g exp %g\n", l, r, h[l * 4 + r], exp); ++bad; }
    }
    printf("layout D_b[r][l&3] at lane l=4b+j: %s\n", bad ? "MISMATCH" : "ok");
    const int iters = 2000, blocks = 256 * 4;    // 4 single-wave blocks per CU... one wave per SIMD
    auto rep = [&](const char* name, float ms, double flop_per_mfma, int nacc) {
        const double n = double(iters) * 16 * nacc * blocks * 4;   // 256 threads = 4 waves per block
        printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, n * flop_per_mfma / ms / 1e9);
    };
    rep("4x4x1 dependent chain (1 acc)", timeit([&] { rate<1><<<blocks, 256>>>(d, iters); }), 512, 1);
    rep("4x4x1 2 accs", timeit([&] { rate<2><<<blocks, 256>>>(d, iters); }), 512, 2);
    rep("4x4x1 4 accs", timeit([&] { rate<4><<<blocks, 256>>>(d, iters); }), 512, 4);
    rep("16x16x4 dependent chain (1 acc)", timeit([&] { rate16<1><<<blocks, 256>>>(d, iters); }), 2048, 1);
    rep("16x16x4 2 accs", timeit([&] { rate16<2><<<blocks, 256>>>(d, iters); }), 2048, 2);
    return 0;
}
