// Do VALU instructions of one wave overlap with MFMAs of another wave on the same SIMD (gfx950)?
// 8 waves per workgroup = 2 per SIMD: waves 0-3 run a 16x16x4 f32 MFMA stream, waves 4-7 a VALU stream (plain FMA, DPP FMA or
// LDS reads); each role is timed alone and together.  hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// mode bit0: MFMA waves active, bit1: second-role waves active; kind: 0 = v_fma_f32, 1 = DPP fmac, 2 = ds_read_b128, 3 = MFMA too
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int mode, int kind, int gap, int prio) {
    __shared__ float lds[4096];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    lds[threadIdx.x] = threadIdx.x; lds[threadIdx.x + 512] = 1.f;
    __syncthreads();
    const bool mf = w < 4;
    if (mf ? !(mode & 1) : !(mode & 2)) return;
    if (!mf && prio == 1) __builtin_amdgcn_s_setprio(3);
    if (mf && prio == 2) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    float s = 0.f;
    if (mf || kind == 3) {
        float a = l * 0.001f, b = 1.0f + l * 1e-6f;
        f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0);
                if (gap >= 1) asm volatile("s_nop 7"); if (gap >= 2) asm volatile("s_nop 7"); if (gap >= 3) asm volatile("s_nop 7");
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
                if (gap >= 1) asm volatile("s_nop 7"); if (gap >= 2) asm volatile("s_nop 7"); if (gap >= 3) asm volatile("s_nop 7");
            }
        }
        s = c0[0] + c1[1];
    } else if (kind == 0) {
        float x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, m = 1.0001f, ad = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {       // 128 independent-ish FMAs = 16 MFMA slots' worth of time (4 cycles each)
                x0 = fmaf(x0, m, ad); x1 = fmaf(x1, m, ad); x2 = fmaf(x2, m, ad); x3 = fmaf(x3, m, ad);
            }
        }
        s = x0 + x1 + x2 + x3;
    } else if (kind == 1) {
        float x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, m = 1.0001f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x0) : "v"(m), "v"(x1));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(x1) : "v"(m), "v"(x2));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(x2) : "v"(m), "v"(x3));
                asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(x3) : "v"(m), "v"(x0));
            }
        }
        s = x0 + x1 + x2 + x3;
    } else {
        const float4* p = reinterpret_cast<const float4*>(lds) + l;
        float4 acc = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(size_t)(l * 16 + ((u * 1024) & 8191))) : "memory");
                acc.x += v[0]; acc.y += v[1]; acc.z += v[2]; acc.w += v[3];
            }
        }
        s = acc.x + acc.y + acc.z + acc.w;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (l == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* d; unsigned long long* c;
    hipMalloc(&d, 256 * 512 * 4); hipMalloc(&c, 256 * 8 * 8);
    const int iters = 2000;
    const char* kinds[] = {"v_fma_f32", "v_fmac_dpp", "ds_read_b128+4 adds", "mfma (second pair of waves)"};
    for (int var = 0; var < 6; ++var) {
        const int gap = var < 4 ? var : 0, prio = var == 4 ? 1 : var == 5 ? 2 : 0;
        printf("--- MFMA waves: %d x s_nop 7 after each MFMA; setprio 3 on %s\n", gap, prio == 1 ? "the other waves" : prio == 2 ? "the MFMA waves" : "nobody");
        for (int kind = 0; kind < 4; ++kind) {
            double res[4][2] = {};
            for (int mode = 1; mode <= 3; ++mode) {
                hipMemset(c, 0, 256 * 8 * 8);
                k<<<256, 512>>>(d, c, iters, mode, kind, gap, prio);
                hipDeviceSynchronize();
                std::vector<unsigned long long> h(256 * 8);
                hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
                double a = 0, b = 0;
                for (int i = 0; i < 256; ++i) for (int w = 0; w < 8; ++w) (w < 4 ? a : b) += h[i * 8 + w] / (256.0 * 4);
                res[mode][0] = a; res[mode][1] = b;
            }
            printf("%-28s: mfma alone %.0f cyc | other alone %.0f | together: mfma %.0f (x%.2f), other %.0f (x%.2f)\n", kinds[kind],
                   res[1][0], res[2][1], res[3][0], res[3][0] / res[1][0], res[3][1], res[3][1] / res[2][1]);
        }
    }
    return 0;
}
