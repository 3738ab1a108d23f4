// mix_ceiling.hip -- what matrix-pipe utilisation does the INSTRUCTION MIX of the headline trajectory kernel admit on gfx950?
//
// score_kernel<3,2,4> (BASELINE configs[1]) issues per wave and U-Net pass, by its PMC set (profiles/r04zy_avenue_pmc.txt,
// 184 320 wave-passes per launch):   575 v_mfma_f32_16x16x4_f32, 1 214 other VALU (42 % floating point: DPP FMAs, med3, packed
// add / fma / mul; 58 % integer / move: address arithmetic, selects, v_readlane), 1 018 SALU, 390 LDS, 212 vector-memory loads.
// This kernel issues that multiset -- per "unit" of 16 MFMAs: 14 FP VALU, 20 integer / move VALU, 28 SALU, 11 LDS, 6 global
// loads, the instruction types in the proportions of the kernel's ISA -- with NOTHING else in the way: no workgroup barriers,
// no data dependences except the four rotating MFMA accumulator chains, one counter wait per unit, every wave identical, two
// 8-wave workgroups per CU like the real kernel.  Its matrix-pipe busy fraction is the ceiling of this instruction mix; the
// variants take classes of instructions out (integer VALU halved / removed, no SALU, no memory) or put 24 barriers per pass in.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mix_ceiling.hip -o /tmp/mix_ceiling && /tmp/mix_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MF(k) "v_mfma_f32_16x16x4_f32 %[c" #k "], %[a], %[b], %[c" #k "]\n\t"
// floating-point VALU, 14 per unit: 5 DPP FMAs (the time mix), 2 med3 (PReLU), 3 packed, 4 plain
#define FP_A "v_fmac_f32_dpp %[f0], %[m], %[f1] row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t" \
             "v_fmac_f32_dpp %[f1], %[m], %[f2] row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t" \
             "v_med3_f32 %[f3], %[f3], %[m], %[f0]\n\t" \
             "v_pk_add_f32 %[p0], %[p0], %[p1]\n\t" \
             "v_add_f32 %[f2], %[f2], %[m]\n\t" \
             "v_fmac_f32_dpp %[f2], %[m], %[f3] row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t" \
             "v_fmac_f32 %[f3], %[m], %[f0]\n\t"
#define FP_B "v_fmac_f32_dpp %[f0], %[m], %[f3] row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t" \
             "v_pk_fma_f32 %[p1], %[p1], %[p0], %[p1]\n\t" \
             "v_fmac_f32_dpp %[f1], %[m], %[f0] row_newbcast:11 row_mask:0xf bank_mask:0xf\n\t" \
             "v_med3_f32 %[f2], %[f2], %[m], %[f1]\n\t" \
             "v_pk_mul_f32 %[p0], %[p0], %[p1]\n\t" \
             "v_add_f32 %[f3], %[f3], %[m]\n\t" \
             "v_fmac_f32 %[f0], %[m], %[f2]\n\t"
// integer / move VALU, 2 x 10 per unit
#define IN_A "v_add_u32 %[i0], %[i0], %[i1]\n\t" "v_mov_b32 %[i2], %[i3]\n\t" "v_add_u32 %[i1], %[i1], %[i2]\n\t" \
             "v_add3_u32 %[i3], %[i3], %[i0], %[i1]\n\t" "v_add_u32 %[i0], 4, %[i0]\n\t" "v_mov_b32 %[i1], %[i0]\n\t" \
             "v_lshl_add_u32 %[i2], %[i2], 2, %[i3]\n\t" "v_xor_b32 %[i3], %[i3], %[i0]\n\t" "v_add_u32 %[i2], %[i2], %[i1]\n\t" \
             "v_readlane_b32 %[s3], %[i0], 3\n\t"
#define IN_B "v_add_u32 %[i1], %[i1], %[i0]\n\t" "v_mov_b32 %[i3], %[i2]\n\t" "v_add_u32 %[i0], %[i0], %[i3]\n\t" \
             "v_add3_u32 %[i2], %[i2], %[i1], %[i0]\n\t" "v_mov_b32 %[i0], %[i1]\n\t" "v_add_u32 %[i3], 8, %[i3]\n\t" \
             "v_lshl_add_u32 %[i1], %[i1], 1, %[i2]\n\t" "v_mov_b32 %[i2], %[i0]\n\t" "v_add_u32 %[i1], %[i1], %[i3]\n\t" \
             "v_mul_lo_u32 %[i3], %[i3], %[i0]\n\t"
// scalar unit, 4 x 7 per unit
#define SA "s_add_u32 %[s0], %[s0], %[s1]\n\t" "s_and_b32 %[s1], %[s1], 0xffff\n\t" "s_lshl_b32 %[s2], %[s0], 1\n\t" \
           "s_add_u32 %[s1], %[s1], %[s2]\n\t" "s_cmp_lt_i32 %[s0], %[s1]\n\t" "s_cselect_b32 %[s2], %[s0], %[s1]\n\t" "s_nop 0\n\t"
// LDS, 11 per unit (reads 4 b128 + 3 b32 + 1 read2, writes 2 b32 + 1 b128) and 6 global loads; one counter wait per unit
#define LD_A "ds_read_b128 %[l0], %[la]\n\t" "ds_read_b32 %[l4], %[la] offset:64\n\t" "ds_read_b128 %[l1], %[la] offset:2048\n\t" \
             "ds_write_b32 %[lw], %[f0] offset:128\n\t" "ds_read2_b32 %[l5], %[la] offset0:8 offset1:24\n\t" "ds_read_b32 %[l6], %[la] offset:192\n\t"
#define LD_B "ds_read_b128 %[l2], %[la] offset:4096\n\t" "ds_write_b128 %[lw], %[l0] offset:256\n\t" "ds_read_b128 %[l3], %[la] offset:6144\n\t" \
             "ds_read_b32 %[l7], %[la] offset:320\n\t" "ds_write_b32 %[lw], %[f1] offset:512\n\t"
#define VM_A "global_load_dword %[g0], %[ga], off\n\t" "global_load_dwordx4 %[g4], %[ga], off offset:256\n\t" "global_load_dword %[g1], %[ga], off offset:1024\n\t"
#define VM_B "global_load_dword %[g2], %[ga], off offset:2048\n\t" "global_load_dwordx4 %[g5], %[ga], off offset:512\n\t" "global_load_dword %[g3], %[ga], off offset:3072\n\t"
#define WAITALL "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
#define NONE ""

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const float* tab, float* out, int passes, int units_per_pass, int barriers) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, l = tid & 63;
    for (int i = tid; i < 4096; i += 512) lds[i] = (float)i;
    __syncthreads();
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, l0, l1, l2, l3, g4, g5;
    float a = l * 1e-3f, b = 1.f + l * 1e-6f, m = 1.0001f, f0 = l, f1 = l + 1, f2 = l + 2, f3 = l + 3, l4, l6, l7, g0, g1, g2, g3;
    f32x2 p0 = {1.f, 2.f}, p1 = {0.5f, 0.25f}, l5;
    unsigned i0 = tid, i1 = 3, i2 = 5, i3 = 7, s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lds + (l & 15) * 4 + (l >> 4) * 20);
    const unsigned lw = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(lds + 8192 + tid * 4);
    const float* ga = tab + (blockIdx.x & 7) * 1024 + l;
    for (int p = 0; p < passes; ++p) {
        for (int u = 0; u < units_per_pass; ++u) {
            // per unit of 16 MFMAs: 14 FP VALU (FP_A + FP_B), 20 int VALU (IN_A + IN_B), 28 SALU (4 x SA), 11 LDS, 6 loads
            if constexpr (MODE == 0)      asm volatile(LD_A VM_A MF(0) FP_A SA MF(1) MF(2) IN_A MF(3) SA MF(0) MF(1) LD_B MF(2) VM_B MF(3) FP_B MF(0) SA MF(1) IN_B MF(2) MF(3) SA MF(0) MF(1) MF(2) MF(3) WAITALL
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [p0] "+v"(p0), [p1] "+v"(p1), [i0] "+v"(i0), [i1] "+v"(i1), [i2] "+v"(i2), [i3] "+v"(i3), [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3), [l0] "=&v"(l0), [l1] "=&v"(l1), [l2] "=&v"(l2), [l3] "=&v"(l3), [l4] "=&v"(l4), [l5] "=&v"(l5), [l6] "=&v"(l6), [l7] "=&v"(l7), [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3), [g4] "=&v"(g4), [g5] "=&v"(g5)
                 : [a] "v"(a), [b] "v"(b), [m] "v"(m), [la] "v"(la), [lw] "v"(lw), [ga] "v"(ga) : "memory", "scc");
            else if constexpr (MODE == 1) asm volatile(LD_A VM_A MF(0) FP_A SA MF(1) MF(2) IN_A MF(3) SA MF(0) MF(1) LD_B MF(2) VM_B MF(3) FP_B MF(0) SA MF(1) MF(2) MF(3) SA MF(0) MF(1) MF(2) MF(3) WAITALL      // integer VALU halved
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [p0] "+v"(p0), [p1] "+v"(p1), [i0] "+v"(i0), [i1] "+v"(i1), [i2] "+v"(i2), [i3] "+v"(i3), [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3), [l0] "=&v"(l0), [l1] "=&v"(l1), [l2] "=&v"(l2), [l3] "=&v"(l3), [l4] "=&v"(l4), [l5] "=&v"(l5), [l6] "=&v"(l6), [l7] "=&v"(l7), [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3), [g4] "=&v"(g4), [g5] "=&v"(g5)
                 : [a] "v"(a), [b] "v"(b), [m] "v"(m), [la] "v"(la), [lw] "v"(lw), [ga] "v"(ga) : "memory", "scc");
            else if constexpr (MODE == 2) asm volatile(LD_A VM_A MF(0) FP_A SA MF(1) MF(2) MF(3) SA MF(0) MF(1) LD_B MF(2) VM_B MF(3) FP_B MF(0) SA MF(1) MF(2) MF(3) SA MF(0) MF(1) MF(2) MF(3) WAITALL                  // no integer VALU
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [p0] "+v"(p0), [p1] "+v"(p1), [i0] "+v"(i0), [i1] "+v"(i1), [i2] "+v"(i2), [i3] "+v"(i3), [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3), [l0] "=&v"(l0), [l1] "=&v"(l1), [l2] "=&v"(l2), [l3] "=&v"(l3), [l4] "=&v"(l4), [l5] "=&v"(l5), [l6] "=&v"(l6), [l7] "=&v"(l7), [g0] "=&v"(g0), [g1] "=&v"(g1), [g2] "=&v"(g2), [g3] "=&v"(g3), [g4] "=&v"(g4), [g5] "=&v"(g5)
                 : [a] "v"(a), [b] "v"(b), [m] "v"(m), [la] "v"(la), [lw] "v"(lw), [ga] "v"(ga) : "memory", "scc");
            else if constexpr (MODE == 3) asm volatile(MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3)                                                                      // MFMAs only
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3) : [a] "v"(a), [b] "v"(b));
            else if constexpr (MODE == 4) asm volatile(MF(0) FP_A SA MF(1) MF(2) IN_A MF(3) SA MF(0) MF(1) MF(2) MF(3) FP_B MF(0) SA MF(1) IN_B MF(2) MF(3) SA MF(0) MF(1) MF(2) MF(3)                                   // no LDS / global memory
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [p0] "+v"(p0), [p1] "+v"(p1), [i0] "+v"(i0), [i1] "+v"(i1), [i2] "+v"(i2), [i3] "+v"(i3), [s0] "+s"(s0), [s1] "+s"(s1), [s2] "+s"(s2), [s3] "+s"(s3)
                 : [a] "v"(a), [b] "v"(b), [m] "v"(m) : "scc");
            else if constexpr (MODE == 5) asm volatile(MF(0) FP_A MF(1) MF(2) IN_A MF(3) MF(0) MF(1) MF(2) MF(3) FP_B MF(0) MF(1) IN_B MF(2) MF(3) MF(0) MF(1) MF(2) MF(3)                                               // MFMA + VALU only (no SALU, no memory)
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [p0] "+v"(p0), [p1] "+v"(p1), [i0] "+v"(i0), [i1] "+v"(i1), [i2] "+v"(i2), [i3] "+v"(i3), [s3] "+s"(s3)
                 : [a] "v"(a), [b] "v"(b), [m] "v"(m));
            else if constexpr (MODE == 6) asm volatile(MF(0) FP_A MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3) FP_B MF(0) MF(1) MF(2) MF(3) MF(0) MF(1) MF(2) MF(3)                                                         // MFMA + FP VALU only
                 : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), [f0] "+v"(f0), [f1] "+v"(f1), [f2] "+v"(f2), [f3] "+v"(f3), [p0] "+v"(p0), [p1] "+v"(p1)
                 : [a] "v"(a), [b] "v"(b), [m] "v"(m));
            if (barriers > 0 && (u + 1) * barriers / units_per_pass != u * barriers / units_per_pass) __syncthreads();      // `barriers` per pass, evenly spaced
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float r = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + p0[0] + p1[1] + (float)(i0 + i1 + i2 + i3 + s0 + s1 + s2 + s3);
    if (r == 123.456f) out[tid] = r;
}

int main() {
    const int passes = 45 * 4, units = 36;          // 36 units x 16 = 576 MFMAs per wave-pass (the kernel: 575)
    const double GHZ = 2.4;
    float *tab, *out;
    hipMalloc(&tab, 1 << 20); hipMemset(tab, 0, 1 << 20); hipMalloc(&out, 4096);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const int cus = pr.multiProcessorCount;
    const size_t lds = 72 * 1024;                     // two workgroups per CU, like score_kernel<3,2,4> (75.8 KB each)
    const char* names[] = {"the kernel's mix (575 MFMA : 1214 VALU : 1018 SALU : 390 LDS : 212 VMEM per wave-pass)", "integer / move VALU halved", "no integer / move VALU",
                           "MFMAs only", "no LDS / global memory instructions", "MFMA + VALU only (no SALU, no memory)", "MFMA + floating-point VALU only"};
    void (*ks[])(const float*, float*, int, int, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
    printf("# %s, %d CUs; %d workgroups of 8 waves (2 per CU), %d passes x %d units x 16 MFMAs per wave; clock taken as %.1f GHz\n", pr.name, cus, 2 * cus, passes, units, GHZ);
    printf("# pipe_busy = MFMAs per SIMD x 32 cycles / (elapsed x clock); 4 waves per SIMD\n");
    for (int bar = 0; bar <= 24; bar += 24) {
        for (int mode = 0; mode < 7; ++mode) {
            if (bar && mode != 0 && mode != 3) continue;
            hipFuncSetAttribute((const void*)ks[mode], hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(ks[mode], dim3(2 * cus), dim3(512), lds, 0, tab, out, passes, units, bar);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const double mfma_per_simd = 4.0 * passes * units * 16;      // 4 waves per SIMD (2 workgroups x 8 waves over 4 SIMDs)
            const double busy = mfma_per_simd * 32.0 / (best * 1e-3 * GHZ * 1e9);
            printf("%-100s %s  %8.3f ms   pipe_busy %.4f   cycles per wave-pass %.0f\n", names[mode], bar ? "24 barriers/pass" : "no barriers     ", best, busy,
                   best * 1e-3 * GHZ * 1e9 / passes);
        }
    }
    if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
    return 0;
}
