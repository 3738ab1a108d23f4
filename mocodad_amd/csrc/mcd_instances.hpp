// mcd_instances.hpp — every kernel instantiation of libmocodad_hip.so and the translation unit that holds it.
//
// The library is built from mcd_api.hip (C ABI, packers, dispatch, the runtime-shape kernels) plus mcd_inst.hip compiled once
// per unit with -DMCD_INST_UNIT=<n>: a unit explicitly instantiates the launcher templates of its rows (and with them the
// kernels); every other translation unit sees them as `extern template` and compiles none of that device code.  The units
// build in parallel (mocodad_amd/build.py); their number and the assignment below only balance compile times.
//   X(unit, T_u, NB, MINW, LT)   score_kernel<T_u, NB, MINW, LT>      (LT: the layer-test form behind mcd_layer_forward)
//   X(unit, T_c, NB)             cond_fast_kernel / cond_unet_kernel<T_c, NB>
//   X(unit, TP, NB, LT)          score_tiled_kernel<TP, NB, LT>
//   X(unit, TP, NB)              score_tiled_kernel<TP, NB, false, true>: the 'E_unet' condition encoder at 13 .. 32 condition frames
#pragma once

#define MCD_INST_UNITS 25

// Per-unit compile flags (mocodad_amd/build.py reads these lines).  The wave count of a workgroup is a translation-unit constant
// (MCD_NWAVES): units 3, 5, 23, 11 and 12 hold ONLY the 12-frame, the 9- / 10- / 11-frame and the 24- / 32-frame (slab-tiled) trajectory kernels and build them with twelve waves per workgroup -- three per SIMD,
// 168 registers, 12 mix units per stage, n-thirds in the 64-channel GEMMs: +2.6 % over eight waves (profiles/r04ak_t12_w12_ab.txt).
// Its launcher (the same translation unit) launches 768 threads; nothing outside the unit depends on the wave count.
#define MCD_UNIT_FLAGS_3 "-DMCD_NWAVES=12"      // (round 4 added -mllvm -amdgpu-sched-strategy=iterative-minreg: 25 -> 19 spilled registers, +0.6 %; the kernel has spilled nothing since round 5, and the default strategy is +0.3 .. 0.4 % now: profiles/r06j_switch_sweep_ab.txt)
#define MCD_UNIT_FLAGS_23 "-DMCD_NWAVES=12 -mllvm -amdgpu-sched-strategy=iterative-minreg"     // 9 frames: +1.3 % with it (11 frames -2.4 %, 24 / 32 frames -0.7 / -3.2 %: default strategy, profiles/r04at_minreg_ab.txt)
#define MCD_UNIT_FLAGS_11 "-DMCD_NWAVES=12"     // the slab-tiled kernel at 24 frames: +4.5 % (profiles/r04aq_tiled_w12_ab.txt)
#define MCD_UNIT_FLAGS_12 "-DMCD_NWAVES=12"     // ... and at 32 frames: +2.0 %
#define MCD_UNIT_FLAGS_5 "-DMCD_NWAVES=12"      // 9, 10 and 11 frames (profiles/r04al_w12_shapes_ab.txt, r04an_t10_w12_ab.txt)
// The layer-test (LT) forms behind mcd_layer_forward are built with the flags AND the template arguments of their production
// twins, so that the stage tests run the shipped stage code (the twelve-wave mix tables, n-thirds tiling, skip-round branches):
#define MCD_UNIT_FLAGS_24 "-DMCD_NWAVES=12 -mllvm -amdgpu-sched-strategy=iterative-minreg"     // LT of 9 frames (= unit 23)
#define MCD_UNIT_FLAGS_25 "-DMCD_NWAVES=12"     // LT of 10, 11 and 12 frames (= units 5, 3)
#define MCD_UNIT_FLAGS_15 "-DMCD_NWAVES=12"     // LT of the slab-tiled kernel at 24 frames (= unit 11)
#define MCD_UNIT_FLAGS_16 "-DMCD_NWAVES=12"     // ... and at 32 frames (= unit 12)

#ifdef MCD_TUNING_VARIANTS      // alternative workgroup shapes (MCD_OPT_VARIANT): developer builds only
#define MCD_SCORE_VARIANT_INSTANCES(X) X(1, 3, 4, 2, false) X(1, 3, 1, 4, false) X(1, 3, 2, 2, false) X(2, 6, 2, 2, false)
#else
#define MCD_SCORE_VARIANT_INSTANCES(X)
#endif

#define MCD_SCORE_INSTANCES(X) \
    X(1, 3, 2, 4, false)  /* HR-Avenue / HR-STC: 2 chains per workgroup, 2 workgroups per CU (<= 128 VGPRs) */ \
    X(1, 1, 4, 4, false) X(1, 2, 2, 4, false) \
    X(2, 6, 1, 4, false)  /* concat over 6 frames */ \
    X(2, 4, 1, 4, false) \
    X(3, 12, 1, 3, false) /* seg_len 24 split in halves: 1 workgroup of TWELVE waves per CU (unit 3 is compiled with MCD_UNIT_FLAGS_3), 168 registers */ \
    X(22, 8, 1, 2, false) \
    X(4, 5, 1, 4, false) X(4, 7, 1, 2, false) \
    X(23, 9, 1, 3, false) X(5, 10, 1, 3, false) X(5, 11, 1, 3, false) /* twelve waves as well (unit 5): +3.7 / +0.9 / +0.9 %; 7 and 8 frames measured -1 % / +0.2 %: eight waves */ \
    X(9, 3, 2, 4, true) X(9, 6, 1, 4, true) X(25, 12, 1, 3, true) X(24, 9, 1, 3, true) \
    X(13, 5, 1, 4, true) X(13, 7, 1, 2, true) X(25, 10, 1, 3, true) X(25, 11, 1, 3, true) \
    MCD_SCORE_VARIANT_INSTANCES(X)

#define MCD_COND_FAST_INSTANCES(X) \
    X(7, 1, 4) X(7, 2, 3) X(7, 3, 2) X(7, 4, 2) X(7, 5, 2) X(7, 6, 2) X(7, 7, 1) X(7, 8, 1) X(7, 9, 1) X(7, 10, 1) X(7, 11, 1) X(7, 12, 1) \
    X(20, 13, 1) X(20, 14, 1) X(20, 15, 1) X(20, 16, 1) X(21, 17, 1) X(21, 18, 1) X(21, 19, 1) X(21, 20, 1)   /* 13 .. 20 frames: one window per workgroup, up to 158 KB of LDS */

#define MCD_COND_UNET_INSTANCES(X) \
    X(8, 1, 4) X(8, 2, 2) X(8, 3, 2) X(8, 4, 2) X(8, 5, 2) X(8, 6, 1) X(8, 7, 1) X(10, 8, 1) X(10, 9, 1) X(10, 10, 1) X(10, 11, 1) X(10, 12, 1)

#define MCD_TILED_INSTANCES(X) X(6, 16, 1, false) X(11, 24, 1, false) X(12, 32, 1, false) X(14, 16, 1, true) X(15, 24, 1, true) X(16, 32, 1, true)

#define MCD_TILED_COND_INSTANCES(X) X(17, 16, 1) X(18, 24, 1) X(19, 32, 1)
