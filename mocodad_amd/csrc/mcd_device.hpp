// mcd_device.hpp — shared device code of libmocodad_hip.so: MI355X (gfx950 / CDNA4) kernels + C ABI for the MoCoDAD anomaly-scoring path.
//
// What runs here (reference: /root/reference, Python/PyTorch):
//   MoCoDAD.forward hot loop            models/mocodad.py:155-180      -> score_kernel<T_u, ...> for 1 .. 12 U-Net frames (persistent,
//   STSAE_Unet.forward                  models/stsae/stsae_unet.py:406-438   one launch for all S*(ns-1) passes), score_tiled_kernel
//                                                                       for 13 .. 32 (activations in an L2 slab, stages through LDS);
//                                                                       score_generic_kernel (plain FMAs, any count) cross-checks both
//   ST_GCNN_layer / ConvTemporalGraphical / CNN_layer  models/gcae/stsgcn.py:94-199
//   DDPM ancestral update + SmoothL1    models/mocodad.py:172-178,484
//   STSE.encode (condition encoder)     models/stsae/stsae.py:59-92    -> cond_fast_kernel (1 .. 12 frames) / cond_encode_kernel;
//   STSE_Unet ('E_unet' encoder)        models/stsae/stsae_unet.py:62-146     cond_unet_kernel (1 .. 12) / cond_unet_generic_kernel
//   _aggregation_strategy               models/mocodad.py:454-520      -> aggregate_kernel
//
// Design: one 512-thread workgroup owns NB reverse-diffusion chains (a chain = one
// (window, sample) pair) for their whole trajectory.  Activations live in LDS as [column][channel]
// (column = (chain, frame, joint), channel fastest, row stride = C+4 floats = 4*odd: conflict-free
// MFMA-operand reads and b128 epilogue stores).  Every dense contraction runs on v_mfma_f32_16x16x4_f32
// (exact fp32 = fmaf chain), with weights pre-packed in fragment order and streamed from L2 into registers:
//   mix       joint mix A_q^T x Y_q per (chain, 16-channel block); Y_q (the time mix) is built in registers with
//             DPP-broadcast coefficients as the B operand
//   GEMM      the 1x1 channel convolutions (tcn + residual, BatchNorm folded) as one K-concatenated
//             [W_t | W_r] x [Z ; X] product; the epilogue (+bias, PReLU, +SiLU-Linear embedding, b128 store)
//             runs right behind each 16x16 tile
//   resample  joint down/up-sampling per (frame, 16-channel block); the down-samplers' B operands double as the
//             register-resident U-Net skip tensors d1/d2 that the up-samplers add back
//   W-first   layers 6, 8 and 10: GEMM first, then the mix on the (fewer) output channels with the layer epilogue
//             (layer 10: + U-Net residual + DDPM update) in the mix's store functor
// Everything is fp32 (the reverse chain amplifies error by up to 1e3).

#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#include <string>
#include <vector>
#include <unordered_map>
#include <utility>
#include <type_traits>
#include <atomic>

#include "../../include/mocodad_hip.h"

namespace mcd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// read-only model coefficients addressed wave-uniformly go through the constant address space so that the
// compiler emits scalar loads (s_load_dwordx16) and feeds them to v_fmac as SGPR operands
typedef const float __attribute__((address_space(4))) cfloat;
// Explicit global address space for the packed-weight pointers: after the per-step laundering of the base pointer
// the compiler can no longer infer it and would emit FLAT loads, which tick vmcnt AND lgkmcnt and force every
// LDS wait to also drain the outstanding weight loads.
typedef const float __attribute__((address_space(1))) gfloat;
typedef const f32x4 __attribute__((address_space(1))) gf32x4;
__device__ __forceinline__ gfloat* as_global(const float* p) { return (gfloat*)p; }
// LDS accesses through explicit 32-bit LDS addresses: a per-lane base computed once (and made opaque where the compiler would
// rather re-derive it per use), constant byte offsets folded into the instruction's offset field
typedef float __attribute__((address_space(3))) lds_float;
typedef f32x4 __attribute__((address_space(3))) lds_f32x4;
// (the low half of a generic pointer into LDS IS its LDS address; an addrspacecast would add a null check per use -- and with
// -amdgpu-sched-strategy=iterative-minreg one of those trips "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(uintptr_t)p; }
__device__ __forceinline__ float4 lds_load4(unsigned addr) {
    const f32x4 v = *(lds_f32x4*)(uintptr_t)addr;
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_store4(unsigned addr, float a, float b, float c, float d) {
    *(lds_f32x4*)(uintptr_t)addr = f32x4{a, b, c, d};
}
typedef f32x4 __attribute__((address_space(1))) gf32x4_rw;
__device__ __forceinline__ void store_global4(float* p, float4 v) {    // 16-byte aligned global store (global_store, not flat_store:
    *(gf32x4_rw*)p = f32x4{v.x, v.y, v.z, v.w};                        // a flat access ticks lgkmcnt as well as vmcnt)
}
__device__ __forceinline__ float4 load_global4(const float* p) {       // 16-byte aligned global load
    const f32x4 v = *(gf32x4*)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}

#ifndef MCD_NWAVES
#define MCD_NWAVES 8
#endif
// Layer 8 (su4.1, 64 -> 32 channels at 12 joints) W-first in score_kernel, like layers 6 and 10: mix(W X) = W mix(X), so the
// GEMM [W_t ; W_r] X (64 rows x K = 64: as many MFMAs as [W_t | W_r] [Z ; X], 32 rows x K = 128) comes first and the mix runs on
// the 32 OUTPUT channels instead of the 64 input ones -- half the time-mix FMAs, joint-mix MFMAs and unit overhead of that
// stage (round 6; the packer of mcd_api.hip packs the layer's weights to match).  0: mix-first as in rounds 1-5 (A/B builds).
#ifndef MCD_L8_WFIRST
#define MCD_L8_WFIRST 1
#endif
#ifndef MCD_XB32
#define MCD_XB32 1      // 1: the mixes' X reads of the kernels without a register cap as single ds_read_b32 (mix_stage); 2: + their Z stores
#endif
#ifndef MCD_XB32_CAPPED
#define MCD_XB32_CAPPED 1      // (round 6: +0.7 % at 3 frames, profiles/r06g_t3_variants_ab.txt) 1: the single-read form of the mixes' X reads in the register-capped trajectory kernels too; 2: + the Z stores (equal)
#endif
constexpr int NWAVES = MCD_NWAVES;          // waves per workgroup (8; 16 is a tuning experiment)
constexpr int NTHREADS = NWAVES * 64;
constexpr int C0 = 2;        // num_coords
constexpr int EDIM = 16;     // embedding_dim
constexpr int NLAYERS = 11;  // ST-GCN layers of the U-Net
constexpr int EMB_TOTAL = 532;  // sum of C_out over the 11 layers, last padded to 4
constexpr int EMB_STRIDE = 536;

__host__ __device__ constexpr int cs_of(int c) { return c + 4; }      // LDS row stride for c channels (4*odd)
__host__ __device__ constexpr int ceil16(int x) { return (x + 15) / 16 * 16; }

// channel plan of STSAE_Unet (stsae_unet.py:255-357 defaults; mocodad.py:121-124 never overrides them)
struct LDesc { int cin, cout, V, res; };
__host__ __device__ constexpr LDesc layer_desc(int l) {
    return l == 0 ? LDesc{16, 16, 17, 1} /* 2 real input channels, K zero-padded */ : l == 1 ? LDesc{16, 32, 17, 1} : l == 2 ? LDesc{32, 32, 17, 0}
         : l == 3 ? LDesc{32, 64, 12, 1} : l == 4 ? LDesc{64, 64, 12, 0} : l == 5 ? LDesc{64, 128, 10, 1}
         : l == 6 ? LDesc{128, 64, 10, 1} : l == 7 ? LDesc{64, 64, 12, 0} : l == 8 ? LDesc{64, 32, 12, 1}
         : l == 9 ? LDesc{32, 32, 17, 0} : LDesc{32, 2, 17, 1};
}
__host__ __device__ constexpr int emb_off(int l) {
    return l == 0 ? 0 : l == 1 ? 16 : l == 2 ? 48 : l == 3 ? 80 : l == 4 ? 144 : l == 5 ? 208 : l == 6 ? 336
         : l == 7 ? 400 : l == 8 ? 464 : l == 9 ? 496 : 528;
}

// Offset table stored in the first TAB_FLOATS words of the packed weight buffer (offsets in floats from the
// buffer start).  The kernel scalar-loads an entry right where it is used; keeping the table out of the
// kernarg segment stops the compiler from hoisting ~100 pointers into SGPRs for the whole trajectory loop.
constexpr int TAB_FLOATS = 256;   // [0,128): U-Net table, [128,256): fast condition-encoder table
enum { F_TQ = 0, F_AM = 1, F_WP = 2, F_BIAS = 3, F_SLOPE = 4, F_STRIDE = 8 };
//   tab[l*8 + F_TQ]    time-mix coefficients packed 16 per VGPR for DPP row broadcast, TQD[q][r][64]:
//                      lane 16g+i = gcn.T[v = mix_vmap(s,g)][t][q] with s*T+t = 16r+i
//   tab[l*8 + F_AM]    MFMA A-operand fragments of A_q^T, AF[q][mt][s][64]: lane (i, g) = gcn.A[q][v=mix_vmap(s,g)][w=16mt+i]
//   tab[l*8 + F_WP]    MFMA-packed [W_t' | W_r'] (layers 6, 8, 10: [W_t' ; W_r'] stacked, W-first)
//   tab[l*8 + F_BIAS]  folded bias, padded to 16
//   tab[l*8 + F_SLOPE] PReLU slope (float bits)
constexpr int TAB_WE = 88, TAB_BE = 89;   // WeAll[EMB_TOTAL][16], beAll[EMB_TOTAL]
constexpr int TAB_RSW = 90, TAB_RSB = 94; // down1, down2, up3, up2: MFMA A fragments WF[mt][ks][64] (lane (i,g) =
                                          // Wd'[16mt+i][rs_vmap(ks,g)]) and bd' padded to 32
typedef const int __attribute__((address_space(4))) cint;
__device__ __forceinline__ int tab_i(const float* base, int idx) { return ((cint*)base)[idx]; }
__device__ __forceinline__ float tab_f(const float* base, int idx) { return ((cfloat*)base)[idx]; }
struct LayerW { int tq, am, wp, bias; float slope; };
__device__ __forceinline__ LayerW layer_w(const float* base, int l) {
    LayerW w;
    w.tq = tab_i(base, l * F_STRIDE + F_TQ); w.am = tab_i(base, l * F_STRIDE + F_AM);
    w.wp = tab_i(base, l * F_STRIDE + F_WP); w.bias = tab_i(base, l * F_STRIDE + F_BIAS);
    w.slope = tab_f(base, l * F_STRIDE + F_SLOPE);
    return w;
}

// Optional in-kernel stage timing (build with -DMCD_PROFILE; tools/stage_profile.py): thread 0 of block 0 adds the
// s_memtime delta of each stage to an LDS accumulator (fire-and-forget ds_add: the timing wave never waits on global
// memory for the instrumentation); the accumulators are written to P.prof[] when the kernel ends.
#ifdef MCD_PROFILE
constexpr int PROF_STAGE = 72;                           // stage-time slots
constexpr int PROF_BAR = 30;                             // barriers of one pass that get a slot
constexpr int PROF_NW = NWAVES;                         // (small on purpose: the 3-frame plan has 1.3 KB to spare below 2 workgroups per CU)
constexpr int PROF_SLOTS = PROF_STAGE + PROF_NW + PROF_BAR * PROF_NW;   // stage times | (unused) | wait[barrier][wave]
constexpr int PROF_TRACE = 128;                          // time-stamp slots of the traced pass (after the PROF_SLOTS accumulators)
#endif
// Everything the instrumentation needs lives in registers of the profiled workgroup (block 0): the timing adds are
// fire-and-forget LDS atomics, no global memory access, no LDS round trip on the waves' paths.
// Wave priority by PHASE (MCD_PHPRIO, a tuning experiment): the stages of a pass alternate between latency-bound ones (mixes,
// resamplers, the tail: short dependent chains, a few instructions per wave) and throughput-bound ones (the channel GEMMs).  Two
// workgroups share a CU; the issue arbiter picks by priority, then age.  `lat()` / `thr()` are called at the top of the
// stages: a wave in a latency-bound stage outranks the other workgroup's GEMM waves (it needs few issue slots, but needs them
// promptly), a GEMM wave takes what is left.  `slice` = the time-slice bit of the alternating scheme (see score_kernel).
#ifndef MCD_PHPRIO
#define MCD_PHPRIO 0
#endif
struct PhasePrio {
    int slice;      // 0 / 1, wave-uniform
    __device__ __forceinline__ void lat() const {
        if constexpr (MCD_PHPRIO == 1 || MCD_PHPRIO == 3) __builtin_amdgcn_s_setprio(3);
        else if constexpr (MCD_PHPRIO == 2) { if (slice) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
        else if constexpr (MCD_PHPRIO == 4) __builtin_amdgcn_s_setprio(0);
    }
    __device__ __forceinline__ void thr() const {
        if constexpr (MCD_PHPRIO == 1 || MCD_PHPRIO == 2) { if (slice) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        else if constexpr (MCD_PHPRIO == 3) __builtin_amdgcn_s_setprio(0);
        else if constexpr (MCD_PHPRIO == 4) __builtin_amdgcn_s_setprio(3);
    }
};
struct Prof {
    PhasePrio pp;
#ifdef MCD_PROFILE
    unsigned* acc;               // LDS, PROF_SLOTS words
    unsigned long long tlast;
    bool on;                     // thread 0 of block 0: stage times
    bool won;                    // lane 0 of every wave of block 0: barrier waits
    int bidx;                    // barrier index inside the pass (wave-uniform)
    int wv;
    unsigned* tr;                // LDS: per-wave time stamps (low 32 bits) of the traced pass, tr[slot * NWAVES + wave]; null: none
    bool tr_on;                  // lane 0 of every wave of block 0, during the traced pass only
    __device__ __forceinline__ void trace(int slot) {
        if (tr_on) tr[slot * PROF_NW + wv] = (unsigned)__builtin_readcyclecounter();
    }
    __device__ __forceinline__ void mark(int id) {
        if (on) {
            const unsigned long long t = __builtin_readcyclecounter();
            __hip_atomic_fetch_add(acc + id, (unsigned)(t - tlast), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            tlast = t;
        }
    }
    // workgroup barrier + the cycles this wave waited there (which waves a stage waits for)
    __device__ __forceinline__ void sync() {
        const unsigned long long t0 = __builtin_readcyclecounter();
        __syncthreads();
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (won && bidx < PROF_BAR)
            __hip_atomic_fetch_add(acc + PROF_STAGE + PROF_NW + bidx * PROF_NW + wv, (unsigned)(t1 - t0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        ++bidx;
    }
    __device__ __forceinline__ void off() { pp.slice = 0; on = false; won = false; acc = nullptr; tlast = 0; bidx = 0; wv = 0; tr = nullptr; tr_on = false; }
#else
    __device__ __forceinline__ void trace(int) {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ void off() { pp.slice = 0; }
#endif
};
// workgroup barrier: every scope that synchronises has a `Prof prof` in reach
#define bsync() prof.sync()
#define STAGE(id) prof.mark(id)

// where the windows live: a dense (B,C,T,V) tensor, or a view into trajectory buffers with an optional affine
// test-time transform applied on load (mcd_window_view_t)
struct DataView {
    const float* data;
    const long long* base;
    long long sc, st;
    const int* trans;
    const float* aff;
};
__device__ __forceinline__ float load_coord(const DataView& dv, int b, int c, int t, int v, int seg_len) {
    const long long sc = dv.base ? dv.sc : (long long)seg_len * 17;
    const long long st = dv.base ? dv.st : 17;
    const float* p = dv.data + (dv.base ? dv.base[b] : (long long)b * C0 * seg_len * 17) + t * st + v;
    if (!dv.trans) return p[c * sc];
    const float x = p[0], y = p[sc];
    const float* a = dv.aff + dv.trans[b] * 6 + c * 3;
    return (a[0] * x + a[1] * y) + a[2];
}

struct ScoreParams {
    unsigned long long* prof; // stage timing accumulators (MCD_PROFILE builds) or null
    const float* wbuf;        // packed weights; first TAB_FLOATS words = offset table
    DataView dv;              // windows: (B,C,T,V) tensor or a trajectory view
    const float* noise;       // (S,K,B,C,Tx,V) or null
    const float* cond_emb;    // (B,16) or null
    const float* step_table;  // (ns, 4+16)
    const float* x_in;        // single-pass mode: (B,C,Tu,V)
    float* loss_out;          // (B,S)
    float* pose_out;          // (B,S,C,Tx,V) or null
    float* eps_out;           // single-pass mode
    unsigned long long seed;
    long long first_window;
    int B, S, ns, seg_len, n_corrupt, loss_fn, mode, step_single, n_chains;
    // A workgroup owns NB WINDOWS (group g = blockIdx / split) and runs the samples s = part, part + split, ... of both
    // (part = blockIdx % split) one trajectory after the other.  split = 1: it sees every sample of its windows, so the
    // condition encoder runs once per workgroup in its own LDS (cond_inkernel) and the aggregation over the samples
    // (loss_agg, aggr, aggr_q) happens here too: ONE launch per scoring call.
    int split, aggr, cond_inkernel;
    int force_split;          // host only (MCD_OPT_SPLIT): 0 = choose
    int loss_out_optional;    // host only: loss_out is the library's own scratch, not wanted when the aggregation is fused
    int plan_only;            // host only: choose `split`, do not launch
    int phase;                // tuning experiment (MCD_OPT_PHASE): the second half of the grid starts `phase` x 1024 cycles late
    int prio_shift;           // host: log2 of the priority time slice in 100 MHz ticks (see the top of the step loop); 0 = off
                              // (an ESTIMATE from the shapes; the kernel replaces it by a sixth of the measured duration of the
                              // previous launch of the same grid when `tune` holds one)
    int prio_rounds;          // host: rounds of workgroups of this launch (grid / resident slots, rounded up)
    int* tune;                // per-handle device words: [0] signature (grid, S, ns) of the launch that wrote [1] = lifetime of its
                              // workgroup 0 in 100 MHz ticks; or null
    float aggr_q;
    float* loss_agg;          // (B,) aggregated loss, or null
    int cond_idx[12];         // cond_inkernel: data frames the condition encoder reads
    int upd_shift;            // some prediction updates a frame other than the one it is read at (element-wise tail: barrier between reads and writes)
    unsigned fixed_mask;      // bit t: U-Net frame t is a condition frame copied from the window (concat / imputation)
    int src_frame[12];        // data frame feeding U-Net frame t (condition frame, or ground truth of a denoised one)
    int tx_of[12];            // denoised U-Net frame t -> its index among the corrupt frames
    int pos_of[12];           // corrupt frame k -> its U-Net frame
    int upd_of[12];           // U-Net frame t -> corrupt frame whose eps-prediction is read at t (-1: none)
    const int* win_mask;      // random_imp: (B,) per-window bitmask of the condition frames (frames in natural order;
                              // replaces fixed_mask and the four maps above, which are then derived from the mask)
    // layer-test instantiation only (mcd_layer_forward): stage id (0..10 ST-GCN layer, 11 down1, 12 down2, 13 up3, 14 up2),
    // its input (B,Cin,T,Vin) and output (B,Cout,T,Vout)
    int lt_stage;
    const float* lt_in;
    float* lt_out;
    const float* lt_skip;     // slab-tiled kernel, stages 7 / 9: the U-Net skip tensor added behind the fused resampler (or null)
};
// frame layout of one window: `fixed` = its condition-frame bitmask (P.fixed_mask, or the window's own for random_imp)
__device__ __forceinline__ int fm_tx(const ScoreParams& P, int fixed, int t) {
    return P.win_mask ? __popc(~fixed & ((1 << t) - 1)) : P.tx_of[t];
}
__device__ __forceinline__ int fm_src(const ScoreParams& P, int t) { return P.win_mask ? t : P.src_frame[t]; }

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller (perf mode noise; parity mode reads the caller's noise tensor)
// ------------------------------------------------------------------------------------------------
// all four output words: two Box-Muller pairs = four normals per call
__device__ __forceinline__ void philox_normal4(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2, unsigned c3,
                                               float (&z)[4]) {
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned w[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float u1 = ((float)(w[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(w[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float r = __fsqrt_rn(-1.38629436111989f * __log2f(u1));       // hardware log2 / sqrt / sin / cos, as below
        z[2 * h] = r * __cosf(6.28318530717958647692f * u2);
        z[2 * h + 1] = r * __sinf(6.28318530717958647692f * u2);
    }
}
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned c0, unsigned c1, unsigned c2, unsigned c3) {
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const float u1 = ((float)(c0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(c1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
    // Box-Muller on the hardware transcendental units (v_log_f32, v_cos_f32, v_sqrt_f32): this is a noise source,
    // ~1e-6 relative accuracy is irrelevant to its distribution
    return __fsqrt_rn(-1.38629436111989f * __log2f(u1)) * __cosf(6.28318530717958647692f * u2);
}

// PReLU(x) = x >= 0 ? x : a x  ==  max(x, a x) for a <= 1, min(x, a x) for a > 1: one multiply and one v_med3_f32 against
// +-inf picked by the (wave-uniform) slope -- the compare + select form costs a third VALU instruction per element, and
// the fp32 MFMAs share the SIMD's issue time with the VALU (tools/ubench/coissue.hip)
// +inf for a slope <= 1, -inf above: as an integer compare of the float's bits (monotonic for non-negative floats, negative
// ones are negative integers), which stays on the scalar unit for a wave-uniform slope -- gfx950 has no scalar float compare
__device__ __forceinline__ float prelu_bound(float a) {
    return __int_as_float(__float_as_int(a) <= 0x3f800000 ? 0x7f800000 : (int)0xff800000);
}
__device__ __forceinline__ float prelu(float x, float a) { return __builtin_amdgcn_fmed3f(x, a * x, prelu_bound(a)); }

// sum over the 16 lanes of a DPP row, result in every lane: quad swaps, then half-row and row mirrors (no lane id, no LDS)
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    v += dpp_f<0x140>(v);     // row_mirror
    return v;
}

// aggregation of one window's S per-sample losses (mocodad.py:504-512 best / worst with strict comparisons from 1e10 / -1,
// :489-492 mean / median, :513-516 quantile): torch's conventions -- median = lower middle, quantile = linear interpolation
// (torch.lerp).  Sorts L in place for the order statistics.
__device__ __forceinline__ float aggregate_losses(float* L, int S, int strategy, float q) {
    if (strategy == MCD_AGGR_BEST || strategy == MCD_AGGR_WORST) {
        const bool best = strategy == MCD_AGGR_BEST;
        float cur = best ? 1e10f : -1.f;
        for (int s = 0; s < S; ++s) if (best ? (L[s] < cur) : (L[s] > cur)) cur = L[s];
        return cur;
    }
    if (strategy == MCD_AGGR_MEAN) {
        float sum = 0.f;
        for (int s = 0; s < S; ++s) sum += L[s];
        return sum / (float)S;
    }
    for (int s = 0; s < S; ++s)            // torch.median / torch.quantile return NaN when a sample is NaN (a diverged chain);
        if (L[s] != L[s]) return L[s];     // a comparison sort would leave an arbitrary finite value at the rank instead
    for (int i = 1; i < S; ++i) {          // insertion sort (S <= 64)
        const float x = L[i];
        int k = i - 1;
        while (k >= 0 && L[k] > x) { L[k + 1] = L[k]; --k; }
        L[k + 1] = x;
    }
    if (strategy == MCD_AGGR_MEDIAN) return L[(S - 1) / 2];
    const float pos = fminf(fmaxf(q, 0.f), 1.f) * (float)(S - 1);      // (q is validated on the host; the clamp is a backstop)
    const int lo = (int)floorf(pos);
    const int hi = lo + 1 < S ? lo + 1 : S - 1;
    const float wgt = pos - (float)lo;
    const float a = L[lo], c = L[hi];
    return wgt < 0.5f ? a + wgt * (c - a) : c - (c - a) * (1.f - wgt);
}

// ------------------------------------------------------------------------------------------------
// DPP helpers.  A coefficient row (<= 16 values) lives in ONE VGPR, value i in lane i of every 16-lane row
// (a single coalesced 64 B vector load); each FMA picks its coefficient with the DPP row_newbcast modifier:
//     acc += bcast_L(coef) * y        ->  v_fmac_f32_dpp acc, coef, y row_newbcast:L
// This keeps the learned time/joint mixing matrices out of the scalar cache (16 KB, thrashed by the ~40 KB
// of tables a pass touches) and costs 1/16 of the loads of a broadcast-read scheme.  EXEC must be full.
// ------------------------------------------------------------------------------------------------
// PAD = true appends `s_nop 1`: hipcc does not model the instructions inside an asm statement, so when the
// result feeds an MFMA operand next (VALU write -> MFMA SrcA/B read hazard) the wait states must be ours.
template <int L, bool PAD = false>
__device__ __forceinline__ void fmac_bc(float& acc, float coef, float y) {
    if (PAD)
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "+v"(acc) : "v"(coef), "v"(y), "n"(L));
    else
        asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(coef), "v"(y), "n"(L));
}
template <int L, bool PAD = false>
__device__ __forceinline__ float mul_bc(float coef, float y) {
    float r;
    if (PAD)
        asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "=v"(r) : "v"(coef), "v"(y), "n"(L));
    else
        asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(coef), "v"(y), "n"(L));
    return r;
}
// The time mix of a unit is QC independent chains (one per output frame) of T FMAs each.  Written chain after chain the
// wave issues T DEPENDENT v_fmac_f32_dpp in a row (the compiler keeps asm statements in source order and pads each pair
// with an s_nop): ~2x the issue time of independent instructions, and with two waves per SIMD (12 frames) nothing hides it.
// tm_step issues time step t of ALL chains as ONE asm statement -- QC independent instructions back to back, in a fixed
// order -- so no chain ever waits on itself.  INIT: v_mul (the chains' first step, outputs early-clobbered);
// PAD: the statement ends with `s_nop 1` (its results feed MFMA operands next, see fmac_bc).
#define MCD_DPP(OP, I) OP " %[y" #I "], %[c" #I "], %[x] row_newbcast:%[L] row_mask:0xf bank_mask:0xf\n\t"
#define MCD_TM_IN(I) [c##I] "v"(c[I])
#define MCD_TM2(OP, CON, NOP) asm(MCD_DPP(OP, 0) MCD_DPP(OP, 1) NOP : [y0] CON(y[0]), [y1] CON(y[1]) \
                                  : MCD_TM_IN(0), MCD_TM_IN(1), [x] "v"(x), [L] "n"(L))
#define MCD_TM3(OP, CON, NOP) asm(MCD_DPP(OP, 0) MCD_DPP(OP, 1) MCD_DPP(OP, 2) NOP : [y0] CON(y[0]), [y1] CON(y[1]), [y2] CON(y[2]) \
                                  : MCD_TM_IN(0), MCD_TM_IN(1), MCD_TM_IN(2), [x] "v"(x), [L] "n"(L))
#define MCD_TM4(OP, CON, NOP) asm(MCD_DPP(OP, 0) MCD_DPP(OP, 1) MCD_DPP(OP, 2) MCD_DPP(OP, 3) NOP \
                                  : [y0] CON(y[0]), [y1] CON(y[1]), [y2] CON(y[2]), [y3] CON(y[3]) \
                                  : MCD_TM_IN(0), MCD_TM_IN(1), MCD_TM_IN(2), MCD_TM_IN(3), [x] "v"(x), [L] "n"(L))
#define MCD_TM5(OP, CON, NOP) asm(MCD_DPP(OP, 0) MCD_DPP(OP, 1) MCD_DPP(OP, 2) MCD_DPP(OP, 3) MCD_DPP(OP, 4) NOP \
                                  : [y0] CON(y[0]), [y1] CON(y[1]), [y2] CON(y[2]), [y3] CON(y[3]), [y4] CON(y[4]) \
                                  : MCD_TM_IN(0), MCD_TM_IN(1), MCD_TM_IN(2), MCD_TM_IN(3), MCD_TM_IN(4), [x] "v"(x), [L] "n"(L))
#define MCD_TM6(OP, CON, NOP) asm(MCD_DPP(OP, 0) MCD_DPP(OP, 1) MCD_DPP(OP, 2) MCD_DPP(OP, 3) MCD_DPP(OP, 4) MCD_DPP(OP, 5) NOP \
                                  : [y0] CON(y[0]), [y1] CON(y[1]), [y2] CON(y[2]), [y3] CON(y[3]), [y4] CON(y[4]), [y5] CON(y[5]) \
                                  : MCD_TM_IN(0), MCD_TM_IN(1), MCD_TM_IN(2), MCD_TM_IN(3), MCD_TM_IN(4), MCD_TM_IN(5), [x] "v"(x), [L] "n"(L))
#define MCD_TM(N) do { if constexpr (INIT) { if constexpr (PAD) MCD_TM##N("v_mul_f32_dpp", "=&v", "s_nop 1"); else MCD_TM##N("v_mul_f32_dpp", "=&v", ""); } \
                       else { if constexpr (PAD) MCD_TM##N("v_fmac_f32_dpp", "+v", "s_nop 1"); else MCD_TM##N("v_fmac_f32_dpp", "+v", ""); } } while (0)
template <int QC, int L, bool INIT, bool PAD>
__device__ __forceinline__ void tm_step(float (&y)[QC], const float (&c)[QC], float x) {
    static_assert(QC >= 1 && QC <= 6, "time-mix group sizes");
    if constexpr (QC == 1) {
        if constexpr (INIT) y[0] = mul_bc<L, PAD>(c[0], x);
        else fmac_bc<L, PAD>(y[0], c[0], x);
    } else if constexpr (QC == 2) {
        MCD_TM(2);
    } else if constexpr (QC == 3) {
        MCD_TM(3);
    } else if constexpr (QC == 4) {
        MCD_TM(4);
    } else if constexpr (QC == 5) {
        MCD_TM(5);
    } else {
        MCD_TM(6);
    }
}
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// zero coefficient rows behind the T real output frames of the mix tables (ragged frame groups, MixCfg::RAGGED)
constexpr int MIX_QPAD = 4;

// Joint handled by lane group g at k-step ks of the mix.  K-steps are paired over 8 consecutive joints so that the two
// lane groups sharing an LDS access phase (g = 0,1 and g = 2,3) read rows 4 apart: with a row stride of 4*odd floats
// that is a 16-bank shift, i.e. conflict-free ds_read_b32.  A trailing unpaired k-step uses joints 8p+g.
__host__ __device__ constexpr int mix_vmap(int V, int ks, int g) {
    return ks < 2 * (((V + 3) / 4) / 2) ? 8 * (ks >> 1) + 2 * (ks & 1) + 4 * (g & 1) + (g >> 1) : 8 * (((V + 3) / 4) / 2) + g;
}

// ------------------------------------------------------------------------------------------------
// mix: Z[c,q,w] = sum_v ( sum_t X[c,t,v] T[v,t,q] ) A[q,v,w]          (stsgcn.py:154-155)
// The joint mix runs on the matrix cores: for one (chain n, output frame q, block of 16 channels)
//     D[c][w] = sum_v Y[v][c] * A_q[v][w]        A operand = Y[v][c], built in registers by the time mix:
//     Y[v][c] = sum_t X[(n,t,v)][c] * T[v][t][q]  lane (j = c, g): v = 4s + g for k-step s  (T LDS reads + T FMAs)
//                                                 B operand = A_q fragments (pre-packed, from L2/L1): lane (j = w, g)
// V is padded to KS*4 rows (zero weights) and 16*MT output joints.  The D fragment (lane: output joint w = j, channels
// 4g..4g+3 of the block) is stored to Z[(n,q,w)][c .. c+3] with one ds_write_b128 (round 6; rounds 1-5 had the operands the
// other way round: a lane held one channel x four joints, four row-strided stores).
// init(n,q,w,c) seeds the accumulator (0, or the residual fragment of a W-first layer); store(n,q,w,c,val) consumes the
// result (plain Z store, or the in-place PReLU epilogue of layers 6 / 8; layer 10's two channels feed the DDPM tail).
// ------------------------------------------------------------------------------------------------
template <int CIN, int V, int T, int NB>
struct MixCfg {
    static constexpr int KS = (V + 3) / 4;
    static constexpr int KP = 2 * (KS / 2);                     // paired k-steps (see mix_vmap)
    static constexpr int MT = (V + 15) / 16;
    static constexpr int CB = CIN / 16;
    // output frames computed together by one unit: all of them (shared X reads) when that still gives every wave
    // work, otherwise one frame per unit
    // (round 4: frame counts that neither 3 nor 2 divides -- 5, 7, 11 -- take RAGGED groups, 3 + 2 / 4 + 3 / 4 + 4 + 3 frames: the
    // last unit computes one output frame too many, with zero coefficients (the tables hold MIX_QPAD zero rows behind the
    // real frames), and does not store it.  One output frame per unit, as before, meant T reads of every X value: +4.9 / +5.3 /
    // +6.7 % at 5 / 7 / 11 frames, profiles/r04m_ragged_ab.txt.)
    static constexpr int QALL = (T % 3 == 0) ? 3 : (T % 2 == 0) ? 2 : T == 5 ? 3 : (T == 7 || T == 11) ? 4 : 1;
    // the largest chunk (3, 2, 1 frames) that still gives every wave a unit; failing that, the largest one that keeps more
    // than half of them busy in a single round (e.g. 32 channels at 6 frames: 6 two-frame units -- one round, X reads shared
    // by the pair, no mid-stage coefficient fetch -- instead of 12 single-frame units in two rounds)
    static constexpr int Q2 = (T % 2 == 0 || T == 5 || T == 7 || T == 11) ? 2 : 1;
    static constexpr int units_of(int qc) { return NB * CB * ((T + qc - 1) / qc); }
    // 12 frames, 64 channels: six frames per unit -- one round of 8 units instead of two of 16, every X value read once
    // per half of the output frames (the shape has no register cap)
    static constexpr int Q6 = (T == 12) ? 6 : 1;
    // 10 / 8 frames, 64 channels: five / four frames per unit -- one round of 8 units (+2 %, profiles/r04m_ragged_ab2.txt)
    static constexpr int Q5 = (T == 10) ? 5 : 1, Q4 = (T == 8) ? 4 : 1;      // (9 frames as 5 + 4: measured equal to 3 x 3, not taken)
    // (all three frames per unit for the 32-channel mixes of the 3-frame kernel -- 4 units, one round -- measured: 0 at 12 joints,
    // -1 % with the 17-joint layers included, profiles/r04n_q32_ab.txt)
    // Measured exceptions to the rule below (one chain per workgroup, 8 waves), all of one kind: a stage lasts as long as its busiest
    // wave's units, so unit counts just above the wave count waste a round ("rounds of units"):
    //   10 / 11 frames, 32 channels: 3 + 3 + 3 + 1 / 3 + 3 + 3 + 2 (8 units, ONE round) instead of 10 / 12 two-frame units: +0.6 / +2.3 %
    //     (profiles/r04ab_mixqx_ab.txt; 11 frames at 64 channels as 6 + 5 instead of 4 + 4 + 3: 16 spilled registers, -0.2 %, not taken)
    //   9 frames: the rule picked ONE frame per unit at 32 and 16 channels (18 units = three rounds, 9 units = a second round for one
    //     unit); 3 per unit at 32 channels (6 units) and 2 at 16 channels (2 + 2 + 2 + 2 + 1) are one round each: +0.9 % and +0.7 %
    //     (3 per unit at 16 channels: +0; profiles/r04ad_t9_mixq_ab.txt)
    //   5 frames, 32 channels: 3 + 2 (4 units) instead of 2 + 2 + 1 (6 units): +1.3 %
    // and where the lever ends (-DMCD_QC16/32/64 sweep, profiles/r04ae_qc_sweep_ab.txt): 6 frames / 32 channels as 3 + 3: +0.1 %; fewer,
    // larger units at 16 channels are slower (7 / 8 frames in pairs -1.3 / -2.3 %, 12 frames in triples -0.5 %)
    static constexpr int measured_qc() {
        // 12 waves per workgroup (the 12-frame kernel, mcd_instances.hpp): 12 units per stage where the frame count allows --
        // 4 (3 below 10 frames) / 2 / 1 frames per unit at 64 / 32 / 16 channels; 10 frames at 64 channels in halves (8 units: +2.7 % over 4 + 4 + 2)
        if (NB == 1 && NWAVES == 12 && T >= 7) return CIN >= 64 ? (T == 10 && CIN == 64 ? 5 : T >= 10 ? 4 : 3) : CIN == 32 ? 2 : 1;
        if (NB != 1 || NWAVES != 8) return 0;
        if (CIN == 32 && (T == 5 || T == 9 || T == 10 || T == 11)) return 3;
        if (CIN == 16 && T == 9) return 2;
        return 0;
    }
    // (tuning builds: -DMCD_QC16= / -DMCD_QC32= / -DMCD_QC64= force the frames per unit of the 16- / 32- / 64-channel mixes)
#ifndef MCD_QC16
#define MCD_QC16 0
#endif
#ifndef MCD_QC32
#define MCD_QC32 0
#endif
#ifndef MCD_QC64
#define MCD_QC64 0
#endif
    static constexpr int QF = CIN == 16 ? MCD_QC16 : CIN == 32 ? MCD_QC32 : CIN == 64 ? MCD_QC64 : 0;
    static constexpr int QX = QF > 0 ? QF : measured_qc();
    static constexpr int QC = QX > 0 ? QX : (Q6 > 1 && units_of(Q6) >= NWAVES) ? Q6 : (Q5 > 1 && units_of(Q5) >= NWAVES) ? Q5
                            : (Q4 > 1 && units_of(Q4) >= NWAVES) ? Q4 : units_of(QALL) >= NWAVES ? QALL : units_of(Q2) >= NWAVES ? Q2
                            : 2 * units_of(QALL) > NWAVES ? QALL : 2 * units_of(Q2) > NWAVES ? Q2 : 1;
    static constexpr int NQ = (T + QC - 1) / QC;
    static constexpr bool RAGGED = NQ * QC != T;                // the last chunk of a chain holds fewer than QC frames
    static_assert(NQ * QC - T < MIX_QPAD, "zero rows behind the coefficient tables");
    static constexpr int UNITS = NB * CB * NQ;                  // one unit = (chain, 16-channel block, frame chunk)
    static constexpr int PER = (UNITS + NWAVES - 1) / NWAVES;   // rounds
    static constexpr int NR = (KS * T + 15) / 16;               // VGPRs holding the time-mix coefficients of one q
    // 12 single-frame units on 8 waves (the 32-channel layers at T = 3, NB = 2): four waves take two units.  SAMEQ gives
    // those waves two units of the SAME output frame -- waves 0-3: frame w/2, groups 2(w&1) + round; waves 4-7: frame 2,
    // group w-4 -- so the coefficients (which depend on the frame only) serve both rounds and nothing is fetched mid-stage
    // (round 4: the same map at 9 / 11 frames and 64 channels -- 3 chunks x 4 blocks -- +0.8 / +1.1 % and 18 / 24 registers less,
    // profiles/r04ac_sameqx_ab.txt)
    static constexpr bool SAMEQ = NQ == 3 && UNITS == 12 && NWAVES == 8;
    // a stage lasts as long as its busiest wave's units: a partial extra round (9 units on 8 waves) costs a whole one.  Every
    // instantiated shape is either one round, full rounds, or the SAMEQ pair map -- a new shape that is not has to pick its QC here
    static_assert(T > 12 || PER == 1 || UNITS % NWAVES == 0 || SAMEQ, "mix units: a partial round of units (see MixCfg::QX)");     // (T > 12: the condition encoders of 13 .. 20 frames, < 1 % of a step)
    // unit (frame chunk index, group = chain * CB + channel block) of (wave, round); u < 0: none
    __device__ static __forceinline__ int unit_of(int wave, int round) {
        if constexpr (SAMEQ) {
            if (wave < 4) return (wave >> 1) + NQ * ((wave & 1) * 2 + round);
            return round == 0 ? 2 + NQ * (wave - 4) : -1;
        } else {
            const int u = wave + round * NWAVES;
            return u < UNITS ? u : -1;
        }
    }
};
// time-mix rows + joint-mix A fragments of one unit.  Loaded one stage ahead of their use (behind the barrier of the
// previous stage their ~L2 latency would sit on the critical path of every mix).
template <int CIN, int V, int T, int NB>
struct MixCoef {
    using M = MixCfg<CIN, V, T, NB>;
    float tq[M::QC][M::NR], aop[M::QC][M::MT][M::KS];
    // the coefficients of the wave's first unit
    __device__ __forceinline__ void load(const float* tqd, const float* af, int wave, int lane) {
        load_unit(tqd, af, M::unit_of(wave, 0), lane);
    }
    __device__ __forceinline__ void load_unit(const float* tqd, const float* af, int u, int lane) {
        gfloat* tqd_g = as_global(tqd);
        gfloat* af_g = as_global(af);
        const int uc = u < 0 ? 0 : (u < M::UNITS ? u : M::UNITS - 1);
        const int q0 = (uc % M::NQ) * M::QC;
#pragma unroll
        for (int qi = 0; qi < M::QC; ++qi) {
#pragma unroll
            for (int r = 0; r < M::NR; ++r) tq[qi][r] = tqd_g[((q0 + qi) * M::NR + r) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < M::MT; ++mt)
#pragma unroll
                for (int ks = 0; ks < M::KS; ++ks) aop[qi][mt][ks] = af_g[(((q0 + qi) * M::MT + mt) * M::KS + ks) * 64 + lane];
        }
    }
};

// channel of a lane inside a mix unit: 16 x channel block (wave-uniform: stays on the scalar unit in address sums) + lane's channel
struct ChIdx {
    int cb16, j;
    __device__ __forceinline__ operator int() const { return cb16 + j; }
};
// init functor of a mix whose accumulators start at zero (x + 0.f is not folded away: -0.0)
struct ZeroInit { __device__ __forceinline__ float operator()(int, int, int, int) const { return 0.f; } };
// FORCE (kernels without a register cap): the unit's X reads are pinned in front of its arithmetic (the scheduler otherwise
// sinks each k-step's reads to their first use and the wave pays an LDS round trip per k-step)
#ifndef MCD_ALLW_CAPPED
#define MCD_ALLW_CAPPED 1      // (+1.3 % at 3 frames, +0.9 % at 6: profiles/r05l_capped_ab.txt) tuning: the compile-time "every wave has a unit" of ALLW in the register-capped trajectory kernels too
#endif
#ifndef MCD_RS_FULL
#define MCD_RS_FULL 1          // tuning: resample_stage's FULL
#endif
template <int CIN, int V, int T, int NB, bool FORCE = false, bool SCORE = false, class Init, class Store>
__device__ __forceinline__ void mix_stage(const float* __restrict__ in, int cs_in, const MixCoef<CIN, V, T, NB>& pre,
                                          const float* __restrict__ tqd, const float* __restrict__ af, int wave, int lane,
                                          Init&& init, Store&& store) {
    using M = MixCfg<CIN, V, T, NB>;
    constexpr int KS = M::KS, KP = M::KP, MT = M::MT, CB = M::CB, QC = M::QC, NQ = M::NQ, PER = M::PER;
    constexpr bool XB32 = (FORCE || (SCORE && MCD_XB32_CAPPED)) && MCD_XB32;
    const int j = lane & 15, g = lane >> 4;
    const int voff_pair = 4 * (g & 1) + (g >> 1);
    // the X values of one unit: x[ks][t] = X[(n, t, joint of (ks, lane group))][channel cb*16 + j]
    // Addresses: the unit's part (chain, channel block: wave-uniform) is summed on the scalar unit, the lane's part is ONE
    // v_mad_u32_u24 (lane group x row stride + channel), the reads sit at instruction offsets from their sum.  Written as one
    // product (n T V + voff) x stride the compiler multiplied per unit and rebuilt the sum on the VALU: 6 instructions per unit
    // against 3 -- and integer VALU instructions cost the matrix pipe as much as floating-point ones (tools/ubench/mix_ceiling.hip)
    auto load_x = [&](int u, float (&x)[KS][T]) {
        const int rest = u / NQ;
        const int cb = rest % CB, n = rest / CB;
        unsigned ua = lds_addr(in) + 4u * (unsigned)(n * (T * V) * cs_in + cb * 16);          // scalar unit
        asm volatile("" : "+s"(ua));      // (opaque: the region's offset stays in this sum -- split off into the reads' offset fields it costs the first pair a re-basing add too)
        const lds_float* ub = (const lds_float*)(uintptr_t)ua;
        const lds_float* a_p = ub + (__mul24(voff_pair, cs_in) + j);
        // the trailing unpaired k-step reads joints 4 KP + g.  At 17 joints only g = 0 names a joint (16): the other lane groups'
        // coefficients are zero (pack_mix_mfma), so every group reads joint 16 and the lane's part is its channel alone
        const lds_float* a_l = (V == 17) ? ub + j : ub + (__mul24(g, cs_in) + j);
        static_for<KS>([&](auto si) {
            constexpr int ks = decltype(si)::value;
            constexpr int vbase = ks < KP ? 8 * (ks >> 1) + 2 * (ks & 1) : 4 * KP;
            const lds_float* xb = ks < KP ? a_p : a_l;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                // XB32 (kernels whose reads are pinned anyway): single ds_read_b32 with 16-bit offsets.  The compiler pairs the reads
                // into ds_read2_b32 (8-bit offsets) and re-bases the address with a v_add for all but the first pair: the same
                // number of instructions, but the adds take VALU issue time, which is what the mixes of these kernels are bound by
                if constexpr (XB32) x[ks][t] = ((const volatile lds_float*)xb)[(t * V + vbase) * cs_in];
                else x[ks][t] = xb[(t * V + vbase) * cs_in];
            }
        });
    };
    auto unit = [&](const MixCoef<CIN, V, T, NB>& cur, int u, const float (&xs)[KS][T]) {
        const int q0 = (u % NQ) * QC, rest = u / NQ;
        const int cb = rest % CB, n = rest / CB;
        // V = 17: output joint 16 would cost a whole second m-tile (15/16 wasted, and the kernel is bound by matrix-pipe
        // time); it is accumulated with plain FMAs instead -- 5 per frame, partial sums over this lane group's joints
        constexpr bool J16 = V == 17;
        constexpr int MTM = J16 ? 1 : MT;            // m-tiles on the matrix cores
        // Fragment layout (round 6): the joint mix runs as D[c][w] = sum_v Y[v][c] A_q[v][w] -- the time-mixed Y as the A operand
        // (lane (j, g) = Y[joint of (ks, g)][channel j], the registers it is built in), the pre-packed A_q fragments as the B
        // operand (lane (j, g) = A_q[joint of (ks, g)][w = 16 mt + j]: the same words as before, the operands swapped) -- so that
        // a lane's D fragment holds ONE output joint w = 16 mt + j and FOUR CONSECUTIVE CHANNELS cb*16 + 4g .. + 3: one
        // ds_write_b128 per (unit, frame) instead of four row-strided ds_write_b32 (which the compiler pairs into ds_write2_b32 with
        // a re-basing v_add each), one ds_read_b128 for the residual fragment of a W-first layer, no per-row joint checks.
        //   init(n, q, w, ChIdx{cb16, 4g})  -> f32x4: the accumulator's start for joint w, channels cb16 + 4g .. + 3 (ZeroInit: zeros)
        //   store(n, q, w, ChIdx{cb16, 4g}, f32x4) for w < V (lanes of joints >= V are masked here);
        //   store(n, q, 16, ChIdx{cb16, j}, float): joint 16 of the 17-joint layers, channel cb16 + j
        constexpr bool ZINIT = std::is_same_v<std::decay_t<Init>, ZeroInit>;
        static_assert(ZINIT || !J16, "a seeded accumulator (W-first layers) on a 17-joint layer: joint 16's seed is not wired up");
        f32x4 acc[QC][MTM];
        float part[QC];
#pragma unroll
        for (int qi = 0; qi < QC; ++qi) {
            part[qi] = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTM; ++mt)
                if ((M::RAGGED && q0 + qi >= T) || ZINIT) {
                    acc[qi][mt] = f32x4{0.f, 0.f, 0.f, 0.f};      // (the frame behind a ragged chain's last: computed on zero coefficients, never stored)
                } else {
                    acc[qi][mt] = init(n, q0 + qi, mt * 16 + j, ChIdx{cb * 16, 4 * g});
                }
        }
        static_for<KS>([&](auto si) {
            constexpr int ks = decltype(si)::value;
            const float (&x)[T] = xs[ks];
            // y[qi] = sum_t X[t, v] * T[v, t, q0 + qi]   (coefficient (ks,t) = lane ks*T+t of the DPP row): the QC chains advance
            // together, one time step per asm statement (tm_step)
            float y[QC];
            static_for<T>([&](auto ti) {
                constexpr int t = decltype(ti)::value;
                float c[QC];
#pragma unroll
                for (int qi = 0; qi < QC; ++qi) c[qi] = cur.tq[qi][(ks * T + t) / 16];
                tm_step<QC, (ks * T + t) % 16, t == 0, t == T - 1>(y, c, x[t]);
            });
            static_for<QC>([&](auto qq) {
                constexpr int qi = decltype(qq)::value;
#pragma unroll
                for (int mt = 0; mt < MTM; ++mt)
                    acc[qi][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(y[qi], cur.aop[qi][mt][ks], acc[qi][mt], 0, 0, 0);
                if constexpr (J16) part[qi] = fmaf(cur.aop[qi][1][ks], y[qi], part[qi]);
            });
        });
#pragma unroll
        for (int qi = 0; qi < QC; ++qi) {
            if (M::RAGGED && q0 + qi >= T) continue;
#pragma unroll
            for (int mt = 0; mt < MTM; ++mt) {
                if ((mt + 1) * 16 <= V || mt * 16 + j < V) store(n, q0 + qi, mt * 16 + j, ChIdx{cb * 16, 4 * g}, acc[qi][mt]);
            }
            if constexpr (J16) {
                // sum the four lane groups' partials (lanes j, j+16, j+32, j+48): two register-swap steps
                const unsigned u = __float_as_uint(part[qi]);
                const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                const unsigned v2 = __float_as_uint(__uint_as_float(h[0]) + __uint_as_float(h[1]));
                const auto f = __builtin_amdgcn_permlane16_swap(v2, v2, false, false);
                const float z16 = __uint_as_float(f[0]) + __uint_as_float(f[1]);
                if (g == 0) store(n, q0 + qi, 16, ChIdx{cb * 16, j}, z16);
            }
        }
    };
    // later rounds: coefficients fetched one round ahead where the register budget allows (T = 3), else in place
    {
        MixCoef<CIN, V, T, NB> cur = pre;
        static_for<PER>([&](auto ri) {
            constexpr int rnd = decltype(ri)::value;
            // (full rounds: every wave has a unit -- said at compile time in the kernels without a register cap; in the condition
            // encoders' instantiations the same shortcut trips an "Unsupported instruction" abort of this compiler's backend)
            constexpr bool ALLW = (FORCE || (SCORE && MCD_ALLW_CAPPED)) && !M::SAMEQ && M::UNITS == PER * NWAVES;
            const int u = ALLW ? wave + rnd * NWAVES : M::unit_of(wave, rnd);
            constexpr bool SURE = ALLW;       // (the SAMEQ map's first round gives every wave a unit as well: said too, -0.3 %, profiles/r05n_sameq_ab.txt)
            float xs[KS][T];
            load_x(SURE ? u : (u < 0 ? 0 : u), xs);
            if constexpr (FORCE) __builtin_amdgcn_sched_barrier(0);
            MixCoef<CIN, V, T, NB> nxt;
            if constexpr (rnd + 1 < PER && !M::SAMEQ && NQ > 1) nxt.load_unit(tqd, af, M::unit_of(wave, rnd + 1), lane);
            if (SURE || u >= 0) unit(cur, u, xs);       // (full rounds: every wave has a unit, said at compile time)
            if constexpr (rnd + 1 < PER && !M::SAMEQ && NQ > 1) cur = nxt;       // (one frame group: every unit has the same coefficients)
        });
    }
}
// coefficients loaded at the top of the stage itself (condition encoder)
template <int CIN, int V, int T, int NB, class Init, class Store>
__device__ __forceinline__ void mix_stage(const float* __restrict__ in, int cs_in, const float* __restrict__ tqd,
                                          const float* __restrict__ af, int wave, int lane, Init&& init, Store&& store) {
    MixCoef<CIN, V, T, NB> mc;
    mc.load(tqd, af, wave, lane);
    mix_stage<CIN, V, T, NB>(in, cs_in, mc, tqd, af, wave, lane, init, store);
}

// ------------------------------------------------------------------------------------------------
// joint resampling (CNN_layer over the joint axis, BN folded): out[n,c,t,v'] = b[v'] + sum_v W[v',v] X[n,c,t,v]
// on the matrix cores, one unit = (frame (n,t), 16-channel block):  D[v'][c] = sum_v W[v'][v] X[v][c].
//   A operand: W fragments (pre-packed), B operand: X rows straight from LDS (one ds_read_b32 per k-step).
// CAPTURE (down-samplers): the k-map is v = 4g + ks (ks < 4), 16 + g (ks = 4), so the B values a lane reads are
//   exactly the rows {4g..4g+3 (,16)} it would own in an MFMA D fragment -> they are returned in `skip`
//   (this IS the U-Net skip tensor d1 / d2, kept in registers; rows 4 apart -> conflict-free reads).
// ADD (up-samplers): `skip` (captured by the matching down-sampler with the same unit -> wave mapping) is added
//   to the D fragment before the store: no separate skip-add stage, no extra barrier.
// ------------------------------------------------------------------------------------------------
template <int C, int VIN, int VOUT, int T, int NB, bool CAPTURE>
struct RsCfg {
    static constexpr int KS = CAPTURE ? (VIN > 16 ? 5 : 4) : (VIN + 3) / 4;
    static constexpr int MT = (VOUT + 15) / 16;
    static constexpr int CB = C / 16;
    static constexpr int UNITS = NB * T * CB;
    static constexpr int PER = (UNITS + NWAVES - 1) / NWAVES;   // units per wave
    static constexpr int VS = CAPTURE ? VIN : VOUT;              // joints of the skip tensor
    static constexpr int SK = VS > 16 ? 5 : 4;                   // skip registers per unit
    // ALIGNED: one wave per (chain, 16-channel block), doing that block's T frames -- the same wave then owns the unit
    // (chain, block, all frames) of the mix before / after it, so no barrier is needed between the two stages
    static constexpr bool ALIGNED = NB * CB == NWAVES && PER == T;
};
__host__ __device__ constexpr int rs_vmap(bool capture, int vin, int ks, int g) {
    return capture ? (ks < 4 ? 4 * g + ks : 16 + g) : mix_vmap(vin, ks, g);
}

// resampler weights (A fragments) + bias of one wave, loaded one stage ahead
template <int C, int VIN, int VOUT, int T, int NB, bool CAPTURE>
struct RsCoef {
    using RC = RsCfg<C, VIN, VOUT, T, NB, CAPTURE>;
    float aop[RC::MT][RC::KS];
    float bias[RC::MT][4];
    __device__ __forceinline__ void load(const float* wf, const float* bdp, int lane) {
        gfloat* wf_g = as_global(wf);
        gfloat* bdp_g = as_global(bdp);
        const int g = lane >> 4;
#pragma unroll
        for (int mt = 0; mt < RC::MT; ++mt)
#pragma unroll
            for (int ks = 0; ks < RC::KS; ++ks) aop[mt][ks] = wf_g[(mt * RC::KS + ks) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < RC::MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) bias[mt][r] = bdp_g[mt * 16 + 4 * g + r];
    }
};

// ILP (kernels with two waves per SIMD): the wave's units advance together, k-step by k-step -- PER independent
// accumulator chains instead of PER dependent 4-5-MFMA chains one after the other (40 cycles of latency per link against 32
// of issue, and nothing else on the SIMD half of the time)
// SQMAP (round 6): the units follow the SAMEQ unit map of the 12-unit mixes (MixCfg::unit_of: 12 (chain, block, frame) units on 8
// waves) instead of wave + i NWAVES -- the wave that mixes frame q of (chain, block) in a W-first layer's mix then resamples
// exactly that frame, so no barrier separates the two stages; the matching down-sampler uses the same map so that the skip
// registers it captures meet the up-sampler's units.
template <int C, int VIN, int VOUT, int T, int NB, bool CAPTURE, bool ADD, bool ILP = false, bool SQMAP = false, int NSK>
__device__ __forceinline__ void resample_stage(const float* __restrict__ in, int cs_in, float* __restrict__ out, int cs_out,
                                               const RsCoef<C, VIN, VOUT, T, NB, CAPTURE>& rc,
                                               float (&skip)[NSK], int wave, int lane) {
    using RC = RsCfg<C, VIN, VOUT, T, NB, CAPTURE>;
    constexpr int KS = RC::KS, MT = RC::MT, CB = RC::CB, UNITS = RC::UNITS, PER = RC::PER, SK = RC::SK;
    static_assert(!(CAPTURE || ADD) || NSK == PER * SK, "skip register count");
    constexpr int KP = 2 * (((VIN + 3) / 4) / 2);
    const int j = lane & 15, g = lane >> 4;
    const auto& aop = rc.aop;
    const auto& bias = rc.bias;
    // all the X reads of this wave's units first (for the down-samplers they ARE the skip registers): a unit's stores
    // may alias the next unit's reads, so reading inside the unit loop would serialise the units on LDS latency
    float xr[CAPTURE ? 1 : PER][CAPTURE ? 1 : KS] = {};      // (the down-samplers read straight into `skip`)
    // (FULL: every wave has all PER units -- said at compile time, or the conditional reads cost a copy of the whole `skip` array per unit)
    constexpr bool FULL = MCD_RS_FULL && (CAPTURE || ADD) && UNITS == PER * NWAVES;      // (score_kernel's resamplers; the slab-tiled kernel's fused ones keep the run-time test, see DESIGN)
    // the mix whose unit map SQMAP follows (without SQMAP: a fixed valid instantiation, never used)
    using SM = std::conditional_t<SQMAP, MixCfg<(SQMAP ? C : 32), (SQMAP ? (CAPTURE ? VOUT : VIN) : 12), (SQMAP ? T : 3), (SQMAP ? NB : 2)>, MixCfg<32, 12, 3, 2>>;
    static_assert(!SQMAP || (SM::SAMEQ && UNITS == SM::UNITS && PER == SM::PER && !RC::ALIGNED), "SQMAP: the 12-unit SAMEQ mixes only");
    // unit slot i of this wave -> (exists, channel block, frame)
    auto unit_at = [&](int i, int& cb, int& nt) -> bool {
        if constexpr (SQMAP) {
            const int um = SM::unit_of(wave, i);
            const int uu = um < 0 ? 0 : um, grp = uu / SM::NQ;
            cb = grp % CB; nt = (grp / CB) * T + uu % SM::NQ;
            return um >= 0;
        } else {
            const int u = wave + i * NWAVES;
            cb = RC::ALIGNED ? wave % CB : u % CB; nt = RC::ALIGNED ? (wave / CB) * T + i : u / CB;
            return FULL || u < UNITS;
        }
    };
    static_for<PER>([&](auto pi) {
        constexpr int i = decltype(pi)::value;
        int cb, nt;
        if (unit_at(i, cb, nt)) {
            // addresses: the unit's part on the scalar unit, the lane's part one v_mad, the k-steps at instruction offsets
            // (see mix_stage's load_x)
            const float* ub = in + (nt * VIN * cs_in + cb * 16);
            const float* a_main = ub + (__mul24(CAPTURE ? 4 * g : 4 * (g & 1) + (g >> 1), cs_in) + j);
            // the k-step behind the main ones: joints 16 + g (capture; only joint 16 exists: every lane group reads it, the
            // others' weights are zero and their captured value is never used) or 4 KP + g
            const float* a_tail = CAPTURE ? ub + j : ub + (__mul24(g, cs_in) + j);
            static_for<KS>([&](auto si) {
                constexpr int ks = decltype(si)::value;
                constexpr bool MAIN = CAPTURE ? ks < 4 : ks < KP;
                constexpr int row0 = CAPTURE ? (ks < 4 ? ks : 16) : (ks < KP ? 8 * (ks >> 1) + 2 * (ks & 1) : 4 * KP);
                const float xv = (MAIN ? a_main : a_tail)[row0 * cs_in];
                if constexpr (CAPTURE) skip[i * SK + ks] = xv;
                else xr[i][ks] = xv;
            });
        }
    });
    constexpr bool J16 = VOUT == 17;     // output joint 16 on the VALU (partial sums per lane group, permlane-swap reduction)
    constexpr int MTM = J16 ? 1 : MT;   // instead of a second m-tile with one useful row -- same trade as in mix_stage
    // the stores of one unit (+ the skip tensor of the up-samplers)
    auto finish = [&](auto pi, f32x4 (&acc)[MTM], float part) {
        constexpr int i = decltype(pi)::value;
        int cb, nt;
        unit_at(i, cb, nt);
        if constexpr (ADD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][r] += skip[i * SK + r];
            if constexpr (SK == 5 && !J16) acc[MT - 1][0] += skip[i * SK + 4];   // joint 16: lane group g = 0, row 0 of m-tile 1
        }
        float* uo = out + (nt * VOUT * cs_out + cb * 16);
        float* zo = uo + (__mul24(4 * g, cs_out) + j);
#pragma unroll
        for (int mt = 0; mt < MTM; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (mt * 16 + 4 * g + r < VOUT) zo[(mt * 16 + r) * cs_out] = acc[mt][r];
        if constexpr (J16) {
            const unsigned pu = __float_as_uint(part);
            const auto h = __builtin_amdgcn_permlane32_swap(pu, pu, false, false);
            const unsigned v2 = __float_as_uint(__uint_as_float(h[0]) + __uint_as_float(h[1]));
            const auto f = __builtin_amdgcn_permlane16_swap(v2, v2, false, false);
            float z16 = __uint_as_float(f[0]) + __uint_as_float(f[1]) + bias[1][0];
            if constexpr (ADD && SK == 5) z16 += skip[i * SK + 4];          // captured by lane group g = 0 (k-step 4: joint 16 + g)
            if (g == 0) uo[16 * cs_out + j] = z16;
        }
    };
    if constexpr (ILP) {
        f32x4 acc[PER][MTM];
        float part[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            part[i] = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTM; ++mt) acc[i][mt] = f32x4{bias[mt][0], bias[mt][1], bias[mt][2], bias[mt][3]};
        }
        static_for<KS>([&](auto si) {
            constexpr int ks = decltype(si)::value;
            static_for<PER>([&](auto pi) {            // (a wave without a unit in the last round computes on zeros)
                constexpr int i = decltype(pi)::value;
                float x;
                if constexpr (CAPTURE) x = skip[i * SK + ks]; else x = xr[i][ks];
#pragma unroll
                for (int mt = 0; mt < MTM; ++mt) acc[i][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[mt][ks], x, acc[i][mt], 0, 0, 0);
                if constexpr (J16) part[i] = fmaf(aop[1][ks], x, part[i]);
            });
        });
        static_for<PER>([&](auto pi) {
            int cb_, nt_;
            if (unit_at(decltype(pi)::value, cb_, nt_)) finish(pi, acc[decltype(pi)::value], part[decltype(pi)::value]);
        });
    } else {
        static_for<PER>([&](auto pi) {
            constexpr int i = decltype(pi)::value;
            int cb_, nt_;
            if (unit_at(i, cb_, nt_)) {
                f32x4 acc[MTM];
                float part = 0.f;
#pragma unroll
                for (int mt = 0; mt < MTM; ++mt) acc[mt] = f32x4{bias[mt][0], bias[mt][1], bias[mt][2], bias[mt][3]};
                static_for<KS>([&](auto si) {
                    constexpr int ks = decltype(si)::value;
                    float x;
                    if constexpr (CAPTURE) x = skip[i * SK + ks]; else x = xr[i][ks];
#pragma unroll
                    for (int mt = 0; mt < MTM; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[mt][ks], x, acc[mt], 0, 0, 0);
                    if constexpr (J16) part = fmaf(aop[1][ks], x, part);
                });
                finish(pi, acc, part);
            }
        });
    }
}

// ------------------------------------------------------------------------------------------------
// channel GEMM on v_mfma_f32_16x16x4_f32.  D[c', col] = sum_k Wp[c', k] * B[k, col]
//   A operand (weights): lane l holds W[m0 + (l&15)][k], pre-packed as float4 per 16-channel group
//   B operand (activations in LDS [col][ch]): lane (j = l&15, g = l>>4) reads channels 16kq+4g+{0..3}
//   of column n0+j with one ds_read_b128 -> four k-steps (a merged ds_read2_b64 of two 8-byte pieces is 2-4x
//   slower: 8 LDS cycles and 2-way conflicted at these strides).  The packer applies the same K permutation.
// Wave w owns m-tile w % MT and n-tiles (w / MT) + i * (8 / MT).
// ------------------------------------------------------------------------------------------------
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int MT, int NT>
struct Tiling {
    static constexpr int MW = MT > NWAVES ? MT / NWAVES : 1;      // m-tiles per wave (sequential)
    static constexpr int NG = MT > NWAVES ? 1 : NWAVES / MT;      // waves sharing one m-tile
    static constexpr int MAXN = (NT + NG - 1) / NG;
};
// One 16x16 output tile at a time: accumulate over K (Z part from b1, X part from b2), then hand the accumulator
// fragment to `epi(i, col, c0, acc)` (i = static tile slot of this wave, col = column of this lane, c0 = first of the
// lane's 4 consecutive output channels).  The output buffer never aliases b1/b2 (3-region plan), so the epilogue
// runs right behind the tile's MFMAs and no barrier separates GEMM and epilogue.
// weight fragments of this wave's m-tile: issued early (before the barrier that precedes the GEMM) so that their L2
// latency overlaps the mix stage
template <int MT, int KQ>
__device__ __forceinline__ void load_afrags(const float4* __restrict__ wp, int wave, int lane, float4 (&a)[KQ], int mi = 0) {
    const float* wpl = reinterpret_cast<const float*>(wp + (((wave + mi * NWAVES) % MT) * KQ) * 64 + lane);
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) a[kq] = load_global4(wpl + kq * 256);
}

// FORCE: pin the read-ahead order with scheduling barriers -- only for the kernels without a register cap (the
// scheduler otherwise sinks every read to its use; with the 128-VGPR cap pinning costs spills and loses)
// cinit: what a tile's accumulators start from when the layer has no identity residual -- the folded bias of the lane's 4
// output channels (two packed adds per tile less in the epilogue), or zero
// `pre(tile slot, col, ng)`: optional values the epilogue needs from LDS (the embedding row of the tile's chain), fetched in
// FRONT of the tile's MFMA chain and handed to epi as a 7th argument -- read inside the epilogue they put an LDS round trip
// between the tile's last MFMA and its store
struct NoPre { static constexpr bool none = true; };
template <int MT, int NT, int KQ1, int KQ2, bool IDRES, bool FORCE = false, bool DUAL = FORCE, class Epi, class Pre = NoPre>
__device__ __forceinline__ void gemm_tiles(const float4 (&a)[KQ1 + KQ2], const float* __restrict__ b1, int cs1,
                                           const float* __restrict__ b2, int cs2, int wave, int lane, Epi&& epi, int mi = 0,
                                           const float4 cinit = make_float4(0.f, 0.f, 0.f, 0.f), Pre&& pre = Pre{}) {
    constexpr bool HASPRE = !std::is_same_v<std::decay_t<Pre>, NoPre>;
    constexpr int NG = Tiling<MT, NT>::NG;
    constexpr int MAXN = Tiling<MT, NT>::MAXN;
    const int mt = (wave + mi * NWAVES) % MT, ng = MT > NWAVES ? 0 : wave / MT;
    if (NWAVES % MT != 0 && MT <= NWAVES && ng >= NG) return;     // (wave counts no m-tile count divides: the waves past NG x MT have no tile)
    const int j = lane & 15, g = lane >> 4;
    const int c0 = mt * 16 + 4 * g;
    // per-lane bases of this wave's FIRST tile, once per call; tile i sits at the compile-time offset i * NG * 16 * stride
    // (a per-tile col * stride is a quarter-rate v_mul_lo_u32 plus two adds on the VALU, which shares the SIMD with the MFMAs)
    const int col0 = ng * 16 + j;
    const float* const p1b = b1 + __mul24(col0, cs1) + 4 * g;
    const float* const p2b = b2 + __mul24(col0, cs2) + 4 * g;
    constexpr int KQ = KQ1 + KQ2;
#ifndef MCD_PIPE_DEPTH
#define MCD_PIPE_DEPTH 1
#endif
#ifndef MCD_PIPE_SEED
#define MCD_PIPE_SEED (MCD_NWAVES == 12)      // (12 waves, 168 registers: the seeds with the read-ahead, 16 spilled registers less: +2 % there; 8 waves: -0.4 %)
#endif
    constexpr int DEPTH0 = KQ < 3 ? KQ : 3;
    // one 16x16 output tile: B fragments (one ds_read_b128 = 4 k-steps) fetched DEPTH0 reads ahead of the MFMAs that consume
    // them -- read right before its use each fragment exposes an LDS round trip per 4 MFMAs on this wave's matrix-pipe stream
    auto one_tile = [&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const int col = col0 + i * NG * 16;
        f32x4 c = {cinit.x, cinit.y, cinit.z, cinit.w};
        const float* p1 = p1b + i * NG * 16 * cs1;
        const float* p2 = p2b + i * NG * 16 * cs2;
        if (IDRES) {
            const float4 r = *reinterpret_cast<const float4*>(p2 - 4 * g + c0);
            c[0] = r.x; c[1] = r.y; c[2] = r.z; c[3] = r.w;
        }
        auto rd = [&](auto kk) {
            constexpr int kq = decltype(kk)::value;
            return *reinterpret_cast<const float4*>(kq < KQ1 ? p1 + kq * 16 : p2 + (kq - KQ1) * 16);
        };
        float4 buf[DEPTH0];
        static_for<DEPTH0>([&](auto dd) { buf[decltype(dd)::value] = rd(dd); });
        float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (HASPRE) pe = pre(ii, col, ng);
        if constexpr (FORCE) __builtin_amdgcn_sched_barrier(0);
        static_for<KQ>([&](auto kk) {
            constexpr int kq = decltype(kk)::value;
            const float4 u = buf[kq % DEPTH0];
            if constexpr (kq + DEPTH0 < KQ) buf[kq % DEPTH0] = rd(std::integral_constant<int, kq + DEPTH0>{});
            if constexpr (FORCE) __builtin_amdgcn_sched_barrier(0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].x, u.x, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].y, u.y, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].z, u.z, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].w, u.w, c, 0, 0, 0);
        });
        if constexpr (HASPRE) epi(ii, col, c0, c, col0, ng, pe);
        else epi(ii, col, c0, c, col0, ng);
    };
    // DUAL (the kernels with two waves per SIMD): two of the wave's n-tiles at a time, their MFMA chains interleaved.  One
    // tile is a chain of 4 KQ DEPENDENT MFMAs (40 cycles each against 32 of issue) behind an LDS round trip and in front of
    // its epilogue; alone on its SIMD half of the time, a wave leaves the matrix pipe idle for all of that.  Two independent
    // accumulators issue back to back, and the second tile's reads / the first one's epilogue overlap the other's MFMAs.
    auto two_tiles = [&](auto ia, auto ib) {
        constexpr int i0 = decltype(ia)::value, i1 = decltype(ib)::value;
        const float* p1[2] = {p1b + i0 * NG * 16 * cs1, p1b + i1 * NG * 16 * cs1};
        const float* p2[2] = {p2b + i0 * NG * 16 * cs2, p2b + i1 * NG * 16 * cs2};
        f32x4 c[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            c[h] = f32x4{cinit.x, cinit.y, cinit.z, cinit.w};
            if (IDRES) {
                const float4 r = *reinterpret_cast<const float4*>(p2[h] - 4 * g + c0);
                c[h][0] = r.x; c[h][1] = r.y; c[h][2] = r.z; c[h][3] = r.w;
            }
        }
        auto rd = [&](int h, auto kk) {
            constexpr int kq = decltype(kk)::value;
            return *reinterpret_cast<const float4*>(kq < KQ1 ? p1[h] + kq * 16 : p2[h] + (kq - KQ1) * 16);
        };
        float4 buf[2][DEPTH0];
        static_for<DEPTH0>([&](auto dd) { buf[0][decltype(dd)::value] = rd(0, dd); buf[1][decltype(dd)::value] = rd(1, dd); });
        float4 pe0 = make_float4(0.f, 0.f, 0.f, 0.f), pe1 = pe0;
        if constexpr (HASPRE) { pe0 = pre(ia, col0 + i0 * NG * 16, ng); pe1 = pre(ib, col0 + i1 * NG * 16, ng); }
        __builtin_amdgcn_sched_barrier(0);
        static_for<KQ>([&](auto kk) {
            constexpr int kq = decltype(kk)::value;
            const float4 u0 = buf[0][kq % DEPTH0], u1 = buf[1][kq % DEPTH0];
            if constexpr (kq + DEPTH0 < KQ) {
                buf[0][kq % DEPTH0] = rd(0, std::integral_constant<int, kq + DEPTH0>{});
                buf[1][kq % DEPTH0] = rd(1, std::integral_constant<int, kq + DEPTH0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].x, u0.x, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].x, u1.x, c[1], 0, 0, 0);
            c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].y, u0.y, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].y, u1.y, c[1], 0, 0, 0);
            c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].z, u0.z, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].z, u1.z, c[1], 0, 0, 0);
            c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].w, u0.w, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].w, u1.w, c[1], 0, 0, 0);
        });
        if constexpr (HASPRE) {
            epi(ia, col0 + i0 * NG * 16, c0, c[0], col0, ng, pe0);
            epi(ib, col0 + i1 * NG * 16, c0, c[1], col0, ng, pe1);
        } else {
            epi(ia, col0 + i0 * NG * 16, c0, c[0], col0, ng);
            epi(ib, col0 + i1 * NG * 16, c0, c[1], col0, ng);
        }
    };
#ifndef MCD_GEMM_PIPE
#define MCD_GEMM_PIPE 1
#endif
#ifndef MCD_GEMM_PIPE1
#define MCD_GEMM_PIPE1 0
#endif
    constexpr bool PIPE = (DUAL || MCD_GEMM_PIPE1) && MAXN >= 2 && !HASPRE && MCD_GEMM_PIPE;
    constexpr int PW = DUAL ? 2 : 1;                      // tiles per pipeline step
#ifndef MCD_PIPE_DEPTH_LONG
#define MCD_PIPE_DEPTH_LONG MCD_PIPE_DEPTH
#endif
    constexpr int PD = KQ >= 8 ? MCD_PIPE_DEPTH_LONG : MCD_PIPE_DEPTH;      // (K = 128: 32 weight registers per wave)
    constexpr int DEPTH = PIPE ? (KQ < PD ? KQ : PD) : DEPTH0;
    if constexpr (PIPE) {
        // The pairs of a wave as ONE software pipeline: a pair's first DEPTH B fragments (and identity-residual seeds) are read
        // under the previous pair's last MFMAs, and a pair's epilogue is issued behind the NEXT pair's first k-group -- between
        // two pairs the matrix pipe used to wait for an LDS round trip (reads) plus the MFMA result latency (epilogue).
        // Buffers and accumulators alternate by the pair's parity (compile-time indices: no copies).  Tile slots past the
        // wave's last tile are read like the others (never used; LDS reads past the allocation return 0) so that the stream has
        // no branches.
        constexpr bool PIN = DUAL || MCD_GEMM_PIPE1 == 2;  // (under the register cap the order is left to the scheduler)
        constexpr int NP = (MAXN + PW - 1) / PW;
        const int nv = (NT - ng + NG - 1) / NG;           // this wave's tiles (wave-uniform)
        float4 bufs[2][2][DEPTH];
        float4 seed[2][2];
        f32x4 accs[2][2];
        auto rdt = [&](int i, auto kk) {
            constexpr int kq = decltype(kk)::value;
            return *reinterpret_cast<const float4*>(kq < KQ1 ? p1b + i * NG * 16 * cs1 + kq * 16 : p2b + i * NG * 16 * cs2 + (kq - KQ1) * 16);
        };
        auto fetch = [&](auto pp, auto dd) {              // fragment dd of pair pp's tiles (+ the seeds with the last one)
            constexpr int p = decltype(pp)::value, d = decltype(dd)::value;
            if constexpr (p < NP) {
                constexpr int i0 = PW * p, i1 = (PW == 2 && i0 + 1 < MAXN) ? i0 + 1 : i0;
                bufs[p & 1][0][d] = rdt(i0, dd);
                if constexpr (i1 != i0) bufs[p & 1][1][d] = rdt(i1, dd);
                if constexpr (IDRES && d == DEPTH - 1 && MCD_PIPE_SEED) {
                    seed[p & 1][0] = *reinterpret_cast<const float4*>(p2b + i0 * NG * 16 * cs2 - 4 * g + c0);
                    if constexpr (i1 != i0) seed[p & 1][1] = *reinterpret_cast<const float4*>(p2b + i1 * NG * 16 * cs2 - 4 * g + c0);
                }
            }
        };
        auto epi_pair = [&](auto pp, auto nn) {
            constexpr int p = decltype(pp)::value, N = decltype(nn)::value, i0 = PW * p, i1 = i0 + 1;
            epi(std::integral_constant<int, i0>{}, col0 + i0 * NG * 16, c0, accs[p & 1][0], col0, ng);
            if constexpr (N == 2) epi(std::integral_constant<int, i1>{}, col0 + i1 * NG * 16, c0, accs[p & 1][1], col0, ng);
        };
        auto chain = [&](auto pp, auto nn) {
            constexpr int p = decltype(pp)::value, N = decltype(nn)::value, i0 = PW * p, i1 = N == 2 ? i0 + 1 : i0;
            f32x4 (&c)[2] = accs[p & 1];
#pragma unroll
            for (int h = 0; h < N; ++h) {
                c[h] = f32x4{cinit.x, cinit.y, cinit.z, cinit.w};
                if (IDRES) {
                    if constexpr (!MCD_PIPE_SEED) seed[p & 1][h] = *reinterpret_cast<const float4*>(p2b + (h ? i1 : i0) * NG * 16 * cs2 - 4 * g + c0);
                    c[h] = f32x4{seed[p & 1][h].x, seed[p & 1][h].y, seed[p & 1][h].z, seed[p & 1][h].w};
                }
            }
            static_for<KQ>([&](auto kk) {
                constexpr int kq = decltype(kk)::value;
                float4 u[2];
#pragma unroll
                for (int h = 0; h < N; ++h) u[h] = bufs[p & 1][h][kq % DEPTH];
                if constexpr (kq + DEPTH < KQ) {
                    bufs[p & 1][0][kq % DEPTH] = rdt(i0, std::integral_constant<int, kq + DEPTH>{});
                    if constexpr (N == 2) bufs[p & 1][1][kq % DEPTH] = rdt(i1, std::integral_constant<int, kq + DEPTH>{});
                } else if constexpr (N == PW) {           // (a short step is the wave's last)
                    fetch(std::integral_constant<int, p + 1>{}, std::integral_constant<int, kq + DEPTH - KQ>{});
                }
                if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
                c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].x, u[0].x, c[0], 0, 0, 0);
                if constexpr (N == 2) c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].x, u[1].x, c[1], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].y, u[0].y, c[0], 0, 0, 0);
                if constexpr (N == 2) c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].y, u[1].y, c[1], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].z, u[0].z, c[0], 0, 0, 0);
                if constexpr (N == 2) c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].z, u[1].z, c[1], 0, 0, 0);
                c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].w, u[0].w, c[0], 0, 0, 0);
                if constexpr (N == 2) c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kq].w, u[1].w, c[1], 0, 0, 0);
                if constexpr (kq == 0 && p > 0) {         // the previous pair's epilogue, behind this pair's first k-group
                    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
                    epi_pair(std::integral_constant<int, p - 1>{}, std::integral_constant<int, PW>{});
                    if constexpr (PIN) __builtin_amdgcn_sched_barrier(0);
                }
            });
            // the wave's last pair runs its epilogue itself; the others leave it to the pair behind them
            if (PW * (p + 1) >= nv) epi_pair(pp, nn);
        };
        static_for<DEPTH>([&](auto dd) { fetch(std::integral_constant<int, 0>{}, dd); });
        static_for<NP>([&](auto pp) {
            constexpr int i0 = PW * decltype(pp)::value, i1 = i0 + 1;
            if constexpr (PW == 2 && i1 < MAXN) {
                if (i1 < nv) chain(pp, std::integral_constant<int, 2>{});
                else if (i0 < nv) chain(pp, std::integral_constant<int, 1>{});
            } else {
                if (i0 < nv) chain(pp, std::integral_constant<int, 1>{});
            }
        });
    } else if constexpr (DUAL && MAXN >= 2) {
        static_for<(MAXN + 1) / 2>([&](auto pp) {
            constexpr int i0 = 2 * decltype(pp)::value, i1 = i0 + 1;
            if constexpr (i1 < MAXN) {
                if (ng + i1 * NG < NT) two_tiles(std::integral_constant<int, i0>{}, std::integral_constant<int, i1>{});
                else if (ng + i0 * NG < NT) one_tile(std::integral_constant<int, i0>{});
            } else {
                if (ng + i0 * NG < NT) one_tile(std::integral_constant<int, i0>{});
            }
        });
    } else {
        static_for<MAXN>([&](auto ii) {
            if (ng + decltype(ii)::value * NG < NT) one_tile(ii);
        });
    }
}

// one mix-first ST-GCN layer: LDS `in` -> `out`, with `z` as scratch; the three regions are disjoint.
// generic mix-first ST-GCN layer (CIN -> COUT at V joints), used by the U-Net and by the condition encoder.
// HASEMB = false: no embedding term (condition-encoder layers get t = None, components.py:56-63).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// weight fragments of a wave's m-tile for one layer's GEMM + the folded bias of its 4 output channels, fetched by the CALLER
// at the end of the stage before the layer (in front of that stage's closing barrier, where the older wave of each SIMD only
// waits): at the layer's top the 3 .. 9 KB-wide loads per wave of all waves queued in front of the mix's first LDS reads
template <int KQ>
struct LayerAfr {
    float4 a[KQ];
    float4 bcur;
    template <int MT>
    __device__ __forceinline__ void load(const float* wb, const LayerW& lw, int wave, int lane) {
        load_afrags<MT, KQ>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, a);
        bcur = load_global4(wb + lw.bias + (wave % MT) * 16 + 4 * (lane >> 4));
    }
};

// `mc`: this layer's mix coefficients (already loaded); `pre_gemm` runs between the mix barrier and the GEMM, `pre_barrier`
// between the GEMM and the closing barrier -- the callers use them to issue the NEXT stage's coefficient loads.
// `pre_afr`: the layer's weight fragments when the caller fetched them ahead (null: fetched here).
template <int CIN, int COUT, int V, bool RES, bool HASEMB, int T, int NB, bool FORCE = false, int CSX = cs_of(CIN), class H1, class H2>
__device__ __forceinline__ void layer_generic(const float* wb, const LayerW lw, const MixCoef<CIN, V, T, NB>& mc,
                                              const float* __restrict__ in, float* __restrict__ z, float* __restrict__ out,
                                              const float* __restrict__ embl, int wave, int lane, Prof& prof, int prof_id,
                                              H1&& pre_gemm, H2&& pre_barrier,
                                              const LayerAfr<(CIN / 16) * (RES ? 2 : 1)>* pre_afr = nullptr) {
    constexpr int MT = ceil16(COUT) / 16;
    constexpr int COLS = NB * T * V;
    constexpr int NT = ceil16(COLS) / 16;
    constexpr int TV = T * V;
    constexpr int CSI = cs_of(CIN), CSO = cs_of(COUT);
    constexpr int KQ1 = CIN / 16, KQ2 = RES ? CIN / 16 : 0;
    float4 afr[KQ1 + KQ2];
    const int trs = 8 + 8 * ((prof_id - 32) / 3);      // trace slots of this layer (profile builds)
    prof.trace(trs + 0);
    prof.pp.lat();
    const float* bias = wb + lw.bias;
    float4 bcur;
    if (pre_afr != nullptr) {
#pragma unroll
        for (int k = 0; k < KQ1 + KQ2; ++k) afr[k] = pre_afr->a[k];
        bcur = pre_afr->bcur;
    } else {
        load_afrags<MT, KQ1 + KQ2>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr);
        // (the folded bias starts the first tile's accumulators: fetched here, with the weight fragments, so that its L2
        // latency hides behind the mix as well)
        bcur = load_global4(bias + (wave % MT) * 16 + 4 * (lane >> 4));
    }
#ifndef MCD_MIX_FORCE
#define MCD_MIX_FORCE 1
#endif
    mix_stage<CIN, V, T, NB, (FORCE && MCD_MIX_FORCE), HASEMB>(in, CSX, mc, wb + lw.tq, wb + lw.am, wave, lane,
                             ZeroInit{},
                             [&](int n, int q, int w, ChIdx c, auto v) {
                                 // joint w's row, the lane's channels: the unit's part of the address on the scalar unit, the lane's
                                 // part (w CSI + channel offset) one v_mad (see load_x)
                                 float* zp = (z + (n * (T * V) * CSI + q * (V * CSI) + c.cb16)) + (__mul24(w, CSI) + c.j);
                                 if constexpr (std::is_same_v<decltype(v), f32x4>) lds_store4(lds_addr(zp), v[0], v[1], v[2], v[3]);   // 4 channels: one ds_write_b128
                                 else *zp = v;                                                                                       // joint 16, one channel
                             });
    prof.trace(trs + 1);
    // The NEXT stage's coefficient loads.  The vector-memory path accepts ~1 wave-wide load per 10 cycles and all eight waves
    // issue 10 .. 40 of them at the same point of the stage: in front of the GEMM tiles (where they used to be) the last
    // wave's first MFMA waited ~2 k cycles for its loads to be accepted (profiles/r03c_seq24_trace.txt).  They are issued
    // HERE instead, behind the wave's mix and in front of the barrier: the older wave of each SIMD reaches this point
    // 1 - 2 k cycles before the younger one and would only wait (12 frames +3 % together with EARLY2, 6 frames +1.9 %,
    // 3 frames +1.1 %: profiles/r03k_prebar_ab.txt).
    pre_gemm();
    bsync();
    prof.pp.thr();
    prof.trace(trs + 2);
    prof.mark(prof_id);
    const float slope = lw.slope;
    const float pinf = prelu_bound(slope);     // see prelu()
    constexpr int TILE_STEP = Tiling<MT, NT>::NG * 16;      // columns between a wave's consecutive n-tiles
    // the folded bias starts the accumulators of the layers with a residual convolution (identity residuals start from X)
    constexpr bool FOLD = RES;
    // Per-lane offsets of the wave's first tile in `out` and of its 4 channels in the embedding row: computed ONCE per GEMM call
    // and made opaque, so that the tiles address with immediates instead of re-deriving col * stride + c0 (3 VALU instructions
    // per tile: the compiler prefers rematerialising to holding a register)
    unsigned oaddr = 0, eaddr = 0;      // LDS byte addresses
    auto set_bases = [&](int mi) {
        const int mt = (wave + mi * NWAVES) % MT, ng = MT > NWAVES ? 0 : wave / MT;
        const int c0 = mt * 16 + 4 * (lane >> 4);
        oaddr = lds_addr(out) + 4u * (unsigned)(__mul24(ng * 16 + (lane & 15), CSO) + c0);
        eaddr = HASEMB ? lds_addr(embl) + 4u * (unsigned)c0 : 0u;
        asm volatile("" : "+v"(oaddr), "+v"(eaddr));
    };
    set_bases(0);
    // One chain per workgroup: the embedding values of a lane's 4 output channels are the same for every tile of the GEMM
    // call.  Read once up front (kernels without a register cap): inside the tile epilogue the read sits between the tile's
    // last MFMA and its store -- an LDS round trip on the wave's critical path per tile (the compiler cannot hoist it itself:
    // the epilogue's LDS stores may alias it)
    constexpr bool EHOIST = HASEMB && NB == 1 && FORCE;
    float4 e_pre = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (EHOIST) e_pre = lds_load4(eaddr);
    // the embedding values of this lane's 4 output channels for the chain of a tile's column
    auto emb_of = [&](auto ti, int col, int ng) -> float4 {
        // Two chains: a tile lies on one side of the chain boundary (wave-uniform: picked on the scalar unit) except the one
        // tile that straddles it
        if constexpr (NB == 2) {
            const int tile_lo = (ng + decltype(ti)::value * Tiling<MT, NT>::NG) * 16;
            unsigned eo = tile_lo >= TV ? 4u * EMB_STRIDE : 0u;                          // scalar unit
            if (tile_lo < TV && tile_lo + 16 > TV) {                                      // the straddling tile: per lane
                eo = col >= TV ? 4u * EMB_STRIDE : 0u;
                asm volatile("" : "+v"(eo));       // (keeps this a scalar branch: if-converted it costs every tile 5 VALU instructions)
            }
            return lds_load4(eaddr + eo);
        } else if constexpr (NB > 2) {
            const int n = col / TV;
            return lds_load4(eaddr + 4u * (unsigned)((n < NB ? n : NB - 1) * EMB_STRIDE));
        } else if constexpr (EHOIST) {
            return e_pre;
        } else {
            return lds_load4(eaddr);
        }
    };
    constexpr bool EPRE = HASEMB && NB > 1 && FORCE;       // several chains, no register cap: fetched in front of the tile's MFMAs
    auto epi = [&](auto ti, int col, int c0, f32x4 acc, int, int ng, auto... pe) {
        // pad columns (col >= COLS) are computed and stored like the others: every region has ceil16(COLS) rows, nobody reads
        // them, and no per-tile bounds check runs on the VALU.  Output channels: only COUT not a multiple of 16 needs the check.
        if (COUT % 16 == 0 || c0 < COUT) {
            float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (sizeof...(pe) > 0) e = (pe, ...);
            else if constexpr (HASEMB) e = emb_of(ti, col, ng);
            // packed adds / multiply on channel pairs (v_pk_add_f32, v_pk_mul_f32) around the four v_med3_f32 of the PReLU
            f32x2 t0 = f32x2{acc[0], acc[1]}, t1 = f32x2{acc[2], acc[3]};
            if constexpr (!FOLD) { t0 += f32x2{bcur.x, bcur.y}; t1 += f32x2{bcur.z, bcur.w}; }
            const f32x2 m0 = t0 * slope, m1 = t1 * slope;
            const f32x2 r0 = f32x2{__builtin_amdgcn_fmed3f(t0[0], m0[0], pinf), __builtin_amdgcn_fmed3f(t0[1], m0[1], pinf)} + f32x2{e.x, e.y};
            const f32x2 r1 = f32x2{__builtin_amdgcn_fmed3f(t1[0], m1[0], pinf), __builtin_amdgcn_fmed3f(t1[1], m1[1], pinf)} + f32x2{e.z, e.w};
            constexpr unsigned tile_bytes = 4u * decltype(ti)::value * TILE_STEP * CSO;
            lds_store4(oaddr + tile_bytes, r0[0], r0[1], r1[0], r1[1]);
        }
    };
    prof.trace(trs + 3);
    if constexpr (EPRE) gemm_tiles<MT, NT, KQ1, KQ2, !RES, FORCE>(afr, z, CSI, in, CSX, wave, lane, epi, 0, FOLD ? bcur : make_float4(0.f, 0.f, 0.f, 0.f), emb_of);
    else gemm_tiles<MT, NT, KQ1, KQ2, !RES, FORCE>(afr, z, CSI, in, CSX, wave, lane, epi, 0, FOLD ? bcur : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
    for (int mi = 1; mi < Tiling<MT, NT>::MW; ++mi) {     // workgroups with fewer waves than m-tiles: next m-tile(s)
        load_afrags<MT, KQ1 + KQ2>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr, mi);
        bcur = load_global4(bias + ((wave + mi * NWAVES) % MT) * 16 + 4 * (lane >> 4));
        set_bases(mi);
        if constexpr (EHOIST) e_pre = lds_load4(eaddr);
        if constexpr (EPRE) gemm_tiles<MT, NT, KQ1, KQ2, !RES, FORCE>(afr, z, CSI, in, CSX, wave, lane, epi, mi, FOLD ? bcur : make_float4(0.f, 0.f, 0.f, 0.f), emb_of);
        else gemm_tiles<MT, NT, KQ1, KQ2, !RES, FORCE>(afr, z, CSI, in, CSX, wave, lane, epi, mi, FOLD ? bcur : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    prof.trace(trs + 4);
    pre_barrier();
#ifdef MCD_PROFILE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (stamp 5 = this wave's stores have landed)
#endif
    prof.trace(trs + 5);
    bsync();
    prof.pp.lat();
    prof.trace(trs + 6);
    prof.mark(prof_id + 1);
}
// self-contained form (condition encoder): coefficients loaded at the top of the layer
template <int CIN, int COUT, int V, bool RES, bool HASEMB, int T, int NB>
__device__ __forceinline__ void layer_generic(const float* wb, const LayerW lw, const float* __restrict__ in,
                                              float* __restrict__ z, float* __restrict__ out,
                                              const float* __restrict__ embl, int wave, int lane, Prof& prof, int prof_id) {
    MixCoef<CIN, V, T, NB> mc;
    mc.load(wb + lw.tq, wb + lw.am, wave, lane);
    layer_generic<CIN, COUT, V, RES, HASEMB, T, NB>(wb, lw, mc, in, z, out, embl, wave, lane, prof, prof_id, NoHook{}, NoHook{});
}

// U-Net layer L of the fixed channel plan
template <int L, int T, int NB>
using LMix = MixCoef<layer_desc(L).cin, layer_desc(L).V, T, NB>;
template <int L>
using LAfr = LayerAfr<(layer_desc(L).cin / 16) * (layer_desc(L).res ? 2 : 1)>;
template <int L>
__device__ __forceinline__ void load_lafr(LAfr<L>& A, const float* wb, int wave, int lane) {
    A.template load<ceil16(layer_desc(L).cout) / 16>(wb, layer_w(wb, L), wave, lane);
}
template <int L, int T, int NB, bool FORCE = false, int CSX = cs_of(layer_desc(L).cin), class H1, class H2>
__device__ __forceinline__ void layer_std(const float* wb, const LMix<L, T, NB>& mc, const float* in, float* z, float* out,
                                          const float* emb, int wave, int lane, Prof& prof, H1&& pre_gemm, H2&& pre_barrier,
                                          const LAfr<L>* pre_afr = nullptr) {
    constexpr LDesc D = layer_desc(L);
    layer_generic<D.cin, D.cout, D.V, D.res != 0, true, T, NB, FORCE, CSX>(wb, layer_w(wb, L), mc, in, z, out, emb + emb_off(L), wave, lane,
                                                               prof, 32 + 3 * L, pre_gemm, pre_barrier, pre_afr);
}

// ------------------------------------------------------------------------------------------------
// layer embeddings of one pass: EMB[n][o] = b_e[o] + sum_k W_e[o][k] SiLU(pe(i) + cond_n)[k] for the 530 (+2 pad)
// output channels of the 11 layers (the Linear(SiLU(.)) of every ST-GCN layer, stsgcn.py:184-186, fed by the U-Net's
// time embedding, stsae_unet.py:173-179, 424-431).  They depend on the step and the window only -- not on x_t -- so
// pass i-1's are computed during the last (light) GEMM stage of pass i from rows fetched one stage earlier.
// Plain FMAs, one output row per thread: with 2 chains the matrix cores would run this at 1/8 utilisation and the
// kernel is bound by their time, not by the VALU's.
// ------------------------------------------------------------------------------------------------
struct EmbRow {              // W_e row + bias of one output channel
    float4 w[4];
    float b;
    __device__ __forceinline__ void load(const float* wb, int o) {
        const float* we = wb + tab_i(wb, TAB_WE);
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = load_global4(we + o * EDIM + 4 * q);
        b = as_global(wb + tab_i(wb, TAB_BE))[o];
    }
};
// se: SiLU(pe + cond) [NB][16] in LDS; emb: EMB[n][536] (layers 0..9); e10: layer 10's outputs [n][4]
template <int NB>
__device__ __forceinline__ void emb_row(const EmbRow& f, int o, const float* __restrict__ se, float* __restrict__ emb,
                                        float* __restrict__ e10) {
    // even / odd k partial sums per packed FMA (weight pairs and SiLU pairs are adjacent registers / LDS words: no
    // broadcast operand, which the compiler otherwise builds with extra v_mov): 8 v_pk_fma_f32 + 1 add per chain
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        f32x2 acc = {f.b, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 sv = *reinterpret_cast<const float4*>(se + n * EDIM + 4 * q);   // LDS broadcast
            acc = f32x2{f.w[q].x, f.w[q].y} * f32x2{sv.x, sv.y} + acc;
            acc = f32x2{f.w[q].z, f.w[q].w} * f32x2{sv.z, sv.w} + acc;
        }
        const float r = acc[0] + acc[1];
        if (o < emb_off(10)) emb[n * EMB_STRIDE + o] = r;
        else if (o < EMB_TOTAL) e10[n * 4 + (o - emb_off(10))] = r;
    }
}
// thread tid owns output channel tid (row `f`, fetched from L2 a stage ahead) and, for the EMB_TOTAL - NTHREADS channels
// beyond, tid + NTHREADS: those few rows are kept in LDS (exw[row][20]: 16 weights, bias; copied once per workgroup) -- as
// a second register row per thread they were fetched right before their use and the first wave waited an L2 round trip
// for them in front of the stage's barrier
constexpr int EMB_EXTRA = EMB_TOTAL > NTHREADS ? EMB_TOTAL - NTHREADS : 0;
template <int NB>
__device__ __forceinline__ void emb_compute(const EmbRow& f, const float* __restrict__ exw, const float* __restrict__ se,
                                            float* __restrict__ emb, float* __restrict__ e10, int tid) {
    static_assert(EMB_TOTAL <= 2 * NTHREADS, "two rows per thread cover the embedding outputs");
    emb_row<NB>(f, tid, se, emb, e10);
    if (tid < EMB_EXTRA) {
        EmbRow f2;
#pragma unroll
        for (int q = 0; q < 4; ++q) f2.w[q] = *reinterpret_cast<const float4*>(exw + tid * 20 + 4 * q);
        f2.b = exw[tid * 20 + 16];
        emb_row<NB>(f2, tid + NTHREADS, se, emb, e10);
    }
}

// LDS plan: one work region R carved per layer into disjoint (in, z, out) pieces + the persistent x_t / embedding
// tables.  Sizes follow the padded column counts P17/P12/P10 and the row strides C+4.
template <int T, int NB>
struct Plan {
    static constexpr int NBT = NB * T;
    static constexpr int P17 = ceil16(NBT * 17), P12 = ceil16(NBT * 12), P10 = ceil16(NBT * 10);
    static constexpr int s16 = P17 * 20, s32a = P17 * 36, s32b = P12 * 36, s64b = P12 * 68, s64c = P10 * 68, s128 = P10 * 132;
    // (in, z, out) offsets of every stage
    static constexpr int L0_in = 0, L0_z = s16, L0_out = 2 * s16;
    static constexpr int L1_in = 2 * s16, L1_z = 0, L1_out = 3 * s16;
    static constexpr int L2_in = 3 * s16, L2_z = 0, L2_out = 3 * s16 + s32a;
    static constexpr int DN1_out = 0;
    static constexpr int L3_in = 0, L3_z = s32b, L3_out = 2 * s32b;
    static constexpr int L4_in = 2 * s32b, L4_z = 2 * s32b + s64b, L4_out = 0;
    static constexpr int DN2_out = s128 + s64c;
    static constexpr int L5_in = s128 + s64c, L5_z = s128, L5_out = 0;
    static constexpr int L6_in = 0, L6_p = s128;
    static constexpr int UP3_out = 0;
    static constexpr int L7_in = 0, L7_z = s64b, L7_out = 2 * s64b;
    static constexpr int L8_in = 2 * s64b, L8_z = 0, L8_out = s64b;
    static constexpr int L8_p = 0;              // W-first layer 8 (MCD_L8_WFIRST): P = [P_t | P_r] [P12][68] where z was; the layer's output replaces P_r
    static constexpr int UP2_out = cmax(s64b + s32b, 2 * s32a);      // behind layer 8's output and layer 9's (z, out)
    static constexpr int L9_in = UP2_out, L9_z = 0, L9_out = s32a;
    static constexpr int L10_in = s32a, L10_p = 2 * s32a;
    static constexpr int R = cmax(cmax(cmax(s128 + 2 * s64c, 2 * s128), cmax(3 * s16 + 2 * s32a, 2 * s32b + 2 * s64b)),
                                  cmax(cmax(3 * s64b, 3 * s32a), UP2_out + s32a));
    static_assert(s32a <= 3 * s16 && s64b <= 2 * s32b && s32a <= s64b && s64b <= s128, "LDS plan: regions would overlap");
    static constexpr int XT = P17 * 4;
    static constexpr int EMB = NB * EMB_STRIDE;
    static constexpr int EAUX = 2 * 4 * 4 + 4 * EDIM;   // layer 10's embedding outputs, double-buffered by step parity: [2][NB<=4][4];
                                                     // then SiLU(pe + cond) of the NEXT pass [NB<=4][16]
    static constexpr int ZN = P17 * 2;          // this step's DDPM noise z[col][c]
    static constexpr int WM = 12;               // per chain (NB <= 4): condition-frame bitmask [0,4), window [4,8), sample [8,12)
    static constexpr int BIA = 64 + 16 + 32;    // biases of the W-first layers (6: 64, 10: 2 (+ pad), 8: 32), read inside their store functors
    static constexpr int UPD = 16;              // per (chain, U-Net frame): first column of the frame its prediction updates, or -1
    static constexpr int ZO = P17 * 2;          // layer 10's mixed output Z[col][c] between its mix and the element-wise tail
    static constexpr int TT = P17 * 2;          // per (column, coordinate) of the element-wise tail: packed (chain, frame, joint) indices
    static constexpr int CE = 4 * EDIM;         // condition embeddings of the workgroup's windows [NB <= 4][16]
    static constexpr int LOSS = NB * 64;        // per-sample losses of the workgroup's windows [NB][S <= 64] (in-kernel aggregation)
    static constexpr int EXW = EMB_EXTRA * 20;  // embedding rows beyond the first NTHREADS: [row][16 weights, bias, pad]
#ifdef MCD_PROFILE
    static constexpr int PROFTR = (NB * T >= 10 && NWAVES <= 12) ? PROF_TRACE * PROF_NW : 0;    // time stamps: the one-workgroup-per-CU shapes have the room (12 waves: 6 KB of the 14.8 KB the 12-frame plan leaves)
    static constexpr int PROF = PROF_SLOTS + PROFTR;
#else
    static constexpr int PROF = 0;
#endif
    static constexpr int TOTAL = R + XT + EMB + EAUX + ZN + WM + BIA + UPD + ZO + TT + CE + LOSS + EXW + PROF;
    static_assert((size_t)TOTAL * 4 <= 160 * 1024, "LDS plan: more than 160 KB");
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
};

// data frames a condition encoder reads, in order
struct FrameIdx { int idx[MCD_MAX_FRAMES]; };
// U-Net frame layout of a scoring call for the kernels that take it at run time (more than 12 frames)
struct FrameMaps { int src_frame[MCD_MAX_FRAMES], tx_of[MCD_MAX_FRAMES], pos_of[MCD_MAX_FRAMES], upd_of[MCD_MAX_FRAMES]; };
__device__ __forceinline__ float loss_elem(float a, float b, int fn) {
    const float d = fabsf(a - b);
    if (fn == MCD_LOSS_SMOOTH_L1) return d < 1.f ? 0.5f * d * d : d - 0.5f;
    if (fn == MCD_LOSS_L1) return d;
    return d * d;
}

}  // namespace mcd
