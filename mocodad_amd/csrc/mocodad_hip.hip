// mocodad_hip.hip — MI355X (gfx950 / CDNA4) kernels + C ABI for the MoCoDAD anomaly-scoring path.
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
//   W-first   layers 6 and 10: GEMM first, then the mix on the (fewer) output channels with the layer epilogue
//             (layer 10: + U-Net residual + DDPM update) in the mix's store functor
// Everything is fp32 (the reverse chain amplifies error by up to 1e3).

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

namespace {

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
__device__ __forceinline__ unsigned lds_addr(const float* p) { return (unsigned)(uintptr_t)(lds_float*)p; }
__device__ __forceinline__ float4 lds_load4(unsigned addr) {
    const f32x4 v = *(lds_f32x4*)(uintptr_t)addr;
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_store4(unsigned addr, float a, float b, float c, float d) {
    *(lds_f32x4*)(uintptr_t)addr = f32x4{a, b, c, d};
}
__device__ __forceinline__ float4 load_global4(const float* p) {       // 16-byte aligned global load
    const f32x4 v = *(gf32x4*)p;
    return make_float4(v[0], v[1], v[2], v[3]);
}

#ifndef MCD_NWAVES
#define MCD_NWAVES 8
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
enum { F_TQ = 0, F_AM = 1, F_WP = 2, F_BIAS = 3, F_SLOPE = 4, F_WPB = 5, F_STRIDE = 8 };
//   tab[l*8 + F_TQ]    time-mix coefficients packed 16 per VGPR for DPP row broadcast, TQD[q][r][64]:
//                      lane 16g+i = gcn.T[v = mix_vmap(s,g)][t][q] with s*T+t = 16r+i
//   tab[l*8 + F_AM]    MFMA A-operand fragments of A_q^T, AF[q][mt][s][64]: lane (i, g) = gcn.A[q][v=mix_vmap(s,g)][w=16mt+i]
//   tab[l*8 + F_WP]    MFMA-packed [W_t' | W_r'] (layers 6, 10: [W_t' ; W_r'] stacked, W-first)
//   tab[l*8 + F_BIAS]  folded bias, padded to 16
//   tab[l*8 + F_SLOPE] PReLU slope (float bits)
constexpr int TAB_WE = 88, TAB_BE = 89;   // WeAll[EMB_TOTAL][16], beAll[EMB_TOTAL]
constexpr int TAB_RSW = 90, TAB_RSB = 94; // down1, down2, up3, up2: MFMA A fragments WF[mt][ks][64] (lane (i,g) =
                                          // Wd'[16mt+i][rs_vmap(ks,g)]) and bd' padded to 32
typedef const int __attribute__((address_space(4))) cint;
__device__ __forceinline__ int tab_i(const float* base, int idx) { return ((cint*)base)[idx]; }
__device__ __forceinline__ float tab_f(const float* base, int idx) { return ((cfloat*)base)[idx]; }
struct LayerW { int tq, am, wp, bias; float slope; int wpb; };   // wpb: split-bf16 fragments (layers 2..9), see gemm_tiles_bf3
__device__ __forceinline__ LayerW layer_w(const float* base, int l) {
    LayerW w;
    w.tq = tab_i(base, l * F_STRIDE + F_TQ); w.am = tab_i(base, l * F_STRIDE + F_AM);
    w.wp = tab_i(base, l * F_STRIDE + F_WP); w.bias = tab_i(base, l * F_STRIDE + F_BIAS);
    w.slope = tab_f(base, l * F_STRIDE + F_SLOPE);
    w.wpb = tab_i(base, l * F_STRIDE + F_WPB);
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
struct Prof {
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
    __device__ __forceinline__ void off() { on = false; won = false; acc = nullptr; tlast = 0; bidx = 0; wv = 0; tr = nullptr; tr_on = false; }
#else
    __device__ __forceinline__ void trace(int) {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void sync() { __syncthreads(); }
    __device__ __forceinline__ void off() {}
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
    float aggr_q;
    float* loss_agg;          // (B,) aggregated loss, or null
    int cond_idx[12];         // cond_inkernel: data frames the condition encoder reads
    int upd_shift;            // some prediction updates a frame other than the one it is read at (element-wise tail: barrier between reads and writes)
    int fixed_mask;           // bit t: U-Net frame t is a condition frame copied from the window (concat / imputation)
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
#define MCD_TM6(OP, CON, NOP) asm(MCD_DPP(OP, 0) MCD_DPP(OP, 1) MCD_DPP(OP, 2) MCD_DPP(OP, 3) MCD_DPP(OP, 4) MCD_DPP(OP, 5) NOP \
                                  : [y0] CON(y[0]), [y1] CON(y[1]), [y2] CON(y[2]), [y3] CON(y[3]), [y4] CON(y[4]), [y5] CON(y[5]) \
                                  : MCD_TM_IN(0), MCD_TM_IN(1), MCD_TM_IN(2), MCD_TM_IN(3), MCD_TM_IN(4), MCD_TM_IN(5), [x] "v"(x), [L] "n"(L))
#define MCD_TM(N) do { if constexpr (INIT) { if constexpr (PAD) MCD_TM##N("v_mul_f32_dpp", "=&v", "s_nop 1"); else MCD_TM##N("v_mul_f32_dpp", "=&v", ""); } \
                       else { if constexpr (PAD) MCD_TM##N("v_fmac_f32_dpp", "+v", "s_nop 1"); else MCD_TM##N("v_fmac_f32_dpp", "+v", ""); } } while (0)
template <int QC, int L, bool INIT, bool PAD>
__device__ __forceinline__ void tm_step(float (&y)[QC], const float (&c)[QC], float x) {
    static_assert(QC == 1 || QC == 2 || QC == 3 || QC == 4 || QC == 6, "time-mix group sizes");
    if constexpr (QC == 1) {
        if constexpr (INIT) y[0] = mul_bc<L, PAD>(c[0], x);
        else fmac_bc<L, PAD>(y[0], c[0], x);
    } else if constexpr (QC == 2) {
        MCD_TM(2);
    } else if constexpr (QC == 3) {
        MCD_TM(3);
    } else if constexpr (QC == 4) {
        MCD_TM(4);
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

// Joint handled by lane group g at k-step ks of the mix.  K-steps are paired over 8 consecutive joints so that the two
// lane groups sharing an LDS access phase (g = 0,1 and g = 2,3) read rows 4 apart: with a row stride of 4*odd floats
// that is a 16-bank shift, i.e. conflict-free ds_read_b32.  A trailing unpaired k-step uses joints 8p+g.
__host__ __device__ constexpr int mix_vmap(int V, int ks, int g) {
    return ks < 2 * (((V + 3) / 4) / 2) ? 8 * (ks >> 1) + 2 * (ks & 1) + 4 * (g & 1) + (g >> 1) : 8 * (((V + 3) / 4) / 2) + g;
}

// ------------------------------------------------------------------------------------------------
// mix: Z[c,q,w] = sum_v ( sum_t X[c,t,v] T[v,t,q] ) A[q,v,w]          (stsgcn.py:154-155)
// The joint mix runs on the matrix cores: for one (chain n, output frame q, block of 16 channels)
//     D[w][c] = sum_v A_q[v][w] * Y[v][c]        A operand = A_q^T fragments (pre-packed, from L2/L1)
//                                                 B operand = Y[v][c], built in registers by the time mix:
//     Y[v][c] = sum_t X[(n,t,v)][c] * T[v][t][q]  lane (j = c, g): v = 4s + g for k-step s  (T LDS reads + T FMAs)
// V is padded to KS*4 rows (zero weights) and 16*MT output joints.  The D fragment (lane: channel j,
// joints 4g..4g+3) is stored to Z[(n,q,w)][c].
// init(n,q,w,c) seeds the accumulator (0, or the residual term of a W-first layer); store(n,q,w,c,val) consumes the
// result (plain Z store, in-place PReLU epilogue of layer 6, or the fused DDPM update of layer 10).
// ------------------------------------------------------------------------------------------------
template <int CIN, int V, int T, int NB>
struct MixCfg {
    static constexpr int KS = (V + 3) / 4;
    static constexpr int KP = 2 * (KS / 2);                     // paired k-steps (see mix_vmap)
    static constexpr int MT = (V + 15) / 16;
    static constexpr int CB = CIN / 16;
    // output frames computed together by one unit: all of them (shared X reads) when that still gives every wave
    // work, otherwise one frame per unit
    static constexpr int QALL = (T % 3 == 0) ? 3 : (T % 2 == 0) ? 2 : 1;
    // the largest chunk (3, 2, 1 frames) that still gives every wave a unit; failing that, the largest one that keeps more
    // than half of them busy in a single round (e.g. 32 channels at 6 frames: 6 two-frame units -- one round, X reads shared
    // by the pair, no mid-stage coefficient fetch -- instead of 12 single-frame units in two rounds)
    static constexpr int Q2 = (T % 2 == 0) ? 2 : 1;
    static constexpr int units_of(int qc) { return NB * CB * (T / qc); }
    // 12 frames, 64 channels: six frames per unit -- one round of 8 units instead of two of 16, every X value read once
    // per half of the output frames (the shape has no register cap)
    static constexpr int Q6 = (T == 12) ? 6 : 1;
    static constexpr int QC = (Q6 > 1 && units_of(Q6) >= NWAVES) ? Q6 : units_of(QALL) >= NWAVES ? QALL : units_of(Q2) >= NWAVES ? Q2
                            : 2 * units_of(QALL) > NWAVES ? QALL : 2 * units_of(Q2) > NWAVES ? Q2 : 1;
    static constexpr int NQ = T / QC;
    static constexpr int UNITS = NB * CB * NQ;                  // one unit = (chain, 16-channel block, frame chunk)
    static constexpr int PER = (UNITS + NWAVES - 1) / NWAVES;   // rounds
    static constexpr int NR = (KS * T + 15) / 16;               // VGPRs holding the time-mix coefficients of one q
    // 12 single-frame units on 8 waves (the 32-channel layers at T = 3, NB = 2): four waves take two units.  SAMEQ gives
    // those waves two units of the SAME output frame -- waves 0-3: frame w/2, groups 2(w&1) + round; waves 4-7: frame 2,
    // group w-4 -- so the coefficients (which depend on the frame only) serve both rounds and nothing is fetched mid-stage
    static constexpr bool SAMEQ = QC == 1 && NQ == 3 && UNITS == 12 && NWAVES == 8;
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

// init functor of a mix whose accumulators start at zero (x + 0.f is not folded away: -0.0)
struct ZeroInit { __device__ __forceinline__ float operator()(int, int, int, int) const { return 0.f; } };
// FORCE (kernels without a register cap): the unit's X reads are pinned in front of its arithmetic (the scheduler otherwise
// sinks each k-step's reads to their first use and the wave pays an LDS round trip per k-step)
template <int CIN, int V, int T, int NB, bool FORCE = false, class Init, class Store>
__device__ __forceinline__ void mix_stage(const float* __restrict__ in, int cs_in, const MixCoef<CIN, V, T, NB>& pre,
                                          const float* __restrict__ tqd, const float* __restrict__ af, int wave, int lane,
                                          Init&& init, Store&& store) {
    using M = MixCfg<CIN, V, T, NB>;
    constexpr int KS = M::KS, KP = M::KP, MT = M::MT, CB = M::CB, QC = M::QC, NQ = M::NQ, PER = M::PER;
    const int j = lane & 15, g = lane >> 4;
    const int voff_pair = 4 * (g & 1) + (g >> 1);
    // the X values of one unit: x[ks][t] = X[(n, t, joint of (ks, lane group))][channel cb*16 + j]
    auto load_x = [&](int u, float (&x)[KS][T]) {
        const int rest = u / NQ;
        const int cb = rest % CB, n = rest / CB;
        const float* xin_p = in + __mul24(n * T * V + voff_pair, cs_in) + cb * 16 + j;       // (small indices: full-rate 24-bit multiply)
        const float* xin_l = in + __mul24(n * T * V + g, cs_in) + cb * 16 + j;
        static_for<KS>([&](auto si) {
            constexpr int ks = decltype(si)::value;
            constexpr int vbase = ks < KP ? 8 * (ks >> 1) + 2 * (ks & 1) : 4 * KP;
            const float* xb = ks < KP ? xin_p : xin_l;
#pragma unroll
            for (int t = 0; t < T; ++t) x[ks][t] = xb[(t * V + vbase) * cs_in];
        });
    };
    auto unit = [&](const MixCoef<CIN, V, T, NB>& cur, int u, const float (&xs)[KS][T]) {
        const int q0 = (u % NQ) * QC, rest = u / NQ;
        const int cb = rest % CB, n = rest / CB;
        // V = 17: output joint 16 would cost a whole second m-tile (15/16 wasted, and the kernel is bound by matrix-pipe
        // time); it is accumulated with plain FMAs instead -- 5 per frame, partial sums over this lane group's joints
        constexpr bool J16 = V == 17;
        constexpr int MTM = J16 ? 1 : MT;            // m-tiles on the matrix cores
        f32x4 acc[QC][MTM];
        float part[QC];
#pragma unroll
        for (int qi = 0; qi < QC; ++qi) {
            part[qi] = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTM; ++mt)
                if constexpr (std::is_invocable_v<Init, int, int, int, int, std::true_type>) {
                    acc[qi][mt] = init(n, q0 + qi, mt * 16 + 4 * g, cb * 16 + j, std::true_type{});   // whole fragment (masks joints >= V)
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        acc[qi][mt][r] = 0.f;
                        if (mt * 16 + 4 * g + r < V) acc[qi][mt][r] = init(n, q0 + qi, mt * 16 + 4 * g + r, cb * 16 + j);
                    }
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
                    acc[qi][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.aop[qi][mt][ks], y[qi], acc[qi][mt], 0, 0, 0);
                if constexpr (J16) part[qi] = fmaf(cur.aop[qi][1][ks], y[qi], part[qi]);
            });
        });
#pragma unroll
        for (int qi = 0; qi < QC; ++qi) {
#pragma unroll
            for (int mt = 0; mt < MTM; ++mt) {
                // a store functor that takes the whole 4-joint fragment can issue all its LDS reads before its first
                // write (row-by-row calls serialise: every write may alias the next row's reads)
                if constexpr (std::is_invocable_v<Store, int, int, int, int, f32x4>) {     // (the functor masks joints >= V)
                    store(n, q0 + qi, mt * 16 + 4 * g, cb * 16 + j, acc[qi][mt]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (mt * 16 + 4 * g + r < V) store(n, q0 + qi, mt * 16 + 4 * g + r, cb * 16 + j, acc[qi][mt][r]);
                }
            }
            if constexpr (J16) {
                // sum the four lane groups' partials (lanes j, j+16, j+32, j+48): two register-swap steps
                const unsigned u = __float_as_uint(part[qi]);
                const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                const unsigned v2 = __float_as_uint(__uint_as_float(h[0]) + __uint_as_float(h[1]));
                const auto f = __builtin_amdgcn_permlane16_swap(v2, v2, false, false);
                const float z16 = __uint_as_float(f[0]) + __uint_as_float(f[1]);
                if constexpr (std::is_same_v<std::decay_t<Init>, ZeroInit>) {
                    if (g == 0) store(n, q0 + qi, 16, cb * 16 + j, z16);
                } else {
                    if (g == 0) store(n, q0 + qi, 16, cb * 16 + j, z16 + init(n, q0 + qi, 16, cb * 16 + j));
                }
            }
        }
    };
    // later rounds: coefficients fetched one round ahead where the register budget allows (T = 3), else in place
    {
        MixCoef<CIN, V, T, NB> cur = pre;
        static_for<PER>([&](auto ri) {
            constexpr int rnd = decltype(ri)::value;
            const int u = M::unit_of(wave, rnd);
            float xs[KS][T];
            load_x(u < 0 ? 0 : u, xs);
            if constexpr (FORCE) __builtin_amdgcn_sched_barrier(0);
            MixCoef<CIN, V, T, NB> nxt;
            if constexpr (rnd + 1 < PER && !M::SAMEQ) nxt.load_unit(tqd, af, M::unit_of(wave, rnd + 1), lane);
            if (u >= 0) unit(cur, u, xs);
            if constexpr (rnd + 1 < PER && !M::SAMEQ) cur = nxt;
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
template <int C, int VIN, int VOUT, int T, int NB, bool CAPTURE, bool ADD, bool ILP = false, int NSK>
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
    static_for<PER>([&](auto pi) {
        constexpr int i = decltype(pi)::value;
        const int u = wave + i * NWAVES;
        if (u < UNITS) {
            const int cb = RC::ALIGNED ? wave % CB : u % CB, nt = RC::ALIGNED ? (wave / CB) * T + i : u / CB;
            const float* xin = in + __mul24(nt * VIN, cs_in) + cb * 16 + j;
            static_for<KS>([&](auto si) {
                constexpr int ks = decltype(si)::value;
                int row;
                if (CAPTURE) row = ks < 4 ? 4 * g + ks : 16 + g;
                else row = ks < KP ? 8 * (ks >> 1) + 2 * (ks & 1) + 4 * (g & 1) + (g >> 1) : 4 * KP + g;
                if constexpr (CAPTURE) skip[i * SK + ks] = xin[row * cs_in];
                else xr[i][ks] = xin[row * cs_in];
            });
        }
    });
    constexpr bool J16 = VOUT == 17;     // output joint 16 on the VALU (partial sums per lane group, permlane-swap reduction)
    constexpr int MTM = J16 ? 1 : MT;   // instead of a second m-tile with one useful row -- same trade as in mix_stage
    // the stores of one unit (+ the skip tensor of the up-samplers)
    auto finish = [&](auto pi, f32x4 (&acc)[MTM], float part) {
        constexpr int i = decltype(pi)::value;
        const int u = wave + i * NWAVES;
        const int cb = RC::ALIGNED ? wave % CB : u % CB, nt = RC::ALIGNED ? (wave / CB) * T + i : u / CB;
        if constexpr (ADD) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[0][r] += skip[i * SK + r];
            if constexpr (SK == 5 && !J16) acc[MT - 1][0] += skip[i * SK + 4];   // joint 16: lane group g = 0, row 0 of m-tile 1
        }
        float* zo = out + __mul24(nt * VOUT + 4 * g, cs_out) + cb * 16 + j;
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
            if (g == 0) out[__mul24(nt * VOUT + 16, cs_out) + cb * 16 + j] = z16;
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
            if (wave + decltype(pi)::value * NWAVES < UNITS) finish(pi, acc[decltype(pi)::value], part[decltype(pi)::value]);
        });
    } else {
        static_for<PER>([&](auto pi) {
            constexpr int i = decltype(pi)::value;
            const int u = wave + i * NWAVES;
            if (u < UNITS) {
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
    const int j = lane & 15, g = lane >> 4;
    const int c0 = mt * 16 + 4 * g;
    // per-lane bases of this wave's FIRST tile, once per call; tile i sits at the compile-time offset i * NG * 16 * stride
    // (a per-tile col * stride is a quarter-rate v_mul_lo_u32 plus two adds on the VALU, which shares the SIMD with the MFMAs)
    const int col0 = ng * 16 + j;
    const float* const p1b = b1 + __mul24(col0, cs1) + 4 * g;
    const float* const p2b = b2 + __mul24(col0, cs2) + 4 * g;
    constexpr int KQ = KQ1 + KQ2;
    constexpr int DEPTH = KQ < 3 ? KQ : 3;
    // one 16x16 output tile: B fragments (one ds_read_b128 = 4 k-steps) fetched DEPTH reads ahead of the MFMAs that consume
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
        float4 buf[DEPTH];
        static_for<DEPTH>([&](auto dd) { buf[decltype(dd)::value] = rd(dd); });
        float4 pe = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (HASPRE) pe = pre(ii, col, ng);
        if constexpr (FORCE) __builtin_amdgcn_sched_barrier(0);
        static_for<KQ>([&](auto kk) {
            constexpr int kq = decltype(kk)::value;
            const float4 u = buf[kq % DEPTH];
            if constexpr (kq + DEPTH < KQ) buf[kq % DEPTH] = rd(std::integral_constant<int, kq + DEPTH>{});
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
        float4 buf[2][DEPTH];
        static_for<DEPTH>([&](auto dd) { buf[0][decltype(dd)::value] = rd(0, dd); buf[1][decltype(dd)::value] = rd(1, dd); });
        float4 pe0 = make_float4(0.f, 0.f, 0.f, 0.f), pe1 = pe0;
        if constexpr (HASPRE) { pe0 = pre(ia, col0 + i0 * NG * 16, ng); pe1 = pre(ib, col0 + i1 * NG * 16, ng); }
        __builtin_amdgcn_sched_barrier(0);
        static_for<KQ>([&](auto kk) {
            constexpr int kq = decltype(kk)::value;
            const float4 u0 = buf[0][kq % DEPTH], u1 = buf[1][kq % DEPTH];
            if constexpr (kq + DEPTH < KQ) {
                buf[0][kq % DEPTH] = rd(0, std::integral_constant<int, kq + DEPTH>{});
                buf[1][kq % DEPTH] = rd(1, std::integral_constant<int, kq + DEPTH>{});
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
    if constexpr (DUAL && MAXN >= 2) {
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

// ------------------------------------------------------------------------------------------------
// OPT-IN (mcd_set_option(MCD_OPT_BF16X3, 1), not the measured default): the same channel GEMM on the bf16 matrix path with both operands
// split into bf16 pairs, x = hi + lo, and hi*hi + hi*lo + lo*hi accumulated in fp32 -- three v_mfma_f32_16x16x32_bf16
// per K = 32 instead of eight v_mfma_f32_16x16x4_f32 (which run at the vector-FP32 rate and hold the SIMD's FP32 lanes).
// Scores stay within ~1e-6 of the fp32 path (tests/studies/bf16x3_error.py).  Weights are split at pack time
// ([m-tile][K/32][hi, lo][lane] x 8 bf16: lane (row, g) holds k = 32 ch + 8 g + i); activations are split here, as they
// are read (lane (col j, g): channels 32 ch + 8 g + i, two ds_read_b128).
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int MT, int CH>
__device__ __forceinline__ void load_afrags_bf3(const float4* __restrict__ wp, int wave, int lane, float4 (&a)[2 * CH], int mi = 0) {
    const float* wpl = reinterpret_cast<const float*>(wp + (((wave + mi * NWAVES) % MT) * CH * 2) * 64 + lane);
#pragma unroll
    for (int i = 0; i < 2 * CH; ++i) a[i] = load_global4(wpl + i * 256);
}
__device__ __forceinline__ bf16x8 as_bf16x8(const float4& f) { return __builtin_bit_cast(bf16x8, f); }
// A tensor that only a split-bf16 GEMM reads (a layer's mix output z, layer 5's output) is stored already split, as two
// bf16 planes inside its fp32-sized row: channels' hi halves in bytes [0, 2C), their lo halves in [2C, 4C).  The producer
// splits each element once (hi = the element truncated to bf16 -- its upper 16 bits, stored as they are; lo = bf16 of the
// exact remainder); the GEMM's m-tiles (up to 8 per element) read ready-made fragments with one ds_read_b128 per plane.
__device__ __forceinline__ unsigned short bf16_lo_bits(float x) {
    const __bf16 lo = (__bf16)(x - __uint_as_float(__float_as_uint(x) & 0xffff0000u));
    return __builtin_bit_cast(unsigned short, lo);
}
template <int C>
__device__ __forceinline__ void store_split(float* row, int c, float x) {      // one element
    unsigned short* h = reinterpret_cast<unsigned short*>(row) + c;
    h[0] = (unsigned short)(__float_as_uint(x) >> 16);
    h[C] = bf16_lo_bits(x);
}
template <int C>
__device__ __forceinline__ void store_split4(float* row, int c0, float x0, float x1, float x2, float x3) {   // 4 channels
    unsigned short* h = reinterpret_cast<unsigned short*>(row) + c0;
    const uint2 hv = make_uint2(__builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), 0x07060302u),
                                __builtin_amdgcn_perm(__float_as_uint(x3), __float_as_uint(x2), 0x07060302u));
    const uint2 lv = make_uint2((unsigned)bf16_lo_bits(x0) | ((unsigned)bf16_lo_bits(x1) << 16),
                                (unsigned)bf16_lo_bits(x2) | ((unsigned)bf16_lo_bits(x3) << 16));
    *reinterpret_cast<uint2*>(h) = hv;
    *reinterpret_cast<uint2*>(h + C) = lv;
}
template <int MT, int NT, int CH1, int CH2, bool IDRES = false, bool P1 = false, bool P2 = false, class Epi>
__device__ __forceinline__ void gemm_tiles_bf3(const float4 (&a)[2 * (CH1 + CH2)], const float* __restrict__ b1, int cs1,
                                               const float* __restrict__ b2, int cs2, int wave, int lane, Epi&& epi, int mi = 0) {
    constexpr int NG = Tiling<MT, NT>::NG;
    constexpr int MAXN = Tiling<MT, NT>::MAXN;
    const int mt = (wave + mi * NWAVES) % MT, ng = MT > NWAVES ? 0 : wave / MT;
    const int j = lane & 15, g = lane >> 4;
    const int c0 = mt * 16 + 4 * g;
    static_for<MAXN>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        const int nt = ng + i * NG;
        if (nt < NT) {
            const int col = nt * 16 + j;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            if (IDRES) {
                const float4 r = *reinterpret_cast<const float4*>(b2 + col * cs2 + c0);
                c[0] = r.x; c[1] = r.y; c[2] = r.z; c[3] = r.w;
            }
            const float* p1 = b1 + col * cs1 + 8 * g;
            const float* p2 = b2 + col * cs2 + 8 * g;
            static_for<CH1 + CH2>([&](auto cc) {
                constexpr int ch = decltype(cc)::value;
                bf16x8 hi, lo;
                if constexpr (ch < CH1 ? P1 : P2) {
                    // stored split (see store_split): the two planes of this buffer's row, 8 channels each
                    constexpr int C = (ch < CH1 ? CH1 : CH2) * 32, cl = (ch < CH1 ? ch : ch - CH1) * 32;
                    const unsigned short* rp = reinterpret_cast<const unsigned short*>(ch < CH1 ? b1 + col * cs1 : b2 + col * cs2) + cl + 8 * g;
                    hi = *reinterpret_cast<const bf16x8*>(rp);
                    lo = *reinterpret_cast<const bf16x8*>(rp + C);
                } else {
                    const float* p = ch < CH1 ? p1 + ch * 32 : p2 + (ch - CH1) * 32;
                    const float4 x0 = *reinterpret_cast<const float4*>(p), x1 = *reinterpret_cast<const float4*>(p + 4);
                    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                    unsigned h[4];
#pragma unroll
                    for (int e = 0; e < 8; ++e) lo[e] = (__bf16)(x[e] - __uint_as_float(__float_as_uint(x[e]) & 0xffff0000u));
#pragma unroll
                    for (int e = 0; e < 4; ++e)      // truncated hi halves of an element pair, one v_perm_b32 (see store_split)
                        h[e] = __builtin_amdgcn_perm(__float_as_uint(x[2 * e + 1]), __float_as_uint(x[2 * e]), 0x07060302u);
                    hi = __builtin_bit_cast(bf16x8, make_uint4(h[0], h[1], h[2], h[3]));
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a[2 * ch]), hi, c, 0, 0, 0);       // hi_w * hi_x
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a[2 * ch]), lo, c, 0, 0, 0);       // hi_w * lo_x
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(a[2 * ch + 1]), hi, c, 0, 0, 0);   // lo_w * hi_x
            });
            epi(ii, col, c0, c, ng * 16 + j, ng);
        }
    });
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
template <int CIN, int COUT, int V, bool RES, bool HASEMB, int T, int NB, bool FORCE = false, int CSX = cs_of(CIN), bool BF3 = false, bool OUTP = false, class H1, class H2>
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
    static_assert(!BF3 || KQ1 % 2 == 0, "split-bf16 path: K a multiple of 32 per operand buffer");
    const int trs = 8 + 8 * ((prof_id - 32) / 3);      // trace slots of this layer (profile builds)
    prof.trace(trs + 0);
    const float* bias = wb + lw.bias;
    float4 bcur;
    if (pre_afr != nullptr && !BF3) {
#pragma unroll
        for (int k = 0; k < KQ1 + KQ2; ++k) afr[k] = pre_afr->a[k];
        bcur = pre_afr->bcur;
    } else {
        if constexpr (BF3) load_afrags_bf3<MT, (KQ1 + KQ2) / 2>(reinterpret_cast<const float4*>(wb + lw.wpb), wave, lane, afr);
        else load_afrags<MT, KQ1 + KQ2>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr);
        // (the folded bias starts the first tile's accumulators: fetched here, with the weight fragments, so that its L2
        // latency hides behind the mix as well)
        bcur = load_global4(bias + (wave % MT) * 16 + 4 * (lane >> 4));
    }
    mix_stage<CIN, V, T, NB, FORCE>(in, CSX, mc, wb + lw.tq, wb + lw.am, wave, lane,
                             ZeroInit{},
                             [&](int n, int q, int w0, int c, auto v) {
                                 // one LDS address per 4-joint fragment, the rows at constant offsets from it (row by row the
                                 // compiler recomputes (.. + w) * CSI for every element: 2-3 VALU instructions per store)
                                 float* zp = z + __mul24((n * T + q) * V + w0, CSI) + c;
                                 if constexpr (std::is_same_v<decltype(v), f32x4>) {
#pragma unroll
                                     for (int r = 0; r < 4; ++r)
                                         if (w0 + r < V) {
                                             if constexpr (BF3) store_split<CIN>(zp - c + r * CSI, c, v[r]);    // (z feeds the GEMM only)
                                             else zp[r * CSI] = v[r];
                                         }
                                 } else {
                                     if constexpr (BF3) store_split<CIN>(zp - c, c, v);
                                     else *zp = v;
                                 }
                             });
    prof.trace(trs + 1);
    // The NEXT stage's coefficient loads.  The vector-memory path accepts ~1 wave-wide load per 10 cycles and all eight waves
    // issue 10 .. 40 of them at the same point of the stage: in front of the GEMM tiles (where they used to be) the last
    // wave's first MFMA waited ~2 k cycles for its loads to be accepted (profiles/r03c_seq24_trace.txt).  They are issued
    // HERE instead, behind the wave's mix and in front of the barrier: the older wave of each SIMD reaches this point
    // 1 - 2 k cycles before the younger one and would only wait (12 frames +3 % together with EARLY2, 6 frames +1.9 %,
    // 3 frames +1.1 %: profiles/r03k_prebar_ab.txt).
    constexpr bool PREBAR = !BF3;
    if constexpr (PREBAR) pre_gemm();
    bsync();
    prof.trace(trs + 2);
    prof.mark(prof_id);
    if constexpr (!PREBAR) pre_gemm();
    const float slope = lw.slope;
    const float pinf = prelu_bound(slope);     // see prelu()
    constexpr int TILE_STEP = Tiling<MT, NT>::NG * 16;      // columns between a wave's consecutive n-tiles
    // the folded bias starts the accumulators of the layers with a residual convolution (identity residuals start from X)
    constexpr bool FOLD = RES && !BF3;
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
        // (the split-bf16 path keeps the check: its pad rows hold bf16 planes of stale fp32 data, which can read as NaN, and a
        // NaN written to a pad row would meet the zero weights of the resamplers' K padding: NaN x 0)
        if ((COUT % 16 == 0 || c0 < COUT) && (!BF3 || col < COLS)) {
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
            if constexpr (OUTP) store_split4<COUT>(out + __mul24(col, CSO), c0, r0[0], r0[1], r1[0], r1[1]);
            else lds_store4(oaddr + tile_bytes, r0[0], r0[1], r1[0], r1[1]);
        }
    };
    prof.trace(trs + 3);
    if constexpr (BF3) gemm_tiles_bf3<MT, NT, KQ1 / 2, KQ2 / 2, !RES, true, false>(afr, z, CSI, in, CSX, wave, lane, epi);
    else if constexpr (EPRE) gemm_tiles<MT, NT, KQ1, KQ2, !RES, FORCE>(afr, z, CSI, in, CSX, wave, lane, epi, 0, FOLD ? bcur : make_float4(0.f, 0.f, 0.f, 0.f), emb_of);
    else gemm_tiles<MT, NT, KQ1, KQ2, !RES, FORCE>(afr, z, CSI, in, CSX, wave, lane, epi, 0, FOLD ? bcur : make_float4(0.f, 0.f, 0.f, 0.f));
#pragma unroll
    for (int mi = 1; mi < Tiling<MT, NT>::MW; ++mi) {     // workgroups with fewer waves than m-tiles: next m-tile(s)
        if constexpr (BF3) load_afrags_bf3<MT, (KQ1 + KQ2) / 2>(reinterpret_cast<const float4*>(wb + lw.wpb), wave, lane, afr, mi);
        else load_afrags<MT, KQ1 + KQ2>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr, mi);
        bcur = load_global4(bias + ((wave + mi * NWAVES) % MT) * 16 + 4 * (lane >> 4));
        set_bases(mi);
        if constexpr (EHOIST) e_pre = lds_load4(eaddr);
        if constexpr (BF3) gemm_tiles_bf3<MT, NT, KQ1 / 2, KQ2 / 2, !RES, true, false>(afr, z, CSI, in, CSX, wave, lane, epi, mi);
        else if constexpr (EPRE) gemm_tiles<MT, NT, KQ1, KQ2, !RES, FORCE>(afr, z, CSI, in, CSX, wave, lane, epi, mi, FOLD ? bcur : make_float4(0.f, 0.f, 0.f, 0.f), emb_of);
        else gemm_tiles<MT, NT, KQ1, KQ2, !RES, FORCE>(afr, z, CSI, in, CSX, wave, lane, epi, mi, FOLD ? bcur : make_float4(0.f, 0.f, 0.f, 0.f));
    }
    prof.trace(trs + 4);
    pre_barrier();
#ifdef MCD_PROFILE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (stamp 5 = this wave's stores have landed)
#endif
    prof.trace(trs + 5);
    bsync();
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
template <int L, int T, int NB, bool FORCE = false, int CSX = cs_of(layer_desc(L).cin), bool BF3 = false, bool OUTP = false, class H1, class H2>
__device__ __forceinline__ void layer_std(const float* wb, const LMix<L, T, NB>& mc, const float* in, float* z, float* out,
                                          const float* emb, int wave, int lane, Prof& prof, H1&& pre_gemm, H2&& pre_barrier,
                                          const LAfr<L>* pre_afr = nullptr) {
    constexpr LDesc D = layer_desc(L);
    layer_generic<D.cin, D.cout, D.V, D.res != 0, true, T, NB, FORCE, CSX, BF3, OUTP>(wb, layer_w(wb, L), mc, in, z, out, emb + emb_off(L), wave, lane,
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
    static constexpr int BIA = 64 + 16;         // biases of the two W-first layers (6: 64, 10: 2), read inside their store functors
    static constexpr int UPD = 16;              // per (chain, U-Net frame): first column of the frame its prediction updates, or -1
    static constexpr int ZO = P17 * 2;          // layer 10's mixed output Z[col][c] between its mix and the element-wise tail
    static constexpr int TT = P17 * 2;          // per (column, coordinate) of the element-wise tail: packed (chain, frame, joint) indices
    static constexpr int CE = 4 * EDIM;         // condition embeddings of the workgroup's windows [NB <= 4][16]
    static constexpr int LOSS = NB * 64;        // per-sample losses of the workgroup's windows [NB][S <= 64] (in-kernel aggregation)
    static constexpr int EXW = EMB_EXTRA * 20;  // embedding rows beyond the first NTHREADS: [row][16 weights, bias, pad]
#ifdef MCD_PROFILE
    static constexpr int PROFTR = NB * T >= 10 ? PROF_TRACE * PROF_NW : 0;      // time stamps: the one-workgroup-per-CU shapes have the room
    static constexpr int PROF = PROF_SLOTS + PROFTR;
#else
    static constexpr int PROF = 0;
#endif
    static constexpr int TOTAL = R + XT + EMB + EAUX + ZN + WM + BIA + UPD + ZO + TT + CE + LOSS + EXW + PROF;
    static constexpr size_t BYTES = (size_t)TOTAL * 4;
};

// ------------------------------------------------------------------------------------------------
// condition encoder, fast path for the shipped architecture (channels [32,16,32] + h_dim 32, latent 16):
// the same MFMA mix / GEMM stages as the U-Net, NB windows per 512-thread workgroup, followed by the
// bottleneck Linear over the (c,t,v) flattening (stsae.py:73-89).  Reads the condition frames straight from the
// window tensor (no gather pass).  Other channel lists use cond_encode_kernel below.
// ------------------------------------------------------------------------------------------------
constexpr int TABC = 128;                  // cond table: second 128 words of the weight buffer
constexpr int TABC_LW = 40, TABC_LB = 41;  // bottleneck Linear weight [16][32*T*17] / bias
struct FrameIdx { int idx[MCD_MAX_FRAMES]; };

// body shared by cond_fast_kernel and by the trajectory kernel's prologue (P.cond_inkernel): windows b0 .. b0 + NB - 1,
// frame_of(t) = data frame of condition frame t; the embeddings go to emb_lds[n][16] (LDS) and / or emb_out (B,16).
// smem: P17 * (2 * 20 + 2 * 36) floats, zeroed by the caller.
template <int T, int NB, class FrameOf>
__device__ __forceinline__ void cond_fast_body(const float* wbuf, const DataView& dv, FrameOf&& frame_of, int seg_len, float* smem,
                                               int b0, int B, float* emb_lds, float* __restrict__ emb_out) {
    constexpr int P17 = ceil16(NB * T * 17);
    constexpr int s16 = P17 * 20, s32 = P17 * 36;
    constexpr int TV = T * 17, COLS = NB * TV;
    float* const X0 = smem;                   // [P17][20]  in of layers 0, 2 ; out of layer 1
    float* const Z0 = smem + s16;             // [P17][20]
    float* const Y0 = smem + 2 * s16;         // [P17][36]  out of layers 0, 2 ; in of layers 1, 3
    float* const Z1 = smem + 2 * s16 + s32;   // [P17][36]
    float* const H = smem;                    // [P17][36]  out of layer 3 (over X0/Z0: 36 <= 40)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Prof prof;
    prof.off();
    for (int u = tid; u < COLS * C0; u += NTHREADS) {
        const int c = u % C0, col = u / C0;
        const int n = col / TV, t = (col / 17) % T, v = col % 17;
        const int b = b0 + n < B ? b0 + n : B - 1;
        X0[col * 20 + c] = load_coord(dv, b, c, frame_of(t), v, seg_len);
    }
    bsync();
    const float* wb = wbuf;
    auto lw = [&](int l) {
        LayerW w;
        w.tq = tab_i(wb, TABC + l * F_STRIDE + F_TQ); w.am = tab_i(wb, TABC + l * F_STRIDE + F_AM);
        w.wp = tab_i(wb, TABC + l * F_STRIDE + F_WP); w.bias = tab_i(wb, TABC + l * F_STRIDE + F_BIAS);
        w.slope = tab_f(wb, TABC + l * F_STRIDE + F_SLOPE);
        return w;
    };
    layer_generic<16, 32, 17, true, false, T, NB>(wb, lw(0), X0, Z0, Y0, nullptr, wave, lane, prof, 0);   // 2(16) -> 32
    layer_generic<32, 16, 17, true, false, T, NB>(wb, lw(1), Y0, Z1, X0, nullptr, wave, lane, prof, 0);   // 32 -> 16
    layer_generic<16, 32, 17, true, false, T, NB>(wb, lw(2), X0, Z0, Y0, nullptr, wave, lane, prof, 0);   // 16 -> 32
    layer_generic<32, 32, 17, false, false, T, NB>(wb, lw(3), Y0, Z1, H, nullptr, wave, lane, prof, 0);   // 32 -> 32
    // bottleneck Linear: emb[n][j] = b[j] + sum_k W[j][k] H[n][k], k = c*TV + tv.  thread = (n, j, part of 16)
    constexpr int F = 32 * TV;
    gfloat* W = as_global(wb + tab_i(wb, TABC + TABC_LW));
    gfloat* bb = as_global(wb + tab_i(wb, TABC + TABC_LB));
    for (int u = tid; u < NB * EDIM * 16; u += NTHREADS) {
        const int part = u & 15, jo = (u >> 4) % EDIM, n = u / (16 * EDIM);
        // (c, tv) loops instead of k % TV, k / TV per element; the 16 parts of an output are the 16 lanes of a DPP row.
        // Compile-time trip counts (the ragged last 16-block is predicated): the loops unroll and the weight loads of several
        // channels are in flight together -- with the data-dependent bound `tv + part < TV` every load waited for the FMA
        // before it (one L2 round trip per element: 100 .. 400 of them per thread, the whole encoder's time)
        constexpr int NT16 = (TV + 15) / 16;
        float a = 0.f;
        gfloat* wr = W + jo * F + part;
        const float* hr = H + (n * TV + part) * 36;
#pragma unroll 4
        for (int c = 0; c < 32; ++c) {
            float wv[NT16];
#pragma unroll
            for (int i = 0; i < NT16; ++i) wv[i] = (i * 16 + part < TV) ? wr[c * TV + i * 16] : 0.f;
#pragma unroll
            for (int i = 0; i < NT16; ++i) a = fmaf(wv[i], (i * 16 + part < TV) ? hr[i * 16 * 36 + c] : 0.f, a);
        }
        a = row16_sum(a);
        if (part == 0) {
            const float e = a + bb[jo];
            if (emb_lds) emb_lds[n * EDIM + jo] = e;
            if (emb_out && b0 + n < B) emb_out[(size_t)(b0 + n) * EDIM + jo] = e;
        }
    }
}

template <int T, int NB>
__global__ __launch_bounds__(NTHREADS, 2) void cond_fast_kernel(const float* wbuf, const DataView dv, const FrameIdx fi,
                                                                int seg_len, float* __restrict__ emb_out, int B) {
    constexpr int P17 = ceil16(NB * T * 17);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    for (int u = threadIdx.x; u < P17 * (2 * 20 + 2 * 36); u += NTHREADS) smem[u] = 0.f;
    __syncthreads();
    cond_fast_body<T, NB>(wbuf, dv, [&](int t) { return fi.idx[t]; }, seg_len, smem, blockIdx.x * NB, B, nullptr, emb_out);
}

// ------------------------------------------------------------------------------------------------
// The persistent scoring kernel.  mode 0: full reverse-diffusion trajectories + loss (mcd_score);
// mode 1: one eps-prediction pass (mcd_unet_forward).
// ------------------------------------------------------------------------------------------------
// LT = true (layer test, mcd_layer_forward): a single-pass run in which stage P.lt_stage's input region is overwritten with
// P.lt_in right before the stage and its output region is copied to P.lt_out right after it -- the stage functions and the
// LDS plan under test are the production ones; the production instantiations (LT = false) contain none of this.
template <int T, int NB, int MINW, bool BF3 = false, bool LT = false>
__global__ __launch_bounds__(NTHREADS, MINW) void score_kernel(const ScoreParams P) {
    using PL = Plan<T, NB>;
    constexpr int TV17 = T * 17;
    constexpr int COLS17 = NB * TV17;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const RG = smem;                 // work region (see Plan)
    float* const XT = RG + PL::R;
    float* const EMB = XT + PL::XT;
    float* const E10 = EMB + PL::EMB;
    float* const SEN = E10 + 32;
    float* const ZN = E10 + PL::EAUX;
    int* const WM = reinterpret_cast<int*>(ZN + PL::ZN);
    float* const BIA = reinterpret_cast<float*>(WM + PL::WM);
    int* const UPD = reinterpret_cast<int*>(BIA + PL::BIA);
    float* const ZO = reinterpret_cast<float*>(UPD + PL::UPD);

    const int tid0 = threadIdx.x;
    int tid = tid0;
    int lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Prof prof;
    prof.off();
#ifdef MCD_PROFILE
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();     // (100 MHz constant clock, the same on every CU)
#endif
    if (P.phase > 0 && blockIdx.x * 2 >= gridDim.x) {
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)P.phase * 1024ull) __builtin_amdgcn_s_sleep(32);
    }
    if (P.phase < 0) {      // tuning experiment: every workgroup starts at its own (hashed) offset of 0 .. 63 x |phase| x 16 cycles
        const unsigned h = (blockIdx.x * 2654435761u) >> 26;
        const unsigned long long t0 = __builtin_readcyclecounter();
        while (__builtin_readcyclecounter() - t0 < (unsigned long long)h * (unsigned long long)(-P.phase) * 16ull) __builtin_amdgcn_s_sleep(32);
    }
    // this workgroup: windows win0 .. win0 + NB - 1, samples part, part + split, ...
    const int grp = blockIdx.x / P.split, part = blockIdx.x - grp * P.split;
    const int win0 = grp * NB;
    const int Tx = P.n_corrupt;
    auto window_of = [&](int n) { const int b = win0 + n; return b < P.B ? b : P.B - 1; };     // (clamped: empty slots recompute the last window)
    if (threadIdx.x < NB) {
        const int b = window_of(threadIdx.x);
        WM[threadIdx.x] = P.win_mask ? P.win_mask[b] : P.fixed_mask;
        WM[4 + threadIdx.x] = b;
    }
    // biases the W-first layers add inside their mix store functors: from LDS there, not from global memory (a global
    // load in a functor that also stores to LDS is re-issued per call: one L2 round trip per output row)
    if (threadIdx.x >= 128 && threadIdx.x < 128 + NB * T) {
        // the prediction at frame t drives corrupt frame k = upd_of[t], which lives at frame pos_of[k] (the same frame except
        // for 'concat' with the condition at the END of the window, where the reference reads the prediction at the corrupt
        // frames' ORIGINAL indices, mocodad.py:829-838); with per-window frame sets (random_imp) every clear bit updates itself
        const int i = threadIdx.x - 128, n = i / T, t = i % T;
        const int fixed = P.win_mask ? P.win_mask[window_of(n)] : P.fixed_mask;
        const int k = P.win_mask ? (((fixed >> t) & 1) ? -1 : 0) : P.upd_of[t];
        UPD[i] = k < 0 ? -1 : (n * T + (P.win_mask ? t : P.pos_of[k])) * 17;
    }
    // the tail's thread -> (chain n, frame t, joint v) map, packed: n*T+t | n << 4 | t << 6 | v << 10 (the divisions by 17 and
    // T*17 cost ~35 VALU instructions per thread and pass when done in place)
    static_assert(NB <= 4 && T <= 16 && NB * T <= 16, "tail index packing: 4 bits (chain, frame) | 2 bits chain | 4 bits frame");
    int* const TT = reinterpret_cast<int*>(ZO + PL::ZO);
    float* const CE = reinterpret_cast<float*>(TT + PL::TT);
    float* const LOSSB = CE + PL::CE;
    float* const EXW = LOSSB + PL::LOSS;
    for (int u = threadIdx.x; u < EMB_EXTRA * 17; u += NTHREADS) {
        const int r = u / 17, k = u % 17;
        EXW[r * 20 + k] = k < 16 ? P.wbuf[tab_i(P.wbuf, TAB_WE) + (NTHREADS + r) * EDIM + k] : P.wbuf[tab_i(P.wbuf, TAB_BE) + NTHREADS + r];
    }
    for (int u = threadIdx.x; u < COLS17 * C0; u += NTHREADS) {
        const int col = u / C0, n = col / TV17, t = (col / 17) % T, v = col % 17;
        TT[u] = (n * T + t) | (n << 4) | (t << 6) | (v << 10);
    }
    if (threadIdx.x < 64) BIA[threadIdx.x] = P.wbuf[tab_i(P.wbuf, 6 * F_STRIDE + F_BIAS) + threadIdx.x];
    else if (threadIdx.x < 64 + C0) BIA[threadIdx.x] = P.wbuf[tab_i(P.wbuf, 10 * F_STRIDE + F_BIAS) + threadIdx.x - 64];
    const int CTV = C0 * Tx * 17;          // elements of one generated pose
    const int K = P.ns > 2 ? P.ns - 1 : 1;  // noise slots per sample

    // ---- condition embeddings of the workgroup's windows -> CE[n][16]: computed right here with the condition encoder's
    //      MFMA stages (the shipped architecture at T condition frames), or read from the caller's (B,16) tensor
    if (P.cond_inkernel) {
        for (int u = tid; u < PL::R; u += NTHREADS) smem[u] = 0.f;
        bsync();
        cond_fast_body<T, NB>(P.wbuf, P.dv, [&](int t) { return P.cond_idx[t]; }, P.seg_len, smem, win0, P.B, CE, nullptr);
        bsync();
    } else if (threadIdx.x < NB * EDIM) {
        CE[threadIdx.x] = P.cond_emb ? P.cond_emb[(size_t)window_of(threadIdx.x / EDIM) * EDIM + threadIdx.x % EDIM] : 0.f;
    }
    // zero the whole activation area once: pad columns / pad channels must hold finite values
    for (int u = tid; u < PL::R + PL::XT; u += NTHREADS) smem[u] = 0.f;
    bsync();

#ifdef MCD_PROFILE
    prof.acc = reinterpret_cast<unsigned*>(EXW + PL::EXW);
    for (int i = tid0; i < PROF_SLOTS; i += NTHREADS) prof.acc[i] = 0u;     // a barrier follows before the first mark
    prof.on = (tid0 == 0 && blockIdx.x == 0 && P.prof != nullptr); prof.tlast = __builtin_readcyclecounter();
    prof.won = ((tid0 & 63) == 0 && blockIdx.x == 0 && P.prof != nullptr); prof.wv = tid0 >> 6;
    prof.tr = (P.prof && PL::PROFTR) ? prof.acc + PROF_SLOTS : nullptr;
    for (int i = tid0; i < PL::PROFTR; i += NTHREADS) prof.acc[PROF_SLOTS + i] = 0u;
    __syncthreads();
#endif
    // layer test: (B,C,T,V) global tensor <-> LDS region [col = (n,t,v)][channel]
    auto lt_inject = [&](int id, float* region, int cs, int C, int V) {
        if constexpr (LT) {
            if (P.lt_stage == id) {
                bsync();
                for (int u = threadIdx.x; u < NB * C * T * V; u += NTHREADS) {
                    const int v = u % V, t = (u / V) % T, c = (u / (V * T)) % C, n = u / (V * T * C);
                    const int b = window_of(n);
                    region[((n * T + t) * V + v) * cs + c] = P.lt_in[(((size_t)b * C + c) * T + t) * V + v];
                }
                bsync();
            }
        }
    };
    auto lt_dump = [&](int id, const float* region, int cs, int C, int V) {
        if constexpr (LT) {
            if (P.lt_stage == id) {
                bsync();
                for (int u = threadIdx.x; u < NB * C * T * V; u += NTHREADS) {
                    const int v = u % V, t = (u / V) % T, c = (u / (V * T)) % C, n = u / (V * T * C);
                    const int b = win0 + n;
                    if (b < P.B) P.lt_out[(((size_t)b * C + c) * T + t) * V + v] = region[((n * T + t) * V + v) * cs + c];
                }
                bsync();
            }
        }
    };
    // U-Net skip tensors d1 / d2, register-resident between the down- and the up-samplers
    using RS1 = RsCfg<32, 17, 12, T, NB, true>;
    using RS2 = RsCfg<64, 12, 10, T, NB, true>;
    float skip1[RS1::PER * RS1::SK];
    float skip2[RS2::PER * RS2::SK];
    // 6 frames under the 128-VGPR cap (two workgroups per CU): d2 does not fit beside the 128-channel layers' fragments and
    // the register allocator spilled it where it was read (before its own use) and reloaded it in front of every consumer.
    // Parked in private memory by hand instead -- stored after down2 has used it, fetched back in front of the barrier
    // that precedes up3, whose MFMAs run before the skip values are added: the round trip is off the critical path.
#ifndef MCD_STASH
#define MCD_STASH 1
#endif
    constexpr bool STASH2 = !LT && MINW >= 4 && ((T == 6 && (MCD_STASH & 1)) || (T == 3 && (MCD_STASH & 4)));
    constexpr bool STASH1 = !LT && MINW >= 4 && ((T == 6 && (MCD_STASH & 2)) || (T == 3 && (MCD_STASH & 8)));
    float stash1_mem[STASH1 ? RS1::PER * RS1::SK : 1];
    float stash2_mem[STASH2 ? RS2::PER * RS2::SK : 1];
    typedef float __attribute__((address_space(5))) priv_float;         // (explicit private address space: scratch_*, not flat_*)

    const int i_first = P.mode == 1 ? P.step_single : P.ns - 1;
    const int i_last = P.mode == 1 ? P.step_single : 1;
    // embeddings of the first pass; those of pass i-1 are computed during the last layer of pass i
    auto silu_row = [&](int step, int t_id) {      // SEN[n][k] = SiLU(pe(step)[k] + cond[window of chain n][k])
        if (t_id >= 0 && t_id < NB * EDIM) {
            const float e = P.step_table[step * (4 + EDIM) + 4 + (t_id % EDIM)] + CE[t_id];
            SEN[t_id] = e / (1.f + expf(-e));
        }
    };
    // ================= the samples of this workgroup's windows, one trajectory after the other =================
    for (int s = part; s < P.S; s += P.split) {
    // The parameters the per-sample prologue / epilogue need are read through a pointer to the kernarg segment that is
    // opaque per sample, and the thread id likewise: otherwise their (loop-invariant) scalar loads and per-lane addresses are
    // hoisted above this loop and stay live -- in SGPRs / VGPRs the step loop has none to spare of -- across every pass.
    typedef const ScoreParams __attribute__((address_space(4))) KScoreParams;
    KScoreParams* Q = (KScoreParams*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(Q));
    int tid_s = tid0;
    asm volatile("" : "+v"(tid_s));
    // ---- x_T (or the given x in single-pass mode) -> XT[col] = (x0, x1, z0, z1)
    {
        DataView dv;
        dv.data = Q->dv.data; dv.base = Q->dv.base; dv.sc = Q->dv.sc; dv.st = Q->dv.st; dv.trans = Q->dv.trans; dv.aff = Q->dv.aff;
        const float* noise = Q->noise;
        const float* x_in = Q->x_in;
        const int mode = Q->mode, seg_len = Q->seg_len, Bq = Q->B;
        const unsigned long long seed = Q->seed;
        const long long first_window = Q->first_window;
        for (int u = tid_s; u < COLS17; u += NTHREADS) {
            const int n = u / TV17, t = (u / 17) % T, v = u % 17;
            const int b = WM[4 + n];
            const int fixed = WM[n];
            float xv[C0];
#pragma unroll
            for (int c = 0; c < C0; ++c) {
                if (mode == 1) {
                    xv[c] = x_in ? x_in[((b * C0 + c) * T + t) * 17 + v] : 0.f;
                } else if ((fixed >> t) & 1) {
                    xv[c] = load_coord(dv, b, c, fm_src(P, t), v, seg_len);
                } else {
                    const int e = (c * Tx + fm_tx(P, fixed, t)) * 17 + v;
                    if (noise) xv[c] = noise[((size_t)(s * K + 0) * Bq + b) * CTV + e];
                    else xv[c] = philox_normal(seed, (unsigned)e, 0u, (unsigned)s, (unsigned)(first_window + b));
                }
            }
            XT[u * 4 + 0] = xv[0];
            XT[u * 4 + 1] = xv[1];
        }
    }
    bsync();
    {
        EmbRow er;
        er.load(P.wbuf, tid_s);
        silu_row(i_first, tid_s);
        bsync();
        emb_compute<NB>(er, EXW, SEN, EMB, E10 + (i_first & 1) * 16, tid_s);
        bsync();
    }
    LMix<0, T, NB> mc0;                              // layer 0's mix coefficients: fetched one stage ahead like all the others,
    mc0.load(P.wbuf + tab_i(P.wbuf, F_TQ), P.wbuf + tab_i(P.wbuf, F_AM), wave, lane);   // i.e. in the last stage of the previous pass
    // WEARLY: every layer's GEMM weight fragments are fetched at the end of the stage in front of the layer (see LayerAfr).
    // 3 / 4 frames only: +0.5 % there, nothing at 6 frames, -0.9 % at 12 (profiles/r03l_wearly_ab.txt) -- the larger shapes'
    // GEMM stages end with the other prefetches (EARLY2) already
    constexpr bool WEARLY = !BF3 && T <= 4;
    LAfr<0> A0;
    if constexpr (WEARLY) load_lafr<0>(A0, P.wbuf, wave, lane);
    for (int sidx = i_first; sidx >= i_last; --sidx) {
        const float* srow = P.step_table + sidx * (4 + EDIM);
        const float* wb = P.wbuf;
        asm volatile("" : "+s"(wb));   // opaque per step: offset-table loads stay inside the loop
        // same for the thread id: otherwise every per-lane LDS address of every stage is hoisted out of the
        // step loop (loop-invariant) and the ~250 resulting VGPRs are spilled to scratch
        tid = tid0;
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        if constexpr (MINW >= 4) {
            // Two workgroups share a CU, and the issue arbiter serves the OLDER one first on every SIMD: over a launch in which
            // each runs several trajectories the older one finished 18 % earlier and the younger one ran its last 400 us alone, at
            // half the CU's throughput (tools/wg_times.py: workgroup durations 1.90 / 2.31 ms at 1024 windows) -- the whole of
            // what the one-launch form used to lose against a grid of one-trajectory workgroups.  Priority outranks age, so
            // the two take turns: at the top of every pass a workgroup sets its waves' priority from a time slice of the
            // constant 100 MHz clock XOR its workgroup slot on the CU (HW_ID.TG_ID: 0 / 1) -- opposite for the two, flipping
            // together, independent of their progress.  About six slices per launch are best (the host sizes them,
            // launch_score_t): short ones cost throughput (a pass per slice: -2.5 %), one or two leave the tail
            // (profiles/r03s_prio_slices.txt).
            const int sh = P.prio_shift;
            if (sh > 0) {
                unsigned hwid;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                const unsigned slice = (unsigned)(__builtin_amdgcn_s_memrealtime() >> sh);
                if (((hwid >> 16) ^ slice) & 1u) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
        }
        // ---- step prologue: this step's noise z (layer 0's mix coefficients were fetched at the end of the previous pass)
        // this step's noise z: the Philox + Box-Muller cost is paid by the two waves that have no unit in the mixes of
        // layers 0 and 1 (6 units on 8 waves), half of the elements in each of those two stages; shapes whose layer-0 mix
        // keeps every wave busy generate it here with all threads
        constexpr bool NZ_TAIL = MixCfg<16, 17, T, NB>::UNITS <= NWAVES - 2;
        constexpr int NZ_T0 = NZ_TAIL ? NTHREADS - 128 : 0;
        // one thread per (chain, U-Net frame, pair of joints): a Philox4x32 call yields the four normals of the pair's two
        // coordinates -- NB*T*9 threads (54 at T = 3: one wave's worth of Philox per pass instead of four)
        constexpr int NZ_GROUPS = NB * T * 9;
        static_assert(NZ_GROUPS <= 128, "noise groups fit the two idle waves");
        auto noise_part = [&](int t_id) {
            const int gi = t_id - NZ_T0;
            if (P.mode == 0 && sidx > 1 && gi >= 0 && gi < NZ_GROUPS) {
                const int n = gi / (T * 9), r = gi % (T * 9), t = r / 9, v0 = (r % 9) * 2;
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                const int fixed = WM[n];
                if (!((fixed >> t) & 1)) {
                    const int b = WM[4 + n];
                    const int tx = fm_tx(P, fixed, t);
                    const int k = P.ns - sidx;
                    if (P.noise) {
                        const float* zp = P.noise + ((size_t)(s * K + k) * P.B + b) * CTV + tx * 17 + v0;
                        z[0] = zp[0]; z[1] = zp[Tx * 17];                                   // (c = 0, v0), (c = 1, v0)
                        if (v0 + 1 < 17) { z[2] = zp[1]; z[3] = zp[Tx * 17 + 1]; }         // (c = 0, v0+1), (c = 1, v0+1)
                    } else {
                        philox_normal4(P.seed, (unsigned)(tx * 9 + (v0 >> 1)), (unsigned)k, (unsigned)s, (unsigned)(P.first_window + b), z);
                    }
                }
                float* zo = ZN + ((n * T + t) * 17 + v0) * C0;          // ZN[col][c], columns v0 and v0+1
                zo[0] = z[0]; zo[1] = z[1];
                if (v0 + 1 < 17) { zo[2] = z[2]; zo[3] = z[3]; }
            }
        };
        noise_part(tid);
        STAGE(0);
        STAGE(1);
#ifdef MCD_PROFILE
        prof.bidx = 0;                                         // barrier slots count from the top of the pass
        prof.tr_on = prof.won && prof.tr && s == part && sidx == i_first - 1;      // time stamps: the second pass of the first trajectory
        prof.trace(0);
#endif
        // Every stage issues the coefficient loads of the stage after it (mcN = mix rows / fragments of layer N,
        // rcX = resampler fragments) before its own closing barrier, so no stage starts with an L2 round trip.
        auto mixload = [&](auto& mc, int l) { mc.load(wb + tab_i(wb, l * F_STRIDE + F_TQ), wb + tab_i(wb, l * F_STRIDE + F_AM), wave, lane); };
        auto rsload = [&](auto& rc, int r) { rc.load(wb + tab_i(wb, TAB_RSW + r), wb + tab_i(wb, TAB_RSB + r), lane); };
        auto mix_early = mixload;
        auto rs_early = rsload;
        NoHook nohook;
        // EARLY2 (kernels with two waves per SIMD, and the 6-frame one): the mix coefficients of the layer BEHIND a joint resampler are fetched two
        // stages ahead, at the end of the GEMM in front of the resampler (before its closing barrier, where the older wave of
        // each SIMD only waits): the resampler is too short to cover 36 loads per wave, and issued at its top they delayed its
        // MFMAs by ~2 k cycles
        // (6 frames: +1 % on top of the pre-barrier placement; 3 frames: -0.3 %, the 128-register budget has no room for it)
        constexpr bool EARLY2 = (MINW <= 2 || T == 6) && !BF3;
        LMix<1, T, NB> mc1;
        // layer 0 reads the chain state XT[col][4] in place (x in channels 0,1): its lanes' channels 2..15 are then other
        // columns' coordinates -- finite, and multiplied by the zero-padded K rows of the layer's weights -- so no 16-channel
        // copy of x has to be zeroed and rewritten every pass
        LAfr<1> A1; LAfr<2> A2; LAfr<3> A3; LAfr<4> A4; LAfr<5> A5; LAfr<7> A7; LAfr<8> A8; LAfr<9> A9;
        auto wearly = [&](auto& A, auto lc) { if constexpr (WEARLY) load_lafr<decltype(lc)::value>(A, wb, wave, lane); };
#define MCD_LC(l) std::integral_constant<int, l>{}
        layer_std<0, T, NB, (MINW <= 2), 4>(wb, mc0, XT, RG + PL::L0_z, RG + PL::L0_out, EMB, wave, lane, prof,
                            [&] { mix_early(mc1, 1); }, [&] { wearly(A1, MCD_LC(1)); }, WEARLY ? &A0 : nullptr);     // sp1a (2 -> 16)
        STAGE(2);
        lt_dump(0, RG + PL::L0_out, 20, 16, 17);
        lt_inject(1, RG + PL::L1_in, 20, 16, 17);
        // ---- down path
        LMix<2, T, NB> mc2;
        // SiLU(pe + cond) for the NEXT pass's embeddings (consumed in this pass's last layer): two global loads and an
        // exp -- by the idle waves too, not on wave 0's path at the top of the pass
        silu_row(sidx > 0 ? sidx - 1 : 0, tid - NZ_T0);
        layer_std<1, T, NB, (MINW <= 2)>(wb, mc1, RG + PL::L1_in, RG + PL::L1_z, RG + PL::L1_out, EMB, wave, lane, prof,
                            [&] { mix_early(mc2, 2); }, [&] { wearly(A2, MCD_LC(2)); }, WEARLY ? &A1 : nullptr);     // sd1.0
        STAGE(3);
        lt_dump(1, RG + PL::L1_out, 36, 32, 17);
        lt_inject(2, RG + PL::L2_in, 36, 32, 17);
        RsCoef<32, 17, 12, T, NB, true> rc1;
        LMix<3, T, NB> mc3;
        layer_std<2, T, NB, (MINW <= 2), cs_of(32), BF3>(wb, mc2, RG + PL::L2_in, RG + PL::L2_z, RG + PL::L2_out, EMB, wave, lane, prof,
                            [&] { rs_early(rc1, 0); }, [&] { if constexpr (EARLY2) mix_early(mc3, 3); }, WEARLY ? &A2 : nullptr);      // sd1.1 -> d1
        STAGE(4);
        lt_dump(2, RG + PL::L2_out, 36, 32, 17);
        lt_inject(11, RG + PL::L2_out, 36, 32, 17);
        if constexpr (!EARLY2) mix_early(mc3, 3);
        resample_stage<32, 17, 12, T, NB, true, false, (MINW <= 2)>(RG + PL::L2_out, 36, RG + PL::DN1_out, 36, rc1, skip1, wave, lane);  // down1 (captures d1)
        if constexpr (STASH1) {
            priv_float* sp = (priv_float*)stash1_mem;
            asm volatile("" : "+v"(sp));
#pragma unroll
            for (int i = 0; i < RS1::PER * RS1::SK; ++i) sp[i] = skip1[i];
        }
        wearly(A3, MCD_LC(3));
        bsync();
        STAGE(5);
        lt_dump(11, RG + PL::DN1_out, 36, 32, 12);
        lt_inject(3, RG + PL::L3_in, 36, 32, 12);
        LMix<4, T, NB> mc4;
        layer_std<3, T, NB, (MINW <= 2), cs_of(32), BF3>(wb, mc3, RG + PL::L3_in, RG + PL::L3_z, RG + PL::L3_out, EMB, wave, lane, prof,
                            [&] { mix_early(mc4, 4); }, [&] { wearly(A4, MCD_LC(4)); }, WEARLY ? &A3 : nullptr);     // sd2.0
        STAGE(6);
        lt_dump(3, RG + PL::L3_out, 68, 64, 12);
        lt_inject(4, RG + PL::L4_in, 68, 64, 12);
        RsCoef<64, 12, 10, T, NB, true> rc2;
        LMix<5, T, NB> mc5;
        layer_std<4, T, NB, (MINW <= 2), cs_of(64), BF3>(wb, mc4, RG + PL::L4_in, RG + PL::L4_z, RG + PL::L4_out, EMB, wave, lane, prof,
                            [&] { rs_early(rc2, 1); }, [&] { if constexpr (EARLY2) mix_early(mc5, 5); }, WEARLY ? &A4 : nullptr);      // sd2.1 -> d2
        STAGE(7);
        lt_dump(4, RG + PL::L4_out, 68, 64, 12);
        lt_inject(12, RG + PL::L4_out, 68, 64, 12);
        if constexpr (!EARLY2) mix_early(mc5, 5);
        resample_stage<64, 12, 10, T, NB, true, false, (MINW <= 2)>(RG + PL::L4_out, 68, RG + PL::DN2_out, 68, rc2, skip2, wave, lane);  // down2 (captures d2)
        if constexpr (STASH2) {
            priv_float* sp = (priv_float*)stash2_mem;
            asm volatile("" : "+v"(sp));            // opaque: the array stays in memory, plain (cached) scratch accesses
#pragma unroll
            for (int i = 0; i < RS2::PER * RS2::SK; ++i) sp[i] = skip2[i];
        }
        // wave-aligned units: the layer-5 mix reads only what this wave just wrote -> no barrier (see RsCfg::ALIGNED)
        constexpr bool FUSE64 = RS2::ALIGNED && MixCfg<64, 10, T, NB>::QC == T && MixCfg<64, 10, T, NB>::UNITS == NWAVES;
        wearly(A5, MCD_LC(5));
        if constexpr (!FUSE64) bsync();
        STAGE(8);
        lt_dump(12, RG + PL::DN2_out, 68, 64, 10);
        lt_inject(5, RG + PL::L5_in, 68, 64, 10);
        // ---- sd3.0, then sd3.1 (128 -> 64) W-first: P = [W_t; W_r] G, then out = PReLU(mix(P_t) + P_r + b) + e in place of P_r
        RsCoef<64, 10, 12, T, NB, false> rc3;
        LMix<7, T, NB> mc7;
        {
            constexpr int NT = PL::P10 / 16;
            constexpr int COLS = NB * T * 10;
            const LayerW lw = layer_w(wb, 6);
            float4 afr[8];
            MixCoef<64, 10, T, NB> mc6;
            layer_std<5, T, NB, (MINW <= 2), cs_of(64), BF3, BF3>(wb, mc5, RG + PL::L5_in, RG + PL::L5_z, RG + PL::L5_out, EMB, wave, lane, prof, nohook,
                                [&] {
                                    if constexpr (BF3) load_afrags_bf3<8, 4>(reinterpret_cast<const float4*>(wb + lw.wpb), wave, lane, afr, 0);
                                    else load_afrags<8, 8>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr, 0);
                                    if constexpr (EARLY2) mc6.load(wb + lw.tq, wb + lw.am, wave, lane);
                                }, WEARLY ? &A5 : nullptr);  // sd3.0
            STAGE(9);
            lt_dump(5, RG + PL::L5_out, 132, 128, 10);
            lt_inject(6, RG + PL::L6_in, 132, 128, 10);
            float* Pb = RG + PL::L6_p;
            if constexpr (!EARLY2) mc6.load(wb + lw.tq, wb + lw.am, wave, lane);
            auto epi6 = [&](auto ti, int col, int c0, f32x4 acc, int col0, int) {
                constexpr int STEP = Tiling<8, NT>::NG * 16 * 132;
                if (col < COLS) *reinterpret_cast<float4*>(Pb + __mul24(col0, 132) + c0 + decltype(ti)::value * STEP) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            };
            if constexpr (BF3) gemm_tiles_bf3<8, NT, 4, 0, false, true, true>(afr, RG + PL::L6_in, 132, RG + PL::L6_in, 132, wave, lane, epi6, 0);
            else gemm_tiles<8, NT, 8, 0, false, (MINW <= 2)>(afr, RG + PL::L6_in, 132, RG + PL::L6_in, 132, wave, lane, epi6, 0);
#pragma unroll
            for (int mi = 1; mi < Tiling<8, NT>::MW; ++mi) {
                if constexpr (BF3) {
                    load_afrags_bf3<8, 4>(reinterpret_cast<const float4*>(wb + lw.wpb), wave, lane, afr, mi);
                    gemm_tiles_bf3<8, NT, 4, 0, false, true, true>(afr, RG + PL::L6_in, 132, RG + PL::L6_in, 132, wave, lane, epi6, mi);
                } else {
                    load_afrags<8, 8>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr, mi);
                    gemm_tiles<8, NT, 8, 0, false, (MINW <= 2)>(afr, RG + PL::L6_in, 132, RG + PL::L6_in, 132, wave, lane, epi6, mi);
                }
            }
            if constexpr (EARLY2) { rs_early(rc3, 2); mix_early(mc7, 7); }      // up3's fragments, layer 7's mix coefficients
            bsync();
            STAGE(10);
            const float slope6 = lw.slope;
            const float pinf6 = prelu_bound(slope6);
            mix_stage<64, 10, T, NB, (MINW <= 2)>(Pb, 132, mc6, wb + lw.tq, wb + lw.am, wave, lane,
                                     [&](int n, int q, int w0, int c, std::true_type) {   // the fragment's 4 joints at once
                                         const float* pp = Pb + __mul24((n * T + q) * 10 + w0, 132) + 64 + c;
                                         // joints >= 10 read the next frame's rows (inside the 64-column region): an MFMA's D rows
                                         // are independent and those rows are never stored
                                         return f32x4{pp[0], pp[132], pp[264], pp[396]};
                                     },
                                     [&](int n, int q, int w0, int c, f32x4 v) {
                                         const float bias = BIA[c], e = EMB[n * EMB_STRIDE + emb_off(6) + c];
                                         float* pp = Pb + __mul24((n * T + q) * 10 + w0, 132) + 64 + c;
                                         const f32x2 t0 = f32x2{v[0], v[1]} + f32x2{bias, bias}, t1 = f32x2{v[2], v[3]} + f32x2{bias, bias};
                                         const f32x2 m0 = t0 * slope6, m1 = t1 * slope6;
                                         const f32x2 r0 = f32x2{__builtin_amdgcn_fmed3f(t0[0], m0[0], pinf6), __builtin_amdgcn_fmed3f(t0[1], m0[1], pinf6)} + f32x2{e, e};
                                         const f32x2 r1 = f32x2{__builtin_amdgcn_fmed3f(t1[0], m1[0], pinf6), __builtin_amdgcn_fmed3f(t1[1], m1[1], pinf6)} + f32x2{e, e};
                                         if (w0 < 10) { pp[0] = r0[0]; pp[132] = r0[1]; }
                                         if (w0 + 2 < 10) { pp[264] = r1[0]; pp[396] = r1[1]; }
                                     });
        }
        if constexpr (STASH2) {
            const priv_float* sp = (const priv_float*)stash2_mem;
            asm volatile("" : "+v"(sp));
#pragma unroll
            for (int i = 0; i < RS2::PER * RS2::SK; ++i) skip2[i] = sp[i];
        }
        if constexpr (!EARLY2) rs_early(rc3, 2);
        if constexpr (!FUSE64) bsync();     // aligned: up3 reads only this wave's own layer-6 output block
        STAGE(11);
        lt_dump(6, RG + PL::L6_p + 64, 132, 64, 10);
        lt_inject(13, RG + PL::L6_p + 64, 132, 64, 10);
        if constexpr (LT) {       // the resampler alone: no skip tensor added
            if (P.lt_stage == 13) for (float& f : skip2) f = 0.f;
            if (P.lt_stage == 14) for (float& f : skip1) f = 0.f;
        }
        // ---- up path
        if constexpr (!EARLY2) mix_early(mc7, 7);
        resample_stage<64, 10, 12, T, NB, false, true, (MINW <= 2)>(RG + PL::L6_p + 64, 132, RG + PL::UP3_out, 68, rc3, skip2, wave, lane);  // up3 (+ d2)
        wearly(A7, MCD_LC(7));
        bsync();
        STAGE(12);
        lt_dump(13, RG + PL::UP3_out, 68, 64, 12);
        lt_inject(7, RG + PL::L7_in, 68, 64, 12);
        LMix<8, T, NB> mc8;
        layer_std<7, T, NB, (MINW <= 2), cs_of(64), BF3>(wb, mc7, RG + PL::L7_in, RG + PL::L7_z, RG + PL::L7_out, EMB, wave, lane, prof,
                            [&] { mix_early(mc8, 8); }, [&] { wearly(A8, MCD_LC(8)); }, WEARLY ? &A7 : nullptr);     // su4.0
        STAGE(13);
        lt_dump(7, RG + PL::L7_out, 68, 64, 12);
        lt_inject(8, RG + PL::L8_in, 68, 64, 12);
        RsCoef<32, 12, 17, T, NB, false> rc4;
        LMix<9, T, NB> mc9;
        layer_std<8, T, NB, (MINW <= 2), cs_of(64), BF3>(wb, mc8, RG + PL::L8_in, RG + PL::L8_z, RG + PL::L8_out, EMB, wave, lane, prof,
                            [&] { rs_early(rc4, 3); },
                            [&] {
                                if constexpr (EARLY2) mix_early(mc9, 9);
                                if constexpr (STASH1) {
                                    const priv_float* sp = (const priv_float*)stash1_mem;
                                    asm volatile("" : "+v"(sp));
#pragma unroll
                                    for (int i = 0; i < RS1::PER * RS1::SK; ++i) skip1[i] = sp[i];
                                }
                            }, WEARLY ? &A8 : nullptr);                                            // su4.1
        STAGE(14);
        lt_dump(8, RG + PL::L8_out, 36, 32, 12);
        lt_inject(14, RG + PL::L8_out, 36, 32, 12);
        if constexpr (!EARLY2) mix_early(mc9, 9);
        resample_stage<32, 12, 17, T, NB, false, true, (MINW <= 2)>(RG + PL::L8_out, 36, RG + PL::UP2_out, 36, rc4, skip1, wave, lane);  // up2 (+ d1)
        wearly(A9, MCD_LC(9));
        bsync();
        STAGE(15);
        lt_dump(14, RG + PL::UP2_out, 36, 32, 17);
        lt_inject(9, RG + PL::L9_in, 36, 32, 17);
        // ---- su3.0, then su3.1 (32 -> 2) W-first: P = [W_t; W_r] M32 (4 useful rows), then the 2-channel mix with the PReLU,
        //      embedding, U-Net residual (+X) and the DDPM update fused into its store
        {
            const LayerW lw = layer_w(wb, 10);
            MixCoef<16, 17, T, NB> mc10;
            // next pass's embedding rows.  Unconditional (after the last pass the result is simply unused): a
            // conditionally loaded register struct costs ~35 VGPRs of phi copies here.
            EmbRow ef;
            auto ef_load = [&] { ef.load(wb, tid); };
            layer_std<9, T, NB, (MINW <= 2), cs_of(32), BF3>(wb, mc9, RG + PL::L9_in, RG + PL::L9_z, RG + PL::L9_out, EMB, wave, lane, prof,
                                [&] {
                                    mc10.load(wb + lw.tq, wb + lw.am, wave, lane);
                                    ef_load();
                                }, nohook, WEARLY ? &A9 : nullptr);                                      // su3.0
            STAGE(16);
            lt_dump(9, RG + PL::L9_out, 36, 32, 17);
            lt_inject(10, RG + PL::L10_in, 36, 32, 17);
            const float ca = srow[0], cb = srow[1], csg = srow[2];     // DDPM coefficients of this step (used two stages on)
            float* Pb = RG + PL::L10_p;
            // P[col][r] = sum_k W4[r][k] X[col][k] for the 4 useful rows (P_t 0,1 ; P_r 2,3) with plain FMAs: as a 16-row MFMA
            // tile this product is 3/4 padding, and matrix-pipe time is what the kernel is short of.  wave = (row, block
            // of 64 columns), weights as scalar operands.
            {
                const int r = wave & 3;
                const cfloat* w4 = (const cfloat*)(wb + lw.wp + r * 32);
                for (int cblk = wave >> 2; cblk * 64 < COLS17; cblk += NWAVES / 4) {
                    const int col = cblk * 64 + lane;
                    if (col < COLS17) {
                        const float* xp = RG + PL::L10_in + col * 36;
                        f32x2 acc2 = {0.f, 0.f};             // even / odd k partial sums: 16 v_pk_fma_f32
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            const float4 x = *reinterpret_cast<const float4*>(xp + 4 * q);
                            acc2 = f32x2{w4[4 * q + 0], w4[4 * q + 1]} * f32x2{x.x, x.y} + acc2;
                            acc2 = f32x2{w4[4 * q + 2], w4[4 * q + 3]} * f32x2{x.z, x.w} + acc2;
                        }
                        // (channels 4..15 of the mix input block keep whatever the region held: the mix never combines
                        // channels -- they are the N dimension of its MFMAs -- and only channels 0,1 of its output are stored)
                        Pb[col * 20 + r] = acc2[0] + acc2[1];
                    }
                }
            }
            STAGE(18);
            // next pass's embeddings: layers 0..9 straight into EMB (dead by now), layer 10's into the other E10 half
            emb_compute<NB>(ef, EXW, SEN, EMB, E10 + ((sidx - 1) & 1) * 16, tid);
            STAGE(19);
            bsync();
            STAGE(20);
            const float slope10 = lw.slope;
            const bool single = P.mode == 1, zadd = sidx > 1;
            const int e10_off = (sidx & 1) * 16;
            mix_stage<16, 17, T, NB, (MINW <= 2)>(Pb, 20, mc10, wb + lw.tq, wb + lw.am, wave, lane,
                                     ZeroInit{},
                                     [&](int n, int t, int w0, int c, auto val) {     // whole 4-joint fragments: one address, 4 stores
                                         if (c < C0) {
                                             float* zp = ZO + ((n * T + t) * 17 + w0) * C0 + c;
                                             if constexpr (std::is_same_v<decltype(val), f32x4>) {
#pragma unroll
                                                 for (int r = 0; r < 4; ++r)
                                                     if (w0 + r < 17) zp[r * C0] = val[r];
                                             } else {
                                                 *zp = val;
                                             }
                                         }
                                     });
            bsync();
            // element-wise tail of the pass, one (column, coordinate) per thread: eps = PReLU(mix(P_t) + P_r + b) + e + x
            // (layer 10 + the U-Net's residual), then the DDPM update of the frame this prediction drives and the next
            // pass's input block.  (Inside the mix's store functor this ran on 2 of every 16 lanes of 6 waves.)
            // Reads first, then (behind a barrier when a prediction drives a DIFFERENT frame than the one it was made at:
            // 'concat' with the condition at the end of the window) the writes: the frame a thread updates is then another
            // thread's U-Net residual input x.
            constexpr int TAIL_IT = (COLS17 * C0 + NTHREADS - 1) / NTHREADS;
            float xn_t[TAIL_IT];
            int dst_t[TAIL_IT];
#pragma unroll
            for (int it = 0; it < TAIL_IT; ++it) {
                const int u = tid + it * NTHREADS;
                dst_t[it] = -1;
                xn_t[it] = 0.f;
                if (u < COLS17 * C0) {
                    const int c = u % C0, col = u / C0;
                    const int tt = TT[u];
                    const int n = (tt >> 4) & 3, t = (tt >> 6) & 15, v = tt >> 10;
                    const float x = XT[col * 4 + c];
                    const float l10 = prelu(ZO[u] + Pb[col * 20 + C0 + c] + BIA[64 + c], slope10) + E10[e10_off + n * 4 + c];
                    const float eps = l10 + x;
                    if constexpr (LT) {      // layer 10 alone: without the U-Net's residual (+ x)
                        const int b = win0 + n;
                        if (P.lt_stage == 10 && b < P.B) P.lt_out[(((size_t)b * C0 + c) * T + t) * 17 + v] = l10;
                    }
                    if (single) {
                        const int b = win0 + n;
                        if (b < P.B && P.eps_out) P.eps_out[(((size_t)b * C0 + c) * T + t) * 17 + v] = eps;
                    } else {
                        const int cbase = UPD[tt & 15];
                        if (cbase >= 0) {
                            const int colp = cbase + v;
                            const float z = zadd ? ZN[colp * C0 + c] : 0.f;
                            xn_t[it] = ca * (XT[colp * 4 + c] - cb * eps) + csg * z;
                            dst_t[it] = colp * 4 + c;
                        }
                    }
                }
            }
            if (P.upd_shift) bsync();
#pragma unroll
            for (int it = 0; it < TAIL_IT; ++it)
                if (dst_t[it] >= 0) XT[dst_t[it]] = xn_t[it];
            mc0.load(wb + tab_i(wb, F_TQ), wb + tab_i(wb, F_AM), wave, lane);      // for the next pass
            wearly(A0, MCD_LC(0));
            STAGE(21);
            bsync();
            STAGE(17);
        }
    }
    if (P.mode == 1) break;

    // ---- per-chain loss: mean over (C, Tx, V) of loss_fn(x_0 - corrupt)   (mocodad.py:484)
    float* RED = RG;
    const int per = CTV;
    {
        asm volatile("" : "+s"(Q));
        tid_s = tid0;
        asm volatile("" : "+v"(tid_s));
        DataView dv;
        dv.data = Q->dv.data; dv.base = Q->dv.base; dv.sc = Q->dv.sc; dv.st = Q->dv.st; dv.trans = Q->dv.trans; dv.aff = Q->dv.aff;
        const int seg_len = Q->seg_len, Bq = Q->B, Sq = Q->S, loss_fn = Q->loss_fn;
        float* pose_out = Q->pose_out;
        float* loss_out = Q->loss_out;
        const bool wmask = Q->win_mask != nullptr;
        for (int u = tid_s; u < NB * per; u += NTHREADS) {
            const int n = u / per, e = u % per;
            const int c = e / (Tx * 17), tx = (e / 17) % Tx, v = e % 17;
            const bool valid = win0 + n < Bq;
            const int b = WM[4 + n];
            int tu = P.pos_of[tx];
            if (wmask) {                       // frame of the tx-th corrupt frame = tx-th clear bit of the window's mask
                const int fixed = WM[n];
                int cnt = 0;
                for (int t = 0; t < T; ++t)
                    if (!((fixed >> t) & 1)) { if (cnt == tx) tu = t; ++cnt; }
            }
            const float x0 = XT[((n * T + tu) * 17 + v) * 4 + c];
            const float gt = load_coord(dv, b, c, fm_src(P, tu), v, seg_len);
            const float d = fabsf(x0 - gt);
            float l;
            if (loss_fn == MCD_LOSS_SMOOTH_L1) l = d < 1.f ? 0.5f * d * d : d - 0.5f;
            else if (loss_fn == MCD_LOSS_L1) l = d;
            else l = d * d;
            RED[u] = l;
            if (valid && pose_out) pose_out[(size_t)(b * Sq + s) * per + e] = x0;
        }
        bsync();
        // two-level sum in a fixed order, through LDS: 8 partial sums per chain, then one thread per chain.  (A wave shuffle
        // reduction needs the lane id, which the compiler computes at kernel entry and keeps alive -- spilled -- across the
        // whole trajectory; this runs once per sample.)
        float* PART = RED + NB * per;
        if (tid_s < NB * 8) {
            const int n = tid_s >> 3, p8 = tid_s & 7;
            float sum = 0.f;
            for (int e = p8; e < per; e += 8) sum += RED[n * per + e];
            PART[tid_s] = sum;
        }
        bsync();
        if (tid_s < NB) {
            const int n = tid_s;
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += PART[n * 8 + k];
            const float l = sum / (float)per;
            if (win0 + n < Bq && loss_out) loss_out[(size_t)(win0 + n) * Sq + s] = l;
            if (s < 64) LOSSB[n * 64 + s] = l;
        }
    }
    bsync();        // RED (the work region) and XT are rewritten by the next sample
    }   // samples
#ifdef MCD_PROFILE
    __syncthreads();
    // per-workgroup start / end time (100 MHz ticks) and CU id: P.prof[4096 + 3 b + {0, 1, 2}]
    if (P.prof && tid0 == 0 && blockIdx.x < 6000) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        P.prof[4096 + 3 * blockIdx.x] = wg_t0;
        P.prof[4096 + 3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        P.prof[4096 + 3 * blockIdx.x + 2] = ((unsigned long long)(xcc & 0xf) << 32) | hwid;
    }
    if (prof.on) {
        for (int i = 0; i < PROF_SLOTS; ++i) P.prof[i] += prof.acc[i];
        for (int i = 0; i < PL::PROFTR; ++i) P.prof[PROF_SLOTS + i] = prof.acc[PROF_SLOTS + i];
    }
#endif
    // ---- aggregation over the samples (mocodad.py:454-520; loss-based strategies), when this workgroup has seen them all
    int te = tid0;
    asm volatile("" : "+v"(te));       // (opaque: win0 + tid from the prologue would otherwise be kept -- spilled -- until here)
    if (P.mode == 0 && P.loss_agg && te < NB && win0 + te < P.B)
        P.loss_agg[win0 + te] = aggregate_losses(LOSSB + te * 64, P.S, P.aggr, P.aggr_q);
}

// ------------------------------------------------------------------------------------------------
// condition encoder 'E_unet' (STSE_Unet with set_out_layer, stsae_unet.py:62-146,182-251): the U-Net's down path
// 2->16->32->32 | 17->12 | 32->64->64 | 12->10 | 64->128->6 without embeddings (t = None), then
// Linear(6*T*10 -> latent) over the (c,t,v) flattening.  Same MFMA stages and LDS plan as the scoring kernel.
// ------------------------------------------------------------------------------------------------
constexpr int TABC_URS = 56;               // down1 / down2 fragments + bias: 4 words
constexpr int TABC_ULW = 60, TABC_ULB = 61;  // to_time_dim weight [16][6*T*10] / bias
constexpr int CU_OUT = 6;                  // unet_down_channels[6] of STSE_Unet

template <int T, int NB>
struct CondUnetLds {     // the scoring kernel's work region, with the [P10][20] output of the last layer behind 2 x s128
    using PL = Plan<T, NB>;
    static constexpr int H_OFF = 2 * PL::s128;
    static constexpr int FLOATS = cmax(PL::R, H_OFF + PL::P10 * 20);
};

template <int T, int NB>
__global__ __launch_bounds__(NTHREADS, 2) void cond_unet_kernel(const float* wbuf, const DataView dv, const FrameIdx fi,
                                                                int seg_len, float* __restrict__ emb_out, int B) {
    using PL = Plan<T, NB>;
    constexpr int TV17 = T * 17, COLS17 = NB * TV17, TV10 = T * 10;
    constexpr int H_OFF = CondUnetLds<T, NB>::H_OFF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const RG = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b0 = blockIdx.x * NB;
    Prof prof;
    prof.off();
    for (int u = tid; u < CondUnetLds<T, NB>::FLOATS; u += NTHREADS) smem[u] = 0.f;
    __syncthreads();
    for (int u = tid; u < COLS17 * C0; u += NTHREADS) {
        const int c = u % C0, col = u / C0;
        const int n = col / TV17, t = (col / 17) % T, v = col % 17;
        const int b = b0 + n < B ? b0 + n : B - 1;
        RG[PL::L0_in + col * 20 + c] = load_coord(dv, b, c, fi.idx[t], v, seg_len);
    }
    __syncthreads();
    const float* wb = wbuf;
    auto lw = [&](int l) {
        LayerW w;
        w.tq = tab_i(wb, TABC + l * F_STRIDE + F_TQ); w.am = tab_i(wb, TABC + l * F_STRIDE + F_AM);
        w.wp = tab_i(wb, TABC + l * F_STRIDE + F_WP); w.bias = tab_i(wb, TABC + l * F_STRIDE + F_BIAS);
        w.slope = tab_f(wb, TABC + l * F_STRIDE + F_SLOPE);
        return w;
    };
    float nosk[1] = {0.f};
    layer_generic<16, 16, 17, true, false, T, NB>(wb, lw(0), RG + PL::L0_in, RG + PL::L0_z, RG + PL::L0_out, nullptr, wave, lane, prof, 0);
    layer_generic<16, 32, 17, true, false, T, NB>(wb, lw(1), RG + PL::L1_in, RG + PL::L1_z, RG + PL::L1_out, nullptr, wave, lane, prof, 0);
    layer_generic<32, 32, 17, false, false, T, NB>(wb, lw(2), RG + PL::L2_in, RG + PL::L2_z, RG + PL::L2_out, nullptr, wave, lane, prof, 0);
    {
        RsCoef<32, 17, 12, T, NB, false> rc;
        rc.load(wb + tab_i(wb, TABC + TABC_URS + 0), wb + tab_i(wb, TABC + TABC_URS + 1), lane);
        resample_stage<32, 17, 12, T, NB, false, false>(RG + PL::L2_out, 36, RG + PL::DN1_out, 36, rc, nosk, wave, lane);
        __syncthreads();
    }
    layer_generic<32, 64, 12, true, false, T, NB>(wb, lw(3), RG + PL::L3_in, RG + PL::L3_z, RG + PL::L3_out, nullptr, wave, lane, prof, 0);
    layer_generic<64, 64, 12, false, false, T, NB>(wb, lw(4), RG + PL::L4_in, RG + PL::L4_z, RG + PL::L4_out, nullptr, wave, lane, prof, 0);
    {
        RsCoef<64, 12, 10, T, NB, false> rc;
        rc.load(wb + tab_i(wb, TABC + TABC_URS + 2), wb + tab_i(wb, TABC + TABC_URS + 3), lane);
        resample_stage<64, 12, 10, T, NB, false, false>(RG + PL::L4_out, 68, RG + PL::DN2_out, 68, rc, nosk, wave, lane);
        __syncthreads();
    }
    layer_generic<64, 128, 10, true, false, T, NB>(wb, lw(5), RG + PL::L5_in, RG + PL::L5_z, RG + PL::L5_out, nullptr, wave, lane, prof, 0);
    layer_generic<128, 16, 10, true, false, T, NB>(wb, lw(6), RG + PL::L6_in, RG + PL::L6_p, RG + H_OFF, nullptr, wave, lane, prof, 0);
    // to_time_dim: emb[n][j] = b[j] + sum_k W[j][k] H[n][k], k = c*T*10 + t*10 + v.  thread = (n, j, part of 16)
    constexpr int F = CU_OUT * TV10;
    const float* H = RG + H_OFF;
    gfloat* W = as_global(wb + tab_i(wb, TABC + TABC_ULW));
    gfloat* bb = as_global(wb + tab_i(wb, TABC + TABC_ULB));
    for (int u = tid; u < NB * EDIM * 16; u += NTHREADS) {
        const int part = u & 15, jo = (u >> 4) % EDIM, n = u / (16 * EDIM);
        float a = 0.f;
        constexpr int NT16 = (TV10 + 15) / 16;      // compile-time trip counts: the weight loads are issued together (see cond_fast_body)
#pragma unroll
        for (int c = 0; c < CU_OUT; ++c) {
            float wv[NT16];
#pragma unroll
            for (int i = 0; i < NT16; ++i) wv[i] = (i * 16 + part < TV10) ? W[jo * F + c * TV10 + i * 16 + part] : 0.f;
#pragma unroll
            for (int i = 0; i < NT16; ++i) a = fmaf(wv[i], (i * 16 + part < TV10) ? H[(n * TV10 + i * 16 + part) * 20 + c] : 0.f, a);
        }
        a = row16_sum(a);
        if (part == 0 && b0 + n < B) emb_out[(size_t)(b0 + n) * EDIM + jo] = a + bb[jo];
    }
}

// ------------------------------------------------------------------------------------------------
// condition encoder (runtime channel list; 0.3 % of the work): one workgroup per window, VALU only.
// ------------------------------------------------------------------------------------------------
struct CondW {
    const float* base;
    int n_layers, Tc, latent, cmax;
    int gmode;       // 1: three LDS buffers of cmax x Tc x 17 do not fit (26 .. 31 condition frames): the third one lives in global scratch
    int cin[MCD_MAX_COND_LAYERS], cout[MCD_MAX_COND_LAYERS];
    int tq[MCD_MAX_COND_LAYERS], am[MCD_MAX_COND_LAYERS], wt[MCD_MAX_COND_LAYERS], wr[MCD_MAX_COND_LAYERS];
    int bias[MCD_MAX_COND_LAYERS];
    float slope[MCD_MAX_COND_LAYERS];
    int lw, lb;
};

// Stages of a layer as wave tasks of (8 channels, 64 columns): the time mix (Y = X . Tq per joint, into the layer's output
// buffer as scratch), the joint mix (Z = Y . A), the channel GEMM with 8 accumulators per thread whose weights are wave-uniform
// scalar loads.  (The first version ran the two mixes as one 17 x (T + 1) loop per output element: 8x the multiplies, 0.45
// TFLOP/s; at 16 condition frames it was a quarter of the whole scoring step.)
constexpr int CE_THREADS = 512;
// gbuf (W.gmode): one buffer of cmax x Tc x 17 floats per workgroup in global scratch -- the buffers rotate, so a different one of
// the three is the global one in every layer.
__global__ __launch_bounds__(CE_THREADS) void cond_encode_kernel(const CondW W, const float* __restrict__ cond,
                                                                 float* __restrict__ emb_out, int B, float* __restrict__ gbuf) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int Tc = W.Tc, TV = Tc * 17, nblk = (TV + 63) / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = CE_THREADS / 64;
    float* RED = smem + (gbuf ? 2 : 3) * W.cmax * TV;  // CE_THREADS partial sums
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
    float* X = smem;
    float* Z = X + W.cmax * TV;
    float* O = gbuf ? gbuf + (size_t)blockIdx.x * W.cmax * TV : Z + W.cmax * TV;
    __syncthreads();
    for (int u = tid; u < C0 * TV; u += CE_THREADS) X[u] = cond[(size_t)b * C0 * TV + u];  // (c, t, v) row-major
    __syncthreads();
    for (int l = 0; l < W.n_layers; ++l) {
        const int cin = W.cin[l], cout = W.cout[l];
        const float* Tq = W.base + W.tq[l];
        const float* Am = W.base + W.am[l];
        const int ngi = (cin + 7) / 8, ngo = (cout + 7) / 8;
        // time mix: Y[c][q, v] = sum_t X[c][t, v] Tq[q, v][t]   (Y in the output buffer)
        for (int task = wave; task < ngi * nblk; task += NW) {
            const int c0 = (task / nblk) * 8, p = (task % nblk) * 64 + lane;
            if (p < TV) {
                const float* tq = Tq + (size_t)p * Tc;
                const float* xb = X + p % 17;
                int co[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) co[i] = (c0 + i < cin ? c0 + i : cin - 1) * TV;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int t = 0; t < Tc; ++t) {
                    const float tv = tq[t];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(xb[co[i] + t * 17], tv, acc[i]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c0 + i < cin) O[(c0 + i) * TV + p] = acc[i];
            }
        }
        __syncthreads();
        // joint mix: Z[c][q, w] = sum_v Y[c][q, v] A[q, v][w]
        for (int task = wave; task < ngi * nblk; task += NW) {
            const int c0 = (task / nblk) * 8, p = (task % nblk) * 64 + lane;
            if (p < TV) {
                const int q = p / 17, w = p % 17;
                const float* am = Am + (size_t)q * 289 + w;
                const float* yb = O + q * 17;
                int co[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) co[i] = (c0 + i < cin ? c0 + i : cin - 1) * TV;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int v = 0; v < 17; ++v) {
                    const float a = am[v * 17];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(yb[co[i] + v], a, acc[i]);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (c0 + i < cin) Z[(c0 + i) * TV + p] = acc[i];
            }
        }
        __syncthreads();
        // channel GEMM + residual + PReLU: 8 output channels per thread, their weight rows wave-uniform
        const float* wt = W.base + W.wt[l];
        const float* wr = W.wr[l] >= 0 ? W.base + W.wr[l] : nullptr;
        const float* bias = W.base + W.bias[l];
        const float slope = W.slope[l];
        for (int task = wave; task < ngo * nblk; task += NW) {
            const int o0 = (task / nblk) * 8, p = (task % nblk) * 64 + lane;
            int row[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) row[i] = o0 + i < cout ? o0 + i : cout - 1;
            if (p < TV) {
                float acc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = bias[row[i]];
                for (int c = 0; c < cin; ++c) {
                    const float z = Z[c * TV + p];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(wt[row[i] * cin + c], z, acc[i]);
                }
                if (wr) {
                    for (int c = 0; c < cin; ++c) {
                        const float x = X[c * TV + p];
#pragma unroll
                        for (int i = 0; i < 8; ++i) acc[i] = fmaf(wr[row[i] * cin + c], x, acc[i]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += X[row[i] * TV + p];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (o0 + i < cout) O[(o0 + i) * TV + p] = prelu(acc[i], slope);
            }
        }
        __syncthreads();
        float* tmp = X; X = O; O = tmp;
    }
    // bottleneck Linear over the (c,t,v) flattening (stsae.py:73-89)
    const int hd = W.cout[W.n_layers - 1];
    const int F = hd * TV;
    const int jj = tid / 16, part = tid % 16;  // 16 partial sums per output
    for (int j0 = 0; j0 < W.latent; j0 += CE_THREADS / 16) {
        const int jo = j0 + jj;
        float a = 0.f;
        if (jo < W.latent) {
            const float* wrow = W.base + W.lw + (size_t)jo * F;
            for (int k = part; k < F; k += 16) a = fmaf(wrow[k], X[k], a);
        }
        RED[tid] = a;
        __syncthreads();
        if (part == 0 && jo < W.latent) {
            float s = W.base[W.lb + jo];
            for (int k = 0; k < 16; ++k) s += RED[jj * 16 + k];
            emb_out[(size_t)b * W.latent + jo] = s;
        }
        __syncthreads();
    }
    }
}

// ------------------------------------------------------------------------------------------------
// aggregation over the S samples (mocodad.py:454-520); one thread per window, S <= 64
// ------------------------------------------------------------------------------------------------
struct AggrParams {
    const float* loss_all; const float* pose_all; const float* data; float* loss_agg; float* pose_agg;
    int B, S, C, Tx, V, seg_len, strategy, loss_fn;
    float q;
    int corrupt_idx[MCD_MAX_FRAMES];
};

__device__ __forceinline__ float loss_elem(float a, float b, int fn) {
    const float d = fabsf(a - b);
    if (fn == MCD_LOSS_SMOOTH_L1) return d < 1.f ? 0.5f * d * d : d - 0.5f;
    if (fn == MCD_LOSS_L1) return d;
    return d * d;
}

__device__ __forceinline__ void sort_small(float* a, int n) {
    for (int i = 1; i < n; ++i) {
        const float x = a[i];
        int k = i - 1;
        while (k >= 0 && a[k] > x) { a[k + 1] = a[k]; --k; }
        a[k + 1] = x;
    }
}

__global__ void aggregate_kernel(const AggrParams P) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    const int S = P.S, per = P.C * P.Tx * P.V;
    const float* L = P.loss_all + (size_t)b * S;
    float tmp[64];
    if (P.strategy == MCD_AGGR_BEST || P.strategy == MCD_AGGR_WORST) {
        const bool best = P.strategy == MCD_AGGR_BEST;
        float cur = best ? 1e10f : -1.f;
        int sel = -1;
        for (int s = 0; s < S; ++s) {
            const bool m = best ? (L[s] < cur) : (L[s] > cur);
            if (m) { cur = L[s]; sel = s; }
        }
        P.loss_agg[b] = cur;
        if (P.pose_agg) for (int e = 0; e < per; ++e)
            P.pose_agg[(size_t)b * per + e] = sel >= 0 ? P.pose_all[((size_t)b * S + sel) * per + e] : 0.f;
    } else if (P.strategy == MCD_AGGR_MEAN) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += L[k];
        P.loss_agg[b] = s / (float)S;
    } else if (P.strategy == MCD_AGGR_MEDIAN || P.strategy == MCD_AGGR_QUANTILE) {
        for (int k = 0; k < S; ++k) tmp[k] = L[k];
        sort_small(tmp, S);
        if (P.strategy == MCD_AGGR_MEDIAN) {
            P.loss_agg[b] = tmp[(S - 1) / 2];  // torch.median: lower of the two middle values
        } else {
            const float pos = fminf(fmaxf(P.q, 0.f), 1.f) * (float)(S - 1);
            const int lo = (int)floorf(pos);
            const int hi = lo + 1 < S ? lo + 1 : S - 1;
            const float wgt = pos - (float)lo;
            const float a = tmp[lo], c = tmp[hi];
            P.loss_agg[b] = wgt < 0.5f ? a + wgt * (c - a) : c - (c - a) * (1.f - wgt);  // torch.lerp
        }
    } else {  // mean_pose / median_pose
        float acc = 0.f;
        for (int e = 0; e < per; ++e) {
            float val;
            if (P.strategy == MCD_AGGR_MEAN_POSE) {
                float s = 0.f;
                for (int k = 0; k < S; ++k) s += P.pose_all[((size_t)b * S + k) * per + e];
                val = s / (float)S;
            } else {
                for (int k = 0; k < S; ++k) tmp[k] = P.pose_all[((size_t)b * S + k) * per + e];
                sort_small(tmp, S);
                val = tmp[(S - 1) / 2];
            }
            if (P.pose_agg) P.pose_agg[(size_t)b * per + e] = val;
            const int c = e / (P.Tx * P.V), tx = (e / P.V) % P.Tx, v = e % P.V;
            const float gt = P.data[(((size_t)b * P.C + c) * P.seg_len + P.corrupt_idx[tx]) * P.V + v];
            acc += loss_elem(val, gt, P.loss_fn);
        }
        P.loss_agg[b] = acc / (float)per;
    }
}


// ------------------------------------------------------------------------------------------------
// Runtime-shape form of the trajectory kernel: ANY U-Net frame count 1..MCD_MAX_FRAMES (the reference is generic in
// n_frames, mocodad.py:780-796, stsgcn.py:134-141), every strategy.  Plain fp32 FMAs, one 256-thread workgroup per chain
// at a time (persistent grid), activations [channel][frame][joint] in a per-workgroup global scratch slab.  Correct, not fast,
// and since round 3 off every default path (score_kernel<T,...> covers 1 .. 12 frames, score_tiled_kernel 13 .. 32): it is the
// independent implementation MCD_OPT_GENERIC_UNET switches to, which the tests compare the MFMA kernels with.
// Same noise keys, same update, same loss as score_kernel.
// ------------------------------------------------------------------------------------------------
struct GLayer { int cin, cout, V, tq, am, wt, wr, bias, embo; float slope; };    // wr < 0: identity residual; embo < 0: no embedding
struct GenNet { GLayer L[NLAYERS]; int rs_w[4], rs_b[4], we, be; };
struct FrameMaps { int src_frame[MCD_MAX_FRAMES], tx_of[MCD_MAX_FRAMES], pos_of[MCD_MAX_FRAMES], upd_of[MCD_MAX_FRAMES]; };
constexpr int GEN_THREADS = 256;
constexpr int GEN_BUF = 1280;        // floats per frame of the three rotating buffers: 128 ch x 10 joints (>= 32 x 17, 64 x 12)
constexpr int GEN_D1 = 32 * 17, GEN_D2 = 64 * 12;
constexpr int GEN_SLAB = 3 * GEN_BUF + GEN_D1 + GEN_D2;      // per frame and workgroup

// one ST-GCN layer (stsgcn.py:94-116, BatchNorm folded): X [cin][T][V] -> O [cout][T][V]; Y (>= cin T V floats, may be O) and
// Z are scratch.  emb: the pass's embedding outputs (LDS) or null.
// Each stage as wave tasks of (8 channels, 64 columns) with 8 accumulators per thread: a column's activation (or coefficient)
// is loaded once for 8 multiply-adds, and the GEMM's weight rows are wave-uniform scalar loads (see cond_encode_kernel).
__device__ void g_layer(const float* wb, const GLayer& L, int T, const float* X, float* Y, float* Z, float* O, const float* emb) {
    const int V = L.V, TV = T * V, cin = L.cin, cout = L.cout, nblk = (TV + 63) / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    constexpr int NW = GEN_THREADS / 64;
    const float* Tq = wb + L.tq;      // [q][v][t]
    const float* Am = wb + L.am;      // [q][v][w]
    const int ngi = (cin + 7) / 8, ngo = (cout + 7) / 8;
    for (int task = wave; task < ngi * nblk; task += NW) {          // time mix: Y[c][q, v] = sum_t X[c][t, v] Tq[q, v][t]
        const int c0 = (task / nblk) * 8, p = (task % nblk) * 64 + lane;
        if (p < TV) {
            const float* tq = Tq + (size_t)p * T;
            const float* xb = X + p % V;
            int co[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) co[i] = (c0 + i < cin ? c0 + i : cin - 1) * TV;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < T; ++t) {
                const float tv = tq[t];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaf(xb[co[i] + t * V], tv, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c0 + i < cin) Y[(c0 + i) * TV + p] = acc[i];
        }
    }
    __syncthreads();
    for (int task = wave; task < ngi * nblk; task += NW) {          // joint mix: Z[c][q, w] = sum_v Y[c][q, v] A[q, v][w]
        const int c0 = (task / nblk) * 8, p = (task % nblk) * 64 + lane;
        if (p < TV) {
            const int q = p / V, w = p % V;
            const float* am = Am + (size_t)q * V * V + w;
            const float* yb = Y + q * V;
            int co[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) co[i] = (c0 + i < cin ? c0 + i : cin - 1) * TV;
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int v = 0; v < V; ++v) {
                const float a = am[v * V];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaf(yb[co[i] + v], a, acc[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (c0 + i < cin) Z[(c0 + i) * TV + p] = acc[i];
        }
    }
    __syncthreads();
    const float* wt = wb + L.wt;
    const float* wr = L.wr >= 0 ? wb + L.wr : nullptr;
    const float* bias = wb + L.bias;
    const float slope = L.slope;
    const bool has_emb = emb && L.embo >= 0;
    for (int task = wave; task < ngo * nblk; task += NW) {          // channel GEMM + residual + PReLU (+ embedding)
        const int o0 = (task / nblk) * 8, p = (task % nblk) * 64 + lane;
        int row[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) row[i] = o0 + i < cout ? o0 + i : cout - 1;
        if (p < TV) {
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = bias[row[i]];
            for (int c = 0; c < cin; ++c) {
                const float z = Z[c * TV + p];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = fmaf(wt[row[i] * cin + c], z, acc[i]);
            }
            if (wr) {
                for (int c = 0; c < cin; ++c) {
                    const float x = X[c * TV + p];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = fmaf(wr[row[i] * cin + c], x, acc[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += X[row[i] * TV + p];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (o0 + i < cout) O[(o0 + i) * TV + p] = prelu(acc[i], slope) + (has_emb ? emb[L.embo + row[i]] : 0.f);
        }
    }
    __syncthreads();
}
// joint resampler (stsgcn.py:187-199 over the joint axis): X [C][T][vin] -> O [C][T][vout] (+ skip)
__device__ void g_resample(const float* wb, int wo, int bo, int C, int T, int vin, int vout, const float* X, float* O, const float* skip) {
    const float* W = wb + wo;
    const float* bb = wb + bo;
    for (int u = threadIdx.x; u < C * T * vout; u += GEN_THREADS) {
        const int vo = u % vout, ct = u / vout;
        float a = bb[vo];
        for (int v = 0; v < vin; ++v) a = fmaf(W[vo * vin + v], X[ct * vin + v], a);
        if (skip) a += skip[u];
        O[u] = a;
    }
    __syncthreads();
}

__global__ __launch_bounds__(GEN_THREADS) void score_generic_kernel(const ScoreParams P, const FrameMaps M, const GenNet N, int T,
                                                                    float* __restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int TV = T * 17, CTV = C0 * TV, tid = threadIdx.x;
    float* XT = gsm;                  // chain state [c][t][v] over the U-Net frames
    float* EPS = XT + CTV;            // layer 10's output (+ x)
    float* ZN = EPS + CTV;            // this step's noise at the U-Net frames
    float* EMB = ZN + CTV;            // [EMB_TOTAL + 4]
    float* SE = EMB + EMB_TOTAL + 4;  // [16]
    float* RED = SE + EDIM;           // [GEN_THREADS]
    float* slab = scratch + (size_t)blockIdx.x * GEN_SLAB * T;
    float* A = slab;
    float* Bb = A + GEN_BUF * T;
    float* Zb = Bb + GEN_BUF * T;
    float* D1 = Zb + GEN_BUF * T;
    float* D2 = D1 + GEN_D1 * T;
    const float* wb = P.wbuf;
    const int Tx = P.n_corrupt;
    const int K = P.ns > 2 ? P.ns - 1 : 1;
    const int per = C0 * Tx * 17;
    for (long long chain = blockIdx.x; chain < P.n_chains; chain += gridDim.x) {
        const int b = (int)(chain / P.S), s = (int)(chain % P.S);
        const unsigned fixed = (unsigned)(P.win_mask ? P.win_mask[b] : P.fixed_mask);
        auto tx_of = [&](int t) { return P.win_mask ? __popc(~fixed & ((1u << t) - 1u)) : M.tx_of[t]; };
        auto src_of = [&](int t) { return P.win_mask ? t : M.src_frame[t]; };
        __syncthreads();
        for (int u = tid; u < CTV; u += GEN_THREADS) {
            const int c = u / TV, t = (u % TV) / 17, v = u % 17;
            float x;
            if (P.mode == 1) x = P.x_in[((size_t)b * C0 + c) * TV + t * 17 + v];
            else if ((fixed >> t) & 1u) x = load_coord(P.dv, b, c, src_of(t), v, P.seg_len);
            else {
                const int e = (c * Tx + tx_of(t)) * 17 + v;
                x = P.noise ? P.noise[((size_t)(s * K + 0) * P.B + b) * per + e]
                            : philox_normal(P.seed, (unsigned)e, 0u, (unsigned)s, (unsigned)(P.first_window + b));
            }
            XT[u] = x;
        }
        const int i_first = P.mode == 1 ? P.step_single : P.ns - 1;
        const int i_last = P.mode == 1 ? P.step_single : 1;
        for (int sidx = i_first; sidx >= i_last; --sidx) {
            const float* srow = P.step_table + sidx * (4 + EDIM);
            __syncthreads();
            if (tid < EDIM) {
                float e = srow[4 + tid];
                if (P.cond_emb) e += P.cond_emb[(size_t)b * EDIM + tid];
                SE[tid] = e / (1.f + expf(-e));
            }
            // this step's noise, one thread per (frame, joint pair) like score_kernel (same Philox keys)
            if (P.mode == 0 && sidx > 1) {
                const int k = P.ns - sidx;
                for (int gi = tid; gi < T * 9; gi += GEN_THREADS) {
                    const int t = gi / 9, v0 = (gi % 9) * 2;
                    float z[4] = {0.f, 0.f, 0.f, 0.f};
                    if (!((fixed >> t) & 1u)) {
                        const int tx = tx_of(t);
                        if (P.noise) {
                            const float* zp = P.noise + ((size_t)(s * K + k) * P.B + b) * per + tx * 17 + v0;
                            z[0] = zp[0]; z[1] = zp[Tx * 17];
                            if (v0 + 1 < 17) { z[2] = zp[1]; z[3] = zp[Tx * 17 + 1]; }
                        } else {
                            philox_normal4(P.seed, (unsigned)(tx * 9 + (v0 >> 1)), (unsigned)k, (unsigned)s, (unsigned)(P.first_window + b), z);
                        }
                    }
                    ZN[t * 17 + v0] = z[0]; ZN[TV + t * 17 + v0] = z[1];
                    if (v0 + 1 < 17) { ZN[t * 17 + v0 + 1] = z[2]; ZN[TV + t * 17 + v0 + 1] = z[3]; }
                }
            }
            __syncthreads();
            for (int o = tid; o < EMB_TOTAL; o += GEN_THREADS) {
                const float* we = wb + N.we + o * EDIM;
                float a = wb[N.be + o];
                for (int k = 0; k < EDIM; ++k) a = fmaf(we[k], SE[k], a);
                EMB[o] = a;
            }
            __syncthreads();
            // ---- the U-Net (stsae_unet.py:406-438)
            g_layer(wb, N.L[0], T, XT, A, Zb, A, EMB);
            g_layer(wb, N.L[1], T, A, Bb, Zb, Bb, EMB);
            g_layer(wb, N.L[2], T, Bb, D1, Zb, D1, EMB);                                         // d1
            g_resample(wb, N.rs_w[0], N.rs_b[0], 32, T, 17, 12, D1, A, nullptr);                  // down1
            g_layer(wb, N.L[3], T, A, Bb, Zb, Bb, EMB);
            g_layer(wb, N.L[4], T, Bb, D2, Zb, D2, EMB);                                         // d2
            g_resample(wb, N.rs_w[1], N.rs_b[1], 64, T, 12, 10, D2, A, nullptr);                  // down2
            g_layer(wb, N.L[5], T, A, Bb, Zb, Bb, EMB);
            g_layer(wb, N.L[6], T, Bb, A, Zb, A, EMB);
            g_resample(wb, N.rs_w[2], N.rs_b[2], 64, T, 10, 12, A, Bb, D2);                       // up3 + d2
            g_layer(wb, N.L[7], T, Bb, A, Zb, A, EMB);
            g_layer(wb, N.L[8], T, A, Bb, Zb, Bb, EMB);
            g_resample(wb, N.rs_w[3], N.rs_b[3], 32, T, 12, 17, Bb, A, D1);                       // up2 + d1
            g_layer(wb, N.L[9], T, A, Bb, Zb, Bb, EMB);
            g_layer(wb, N.L[10], T, Bb, A, Zb, EPS, EMB);
            // ---- eps = U-Net output + its input; DDPM update of the frame each prediction drives (mocodad.py:172-178,829-838)
            const float ca = srow[0], cb = srow[1], csg = srow[2];
            const bool zadd = sidx > 1;
            float xn[(C0 * MCD_MAX_FRAMES * 17 + GEN_THREADS - 1) / GEN_THREADS];
            int dst[(C0 * MCD_MAX_FRAMES * 17 + GEN_THREADS - 1) / GEN_THREADS];
            int it = 0;
            for (int u = tid; u < CTV; u += GEN_THREADS, ++it) {
                const int c = u / TV, t = (u % TV) / 17, v = u % 17;
                const float eps = EPS[u] + XT[u];
                dst[it] = -1; xn[it] = 0.f;
                if (P.mode == 1) {
                    P.eps_out[((size_t)b * C0 + c) * TV + t * 17 + v] = eps;
                } else {
                    const int k = P.win_mask ? (((fixed >> t) & 1u) ? -1 : 0) : M.upd_of[t];
                    if (k >= 0) {
                        const int tp = P.win_mask ? t : M.pos_of[k];
                        const int up = c * TV + tp * 17 + v;
                        xn[it] = ca * (XT[up] - cb * eps) + csg * (zadd ? ZN[up] : 0.f);
                        dst[it] = up;
                    }
                }
            }
            __syncthreads();
            it = 0;
            for (int u = tid; u < CTV; u += GEN_THREADS, ++it)
                if (dst[it] >= 0) XT[dst[it]] = xn[it];
        }
        if (P.mode == 1) continue;
        __syncthreads();
        // ---- loss over the corrupt frames (mocodad.py:484)
        float part = 0.f;
        for (int e = tid; e < per; e += GEN_THREADS) {
            const int c = e / (Tx * 17), tx = (e / 17) % Tx, v = e % 17;
            int tu = M.pos_of[tx];
            if (P.win_mask) { int cnt = 0; for (int t = 0; t < T; ++t) if (!((fixed >> t) & 1u)) { if (cnt == tx) tu = t; ++cnt; } }
            const float x0 = XT[c * TV + tu * 17 + v];
            const float gt = load_coord(P.dv, b, c, src_of(tu), v, P.seg_len);
            part += loss_elem(x0, gt, P.loss_fn);
            if (P.pose_out) P.pose_out[(size_t)(b * P.S + s) * per + e] = x0;
        }
        RED[tid] = part;
        __syncthreads();
        for (int o = GEN_THREADS / 2; o > 0; o >>= 1) { if (tid < o) RED[tid] += RED[tid + o]; __syncthreads(); }
        if (tid == 0) P.loss_out[chain] = RED[0] / (float)per;
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA path for 12 < T_u <= 32 U-Net frames (concat over 24 frames, 16 + 16, ...): the activations of such a chain do not
// fit LDS (257 KB at layer 5 of a 24-frame chain), so the chain lives in a slab of global memory (L2 / Infinity Cache) and a
// layer is "32 input channels of ALL frames -> LDS -> mix -> channel GEMM -> slab", one 512-thread workgroup per CU:
//   X      one 32-channel part of the layer's input, every frame, staged slab -> registers -> LDS (the next part's loads in
//          flight behind this part's stages).  The four joint resamplers are not stages of their own: the layer behind one
//          builds its X from the resampler's input rows (chunk of frames by chunk, resample_stage LDS -> LDS, + the U-Net skip)
//   mix    both halves on the matrix cores: tl_time_mix (the (frames x frames) time mix of a joint as one MFMA product, Tq
//          pre-packed as A fragments) writes Y to LDS, tl_joint_mix turns it into z in place.  z never leaves LDS
//   GEMM   gemm_part: z . W_t + x . W_r (or + x) of the part into register accumulators (<= 80 per lane); layers with 64 or
//          128 input channels sum their 2 or 4 parts there; one epilogue (bias, PReLU, embedding) -> slab
//   layers at 17 joints (32 input channels, X = 78 KB at 32 frames) take their frames in two groups so that X + z fit
//   layer 6 runs mix-first here (the specialised kernels run it W-first); layer 10 W-first on plain FMAs + mix_long
//   hand-overs: no layer waits for the slab.  A layer's epilogue writes what the next layer reads first straight into LDS --
//          its first 32-channel part (3->4, 5->6, 7->8; the whole output across the 17-joint layers 0->1->2, 9->10, group 0's
//          accumulators held back until group 1's time mix has read the old rows), or the next layer's resampler input (all of
//          channels 0..31: 4->5, 6->7; its first chunk: 2->3, 8->9).  The slab keeps the later parts and the skips d1 / d2
// The frame count is padded to TP = 16, 24 or 32 with zero mixing coefficients (a padded frame's activations are finite
// garbage that no real frame ever reads); two 16-frame chains share a workgroup (<16, 2>: the stage lengths of 32 frames).
// Same noise keys, update, loss and strategies as score_kernel.  Slab traffic: 9 k floats per frame and pass (the first
// version, every stage through the slab: 37 k -- it ran at the HBM / fabric roofline, 4.6 TB/s, profiles/README.md).
// ------------------------------------------------------------------------------------------------
struct TiledNet {
    int tq[NLAYERS], am[NLAYERS], wp[NLAYERS], bias[NLAYERS];
    int tqm[NLAYERS];    // time-mix coefficients as MFMA A fragments (tl_time_mix): [joint][frame tile][k-step][lane]
    float slope[NLAYERS];
    int rsw[4];          // joint resamplers, non-capture fragment packs (RsCoef chunks: fragments then bias)
    int we, be;
};
// frames per chunk of a fused joint resampler (its input rows of those frames pass through the z region): 16, 12 at 24 frames
__host__ __device__ constexpr int tl_fc(int TP) { return TP == 24 ? 12 : 16; }
__host__ __device__ constexpr int tl_ra_floats(int TP) {
    // LDS work region of a layer: X (32 channels of all frames, + pad rows) and z (the same; half the frames at 17 joints).
    // (A fused resampler's input chunk passes through the z region.)
    return cmax((TP * 17 + 16 + TP * 17 / 2 + 16) * 36, 2 * (TP * 12 + 16) * 36);
}
__host__ __device__ constexpr int tl_qc(int TP) { return TP % 3 == 0 ? 3 : 4; }     // output frames per mix unit (6 at 24 frames: 108 coefficient registers, spills)
__host__ __device__ constexpr long long tl_slab_floats(int TP) {
    // A0, A1 (ping-pong, up to 128 ch x 10 joints), the skips D1, D2 -- each with 16 rows of padding behind it
    return (long long)2 * (TP * 10 + 16) * 132 + (long long)(TP * 17 + 16) * 36 + (long long)(TP * 12 + 16) * 68;
}

// cooperative copies between the slab and LDS, `ch` channels (multiple of 4) from channel ch0 of `rows` rows
__device__ __forceinline__ void tl_g2l(float* dst, int ds, const float* src, int ss, int ch0, int ch, int rows) {
    const int q = ch >> 2;
    for (int u = threadIdx.x; u < rows * q; u += NTHREADS) {
        const int r = u / q, c = (u - r * q) * 4;
        *reinterpret_cast<float4*>(dst + r * ds + c) = load_global4(src + (size_t)r * ss + ch0 + c);
    }
}
// slab -> LDS through registers, in two halves: issue() puts the global loads in flight (typically one stage ahead, so that
// their L2 latency runs behind the stage's MFMAs), commit() writes them to LDS once the region is free
template <int ROWS, int CH>
struct TlStage {
    static constexpr int Q = CH / 4, N = (ROWS * Q + NTHREADS - 1) / NTHREADS;
    float4 v[N];
    __device__ __forceinline__ void issue(int tid, const float* src, int ss, int ch0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int u = tid + i * NTHREADS;
            if (u < ROWS * Q) { const int r = u / Q, c = (u - r * Q) * 4; v[i] = load_global4(src + (size_t)r * ss + ch0 + c); }
        }
    }
    __device__ __forceinline__ void commit(int tid, float* dst, int ds) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int u = tid + i * NTHREADS;
            if (u < ROWS * Q) { const int r = u / Q, c = (u - r * Q) * 4; *reinterpret_cast<float4*>(dst + r * ds + c) = v[i]; }
        }
    }
};

// zero start of a mix's accumulators: the 4-joint fragment form and the single-joint form (joint 16 of the V = 17 layers)
struct ZeroInitL {
    __device__ __forceinline__ f32x4 operator()(int, int, int, std::true_type) const { return f32x4{0.f, 0.f, 0.f, 0.f}; }
    __device__ __forceinline__ float operator()(int, int, int) const { return 0.f; }
};
// mix of CINV (16 or 32) channels over ALL TP frames: X (LDS, [frame * V + joint][channel], stride cs) -> store functor.
// unit = (16-channel block, QC output frames); joint mix on the matrix cores exactly as in mix_stage, the time mix as
// tm_step groups over one k-step's TP input frames at a time.
template <int CINV, int V, int TP, int NB = 1>
struct MixLongCoef {      // time-mix rows + joint-mix fragments of one unit's QC output frames (NB chains of TP frames each)
    static constexpr int QC = tl_qc(TP), KS = (V + 3) / 4, MT = (V + 15) / 16, CB = CINV / 16, NQ = TP / QC;
    static constexpr int UNITS = CB * NB * NQ, PER = (UNITS + NWAVES - 1) / NWAVES, NR = (KS * TP + 15) / 16;
    float tq[QC][NR], aop[QC][MT][KS];
    // u: unit index in the flat list of CB x (NB * NQ) units (clamped: waves without a unit fetch the last one's)
    __device__ __forceinline__ void load(const float* tqd, const float* af, int u, int lane) {
        gfloat* tqd_g = as_global(tqd);
        gfloat* af_g = as_global(af);
        const int q0 = (((u < UNITS ? u : UNITS - 1) / CB) % NQ) * QC;
#pragma unroll
        for (int qi = 0; qi < QC; ++qi) {
#pragma unroll
            for (int r = 0; r < NR; ++r) tq[qi][r] = tqd_g[((q0 + qi) * NR + r) * 64 + lane];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) aop[qi][mt][ks] = af_g[(((q0 + qi) * MT + mt) * KS + ks) * 64 + lane];
        }
    }
};
// `first`: the coefficients of the wave's first unit, fetched by the caller before it waited for X to land in LDS
// NGRP > 1: only the output frames of frame group `grp` (the flat frame list cut in NGRP equal parts) -- the caller's z region
// holds one group at a time
template <int CINV, int V, int TP, int NB, int NGRP = 1, class Init, class Store>
__device__ __forceinline__ void mix_long(const float* __restrict__ X, int cs, const MixLongCoef<CINV, V, TP, NB>& first,
                                         const float* __restrict__ tqd, const float* __restrict__ af,
                                         int wave, int lane, Init&& init, Store&& store, int grp = 0) {
    using MC = MixLongCoef<CINV, V, TP, NB>;
    constexpr int QC = MC::QC, KS = MC::KS, KP = 2 * (KS / 2), MT = MC::MT, CB = MC::CB, NQ = MC::NQ;
    static_assert((NB * NQ) % NGRP == 0, "frame groups hold whole mix units");
    constexpr int UNITS = MC::UNITS / NGRP, PER = (UNITS + NWAVES - 1) / NWAVES;
    constexpr bool J16 = V == 17;
    constexpr int MTM = J16 ? 1 : MT;
    const int j = lane & 15, g = lane >> 4;
    static_for<PER>([&](auto rr) {
        constexpr int rnd = decltype(rr)::value;
        if (wave + rnd * NWAVES >= UNITS) return;
        const int u = wave + rnd * NWAVES + (NGRP > 1 ? grp * UNITS : 0);
        // (q0: first output frame of the unit in the flat list of NB * TP frames; its chain's frames start at row fo * V)
        const int cb = u % CB, qg = u / CB, fo = NB > 1 ? (qg / NQ) * TP : 0, q0 = fo + (qg % NQ) * QC;
        MC later;
        if constexpr (rnd > 0) later.load(tqd, af, u, lane);        // (a second live set of 50 .. 80 registers for a prefetch does not fit)
        const MC& cur = rnd > 0 ? later : first;
        const auto& tq = cur.tq;
        const auto& aop = cur.aop;
        f32x4 acc[QC][MTM];
        float part[QC];
#pragma unroll
        for (int qi = 0; qi < QC; ++qi) {
            part[qi] = 0.f;
#pragma unroll
            for (int mt = 0; mt < MTM; ++mt) acc[qi][mt] = init(q0 + qi, mt * 16 + 4 * g, cb * 16 + j, std::true_type{});
        }
        const float* xin_p = X + __mul24(fo * V + 4 * (g & 1) + (g >> 1), cs) + cb * 16 + j;
        const float* xin_l = X + __mul24(fo * V + g, cs) + cb * 16 + j;
        static_for<KS>([&](auto si) {
            constexpr int ks = decltype(si)::value;
            constexpr int vbase = ks < KP ? 8 * (ks >> 1) + 2 * (ks & 1) : 4 * KP;
            const float* xb = ks < KP ? xin_p : xin_l;
            float x[TP];
#pragma unroll
            for (int t = 0; t < TP; ++t) x[t] = xb[(t * V + vbase) * cs];
            __builtin_amdgcn_sched_barrier(0);
            float y[QC];
            static_for<TP>([&](auto ti) {
                constexpr int t = decltype(ti)::value;
                float c[QC];
#pragma unroll
                for (int qi = 0; qi < QC; ++qi) c[qi] = tq[qi][(ks * TP + t) / 16];
                tm_step<QC, (ks * TP + t) % 16, t == 0, t == TP - 1>(y, c, x[t]);
            });
            static_for<QC>([&](auto qq) {
                constexpr int qi = decltype(qq)::value;
#pragma unroll
                for (int mt = 0; mt < MTM; ++mt)
                    acc[qi][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[qi][mt][ks], y[qi], acc[qi][mt], 0, 0, 0);
                if constexpr (J16) part[qi] = fmaf(aop[qi][1][ks], y[qi], part[qi]);
            });
        });
#pragma unroll
        for (int qi = 0; qi < QC; ++qi) {
#pragma unroll
            for (int mt = 0; mt < MTM; ++mt) store(q0 + qi, mt * 16 + 4 * g, cb * 16 + j, acc[qi][mt]);
            if constexpr (J16) {
                const unsigned pu = __float_as_uint(part[qi]);
                const auto h = __builtin_amdgcn_permlane32_swap(pu, pu, false, false);
                const unsigned v2 = __float_as_uint(__uint_as_float(h[0]) + __uint_as_float(h[1]));
                const auto f = __builtin_amdgcn_permlane16_swap(v2, v2, false, false);
                const float z16 = __uint_as_float(f[0]) + __uint_as_float(f[1]);
                if (g == 0) store(q0 + qi, 16, cb * 16 + j, z16 + init(q0 + qi, 16, cb * 16 + j));
            }
        }
    });
}

// The two halves of a layer's mix in the slab-tiled kernel, both on the matrix cores (at 16 .. 32 frames the time mix is a third
// of the layer's multiply-adds: as DPP FMAs it was 40 % of the kernel).
// (1) time mix  Y[q][v][c] = sum_t Tq[q][v][t] X[t][v][c]: per (joint v, 16-channel block, tile of 16 output frames) one
//     (16 frames x TP frames) . (TP frames x 16 channels) product; A = the pre-packed Tq fragments (tl_tqm_floats), B = X read from
//     LDS with the frame index on the k axis; the result rows (frames) go to Y[(frame * V + v)][channel] in LDS.
// (2) joint mix, in place  Z[q][w][c] = sum_v A_q[v][w] Y[q][v][c]: per (frame, 16-channel block) the fragments of mix_stage,
//     B = Y read from LDS; a unit has read all of its frame's Y when it writes Z over it.
// NGRP: frame groups of the layer (the flat list of NB * TP frames in NGRP equal parts; Y / Z hold group `grp`).
template <int TP, int NB, int NGRP>
struct TlGroups {
    static constexpr int FG = NB * TP / NGRP;                   // frames of a group
    static constexpr int NCH = NB >= NGRP ? NB / NGRP : 1;      // chains a group spans
    static constexpr int FGC = FG / NCH;                        // frames of a group in one chain
    static constexpr int MTG = (FGC + 15) / 16;                 // 16-frame tiles of a group per chain
    static constexpr int NTC = MTG * (TP / FGC);                // ... of a chain (table rows)
    static constexpr int KT = TP / 4;
};
__host__ __device__ constexpr int tl_ngrp(int V) { return V == 17 ? 2 : 1; }
// A unit of (1) is (joint v, chain of the group, frame tile): its Tq fragments serve all 16-channel blocks, and the next unit's
// are fetched (L2) while this one runs; a unit of (2) is a frame, likewise.
// (the FIRST unit's fragments come from the caller, who fetched them -- tl_time_fetch / tl_joint_fetch -- a stage earlier)
// unit -> (v, tile, chain of the group): v fastest, so that the waves of a round read neighbouring rows
template <int V, int TP, int NB, int NGRP>
__device__ __forceinline__ void tl_time_fetch(float (&a)[TP / 4], const float* __restrict__ tqm, int u, int lane, int grp) {
    using G = TlGroups<TP, NB, NGRP>;
    constexpr int UNITS = V * G::NCH * G::MTG;
    if (u >= UNITS) u = UNITS - 1;
    const int v = u % V, m = (u / V) % G::MTG;
    gfloat* ap = as_global(tqm) + ((v * G::NTC + (NB >= NGRP ? 0 : grp * G::MTG) + m) * G::KT) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < G::KT; ++ks) a[ks] = ap[ks * 64];
}
// The wave's first unit runs alone on coefficients the caller fetched a stage ahead; the others' coefficients are fetched in
// front of it and those units then run together (their LDS reads, then their MFMA chains interleaved: with two waves per SIMD
// a single unit's read -> 8 dependent MFMAs -> write sequence leaves the matrix pipe idle most of the time).
template <int CINV, int V, int TP, int NB, int NGRP>
__device__ __forceinline__ void tl_time_mix(const float* __restrict__ X, int cs, float* __restrict__ Y, int csy,
                                            const float* __restrict__ tqm, int wave, int lane, int grp, const float (&first)[TP / 4]) {
    using G = TlGroups<TP, NB, NGRP>;
    constexpr int CB = CINV / 16, KT = G::KT, UNITS = V * G::NCH * G::MTG;
    constexpr int PER = (UNITS + NWAVES - 1) / NWAVES, NR = PER > 1 ? PER - 1 : 1;
    const int j = lane & 15, g = lane >> 4;
    float ar[NR][KT];
    if constexpr (PER > 1) {
#pragma unroll
        for (int i = 0; i < PER - 1; ++i) tl_time_fetch<V, TP, NB, NGRP>(ar[i], tqm, wave + (i + 1) * NWAVES, lane, grp);
    }
    // (all LDS reads of the wave's units first, the later units' behind the first one's: they run under its MFMAs.  The same
    // order in tl_joint_mix costs registers the 16- and 32-frame kernels do not have: -1.2 % / +0.4 %, not taken)
    struct Unit { int v, m, c; };
    auto unit_of = [&](int r) {
        const int u0 = wave + r * NWAVES, u = u0 < UNITS ? u0 : UNITS - 1;          // (a wave past the end repeats the last unit, unstored)
        return Unit{u % V, (u / V) % G::MTG, u / (V * G::MTG)};
    };
    auto read_b = [&](const Unit& un, float (&b)[CB][KT]) {
        const int chain = NB >= NGRP ? grp * G::NCH + un.c : 0;                      // chain of the flat frame list
        const float* xp = X + __mul24((chain * TP + g) * V + un.v, cs) + j;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int ks = 0; ks < KT; ++ks) b[cb][ks] = xp[ks * 4 * V * cs + cb * 16];
    };
    auto write_y = [&](int r, const Unit& un, const f32x4 (&acc)[CB]) {
        if (wave + r * NWAVES >= UNITS) return;
        const int fl = un.c * G::FGC + un.m * 16 + 4 * g;             // first of the lane's 4 output frames, group-local
        float* yp = Y + __mul24(fl * V + un.v, csy) + j;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4)
                if (G::FGC % 16 == 0 || un.m * 16 + 4 * g + r4 < G::FGC) yp[r4 * V * csy + cb * 16] = acc[cb][r4];
    };
    Unit u0 = unit_of(0), ur[NR];
    float b0[CB][KT], br[NR][CB][KT];
    read_b(u0, b0);
    if constexpr (PER > 1) {
#pragma unroll
        for (int i = 0; i < PER - 1; ++i) { ur[i] = unit_of(i + 1); read_b(ur[i], br[i]); }
    }
    f32x4 acc0[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc0[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KT; ++ks)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc0[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(first[ks], b0[cb][ks], acc0[cb], 0, 0, 0);
    write_y(0, u0, acc0);
    if constexpr (PER > 1) {
        f32x4 accr[NR][CB];
#pragma unroll
        for (int i = 0; i < PER - 1; ++i)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) accr[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int i = 0; i < PER - 1; ++i)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) accr[i][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i][ks], br[i][cb][ks], accr[i][cb], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < PER - 1; ++i) write_y(i + 1, ur[i], accr[i]);
    }
}
template <int V, int TP, int NB, int NGRP>
__device__ __forceinline__ void tl_joint_fetch(float (&aop)[(V + 15) / 16][(V + 3) / 4], const float* __restrict__ af, int fl, int lane, int grp) {
    using G = TlGroups<TP, NB, NGRP>;
    constexpr int KS = (V + 3) / 4, MT = (V + 15) / 16;
    if (fl >= G::FG) fl = G::FG - 1;
    const int q = (grp * G::FG + fl) % TP;               // the frame in its chain (the coefficient tables are per chain)
    gfloat* af_g = as_global(af);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) aop[mt][ks] = af_g[((q * MT + mt) * KS + ks) * 64 + lane];
}
template <int CINV, int V, int TP, int NB, int NGRP>
__device__ __forceinline__ void tl_joint_mix(float* __restrict__ YZ, int cs, const float* __restrict__ af, int wave, int lane, int grp,
                                             const float (&first)[(V + 15) / 16][(V + 3) / 4]) {
    using G = TlGroups<TP, NB, NGRP>;
    constexpr int CB = CINV / 16, KS = (V + 3) / 4, KP = 2 * (KS / 2), MT = (V + 15) / 16;
    constexpr bool J16 = V == 17;
    constexpr int MTM = J16 ? 1 : MT;
    constexpr int UNITS = G::FG, PER = (UNITS + NWAVES - 1) / NWAVES, NR = PER > 1 ? PER - 1 : 1;
    const int j = lane & 15, g = lane >> 4;
    const int vp = 4 * (g & 1) + (g >> 1);              // mix_vmap: joint of this lane group in a paired k-step
    float ar[NR][MT][KS];
    if constexpr (PER > 1) {
#pragma unroll
        for (int i = 0; i < PER - 1; ++i) tl_joint_fetch<V, TP, NB, NGRP>(ar[i], af, wave + (i + 1) * NWAVES, lane, grp);
    }
    auto run = [&](auto r0c, auto nic, const auto& aop) {
        constexpr int r0 = decltype(r0c)::value, NI = decltype(nic)::value;
        float y[NI][CB][KS];
        f32x4 acc[NI][CB][MTM];
        float part[NI][CB];
        float* base[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f0 = wave + (r0 + i) * NWAVES, fl = f0 < UNITS ? f0 : UNITS - 1;       // frame of the group
            base[i] = YZ + __mul24(fl * V, cs) + j;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                part[i][cb] = 0.f;
#pragma unroll
                for (int mt = 0; mt < MTM; ++mt) acc[i][cb][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    y[i][cb][ks] = base[i][(ks < KP ? 8 * (ks >> 1) + 2 * (ks & 1) + vp : 4 * KP + g) * cs + cb * 16];
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
                    for (int mt = 0; mt < MTM; ++mt)
                        acc[i][cb][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop[i][mt][ks], y[i][cb][ks], acc[i][cb][mt], 0, 0, 0);
                    if constexpr (J16) part[i][cb] = fmaf(aop[i][1][ks], y[i][cb][ks], part[i][cb]);
                }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (wave + (r0 + i) * NWAVES >= UNITS) continue;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
#pragma unroll
                for (int mt = 0; mt < MTM; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (mt * 16 + 4 * g + r < V) base[i][(mt * 16 + 4 * g + r) * cs + cb * 16] = acc[i][cb][mt][r];
                if constexpr (J16) {      // joint 16: the four lane groups' partial sums (see mix_long)
                    const unsigned pu = __float_as_uint(part[i][cb]);
                    const auto h = __builtin_amdgcn_permlane32_swap(pu, pu, false, false);
                    const unsigned v2 = __float_as_uint(__uint_as_float(h[0]) + __uint_as_float(h[1]));
                    const auto f = __builtin_amdgcn_permlane16_swap(v2, v2, false, false);
                    if (g == 0) base[i][16 * cs + cb * 16] = __uint_as_float(f[0]) + __uint_as_float(f[1]);
                }
            }
        }
    };
    const float (&f1)[1][MT][KS] = reinterpret_cast<const float (&)[1][MT][KS]>(first);
    run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, f1);
    if constexpr (PER > 1) run(std::integral_constant<int, 1>{}, std::integral_constant<int, PER - 1>{}, ar);
}

// partial channel GEMM of a layer of the slab-tiled kernel: acc[i] += A[O1 ..] . B1 (+ A[O2 ..] . B2) over this wave's n-tiles
// (Tiling<MT, NT>), K = 16 KQ1 (+ 16 KQ2) channels of the LDS operands b1 / b2 ([col][ch]); two tiles' MFMA chains in flight
// with their B fragments one read ahead, as in gemm_tiles.  The accumulators stay with the caller: a layer with 64 or 128
// input channels sums its 32-channel halves into them and runs its epilogue once.
template <int MT, int NT, int KQ1, int KQ2, int O1, int O2, int NA>
__device__ __forceinline__ void gemm_part(const float4 (&a)[NA], const float* __restrict__ b1, int cs1, const float* __restrict__ b2,
                                          int cs2, int wave, int lane, f32x4 (&acc)[Tiling<MT, NT>::MAXN]) {
    constexpr int NG = Tiling<MT, NT>::NG, MAXN = Tiling<MT, NT>::MAXN, KQ = KQ1 + KQ2;
    const int ng = MT > NWAVES ? 0 : wave / MT;
    const int j = lane & 15, g = lane >> 4;
    const float* const p1b = b1 + __mul24(ng * 16 + j, cs1) + 4 * g;
    const float* const p2b = b2 + __mul24(ng * 16 + j, cs2) + 4 * g;
    constexpr int NP = (MAXN + 1) / 2;
    // first B fragment (k-group 0) of tile slot i -- read a pair ahead: the last k-group of a pair fetches the next pair's
    auto rd0 = [&](int i) { return *reinterpret_cast<const float4*>((KQ1 > 0 ? p1b + i * NG * 16 * cs1 : p2b + i * NG * 16 * cs2)); };
    float4 nxt[2];
    auto prime = [&](auto pp) {
        constexpr int p = decltype(pp)::value;
        if constexpr (p < NP) {
            constexpr int i0 = 2 * p, i1 = i0 + 1 < MAXN ? i0 + 1 : i0;
            if (ng + i0 * NG < NT) nxt[0] = rd0(i0);
            if (i1 != i0 && ng + i1 * NG < NT) nxt[1] = rd0(i1);
        }
    };
    auto chain = [&](auto nn, auto pp, auto ia, auto ib) {
        constexpr int N = decltype(nn)::value, p = decltype(pp)::value, i0 = decltype(ia)::value, i1 = decltype(ib)::value;
        const float* p1[2] = {p1b + i0 * NG * 16 * cs1, p1b + i1 * NG * 16 * cs1};
        const float* p2[2] = {p2b + i0 * NG * 16 * cs2, p2b + i1 * NG * 16 * cs2};
        auto rd = [&](int h, auto kk) {
            constexpr int kq = decltype(kk)::value;
            return *reinterpret_cast<const float4*>(kq < KQ1 ? p1[h] + kq * 16 : p2[h] + (kq - KQ1) * 16);
        };
        static_for<KQ>([&](auto kk) {
            constexpr int kq = decltype(kk)::value;
            const float4 w = a[kq < KQ1 ? O1 + kq : O2 + kq - KQ1];
            float4 u[2];
#pragma unroll
            for (int h = 0; h < N; ++h) u[h] = nxt[h];
            if constexpr (kq + 1 < KQ) {
#pragma unroll
                for (int h = 0; h < N; ++h) nxt[h] = rd(h, std::integral_constant<int, kq + 1>{});
            } else {
                prime(std::integral_constant<int, p + 1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4& c0 = acc[i0];
            f32x4& c1 = acc[i1];
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, u[0].x, c0, 0, 0, 0);
            if constexpr (N == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, u[1].x, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, u[0].y, c0, 0, 0, 0);
            if constexpr (N == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, u[1].y, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, u[0].z, c0, 0, 0, 0);
            if constexpr (N == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, u[1].z, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, u[0].w, c0, 0, 0, 0);
            if constexpr (N == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, u[1].w, c1, 0, 0, 0);
        });
    };
    prime(std::integral_constant<int, 0>{});
    static_for<NP>([&](auto pp) {
        constexpr int i0 = 2 * decltype(pp)::value, i1 = i0 + 1 < MAXN ? i0 + 1 : i0;
        using I0 = std::integral_constant<int, i0>;
        using I1 = std::integral_constant<int, i1>;
        if (i1 != i0 && ng + i1 * NG < NT) chain(std::integral_constant<int, 2>{}, pp, I0{}, I1{});
        else if (ng + i0 * NG < NT) chain(std::integral_constant<int, 1>{}, pp, I0{}, I0{});
    });
}

// profile builds (tools/tiled_stage_profile.py): lane 0 of waves 0 and 7 of workgroup 0 add the cycles since their previous
// mark to slot 2048 (+ 64 for wave 7) + id of the profile buffer
#ifdef MCD_PROFILE
#define TLMARK(id) do { if (tl_prof) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    atomicAdd(P.prof + 2048 + (tid0 ? 64 : 0) + (id), t_ - tl_last); tl_last = t_; } } while (0)
#else
#define TLMARK(id) do { } while (0)
#endif
template <int TP, int NB>
__global__ __launch_bounds__(NTHREADS, 2) void score_tiled_kernel(const ScoreParams P, const FrameMaps M, const TiledNet N, int T,
                                                                  float* __restrict__ slabs) {
    constexpr int TF = TP * NB;
    constexpr int R17 = TF * 17, R12 = TF * 12, R10 = TF * 10;
    constexpr int TL_FC = tl_fc(TF), NFC = TF / TL_FC;
    static_assert(TP % TL_FC == 0, "a GEMM chunk lies inside one chain (its embedding rows are the chain's)");
    constexpr int EMBS = EMB_TOTAL + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // LDS: work region RA (mix: 32 channels of all frames; GEMM: z chunk + x chunk), chain state XT[col][4] (+ pad), tables
    constexpr int RA_F = tl_ra_floats(TF);
    float* const RA = smem;
    float* const XT = RA + RA_F;                    // [R17 + 16][4]
    float* const EMB = XT + (R17 + 16) * 4;         // [NB][EMB_TOTAL + 4]
    float* const SE = EMB + NB * EMBS;              // [NB][16]
    float* const ZN = SE + NB * EDIM;               // [R17][2]  this step's noise
    float* const ZO = ZN + R17 * C0;                // [R17][2]  layer 10's mixed output
    float* const P4 = ZO + R17 * C0;                // [R17 + 16][4]  layer 10's W-first product (+ zero pad rows: its mix reads 16-channel blocks)
    float* const RED = P4 + (R17 + 16) * 4;         // [NTHREADS]
    const int tid0 = threadIdx.x;
    int tid = tid0, lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* wb = P.wbuf;
#ifdef MCD_PROFILE
    const bool tl_prof = P.prof && blockIdx.x == 0 && (tid0 == 0 || tid0 == NTHREADS - 64);
    unsigned long long tl_last = __builtin_readcyclecounter();
#endif
    float* slab = slabs + (size_t)blockIdx.x * tl_slab_floats(TF);
    const int Tx = P.n_corrupt, K = P.ns > 2 ? P.ns - 1 : 1, per = C0 * Tx * 17;
    for (int u = tid; u < (int)tl_slab_floats(TF); u += NTHREADS) slab[u] = 0.f;      // pad rows / pad frames: finite values
    for (int u = tid; u < (R17 + 16) * 4; u += NTHREADS) XT[u] = 0.f;
    if (tid < 64) P4[R17 * 4 + tid] = 0.f;
    for (int u = tid; u < RA_F; u += NTHREADS) RA[u] = 0.f;           // pad rows meet zero coefficients: they must be finite

    for (long long grp = blockIdx.x; grp * NB < P.n_chains; grp += gridDim.x) {
        // chain i of the group (the last group of an odd count runs its last chain twice and writes it once)
        auto chain_of = [&](int i) { const long long c = grp * NB + i; return c < P.n_chains ? c : P.n_chains - 1; };
        auto b_of = [&](int i) { return (int)(chain_of(i) / P.S); };
        auto s_of = [&](int i) { return (int)(chain_of(i) % P.S); };
        auto fixed_of = [&](int i) { return (unsigned)(P.win_mask ? P.win_mask[b_of(i)] : P.fixed_mask); };
        auto tx_of = [&](unsigned fixed, int t) { return P.win_mask ? __popc(~fixed & ((1u << t) - 1u)) : M.tx_of[t]; };
        auto src_of = [&](int t) { return P.win_mask ? t : M.src_frame[t]; };
        __syncthreads();
        for (int u = tid; u < NB * T * 17; u += NTHREADS) {
            const int i = u / (T * 17), t = (u / 17) % T, v = u % 17;
            const int b = b_of(i), s = s_of(i);
            const unsigned fixed = fixed_of(i);
#pragma unroll
            for (int c = 0; c < C0; ++c) {
                float x;
                if ((fixed >> t) & 1u) x = load_coord(P.dv, b, c, src_of(t), v, P.seg_len);
                else {
                    const int e = (c * Tx + tx_of(fixed, t)) * 17 + v;
                    x = P.noise ? P.noise[((size_t)(s * K + 0) * P.B + b) * per + e]
                                : philox_normal(P.seed, (unsigned)e, 0u, (unsigned)s, (unsigned)(P.first_window + b));
                }
                XT[((i * TP + t) * 17 + v) * 4 + c] = x;
            }
        }
        for (int sidx = P.ns - 1; sidx >= 1; --sidx) {
            const float* srow = P.step_table + sidx * (4 + EDIM);
            // opaque per step (see score_kernel): otherwise every per-lane address of every stage is hoisted out of the step
            // loop as loop-invariant and the hundreds of resulting registers are spilled
            tid = tid0;
            asm volatile("" : "+v"(tid));
            lane = tid & 63;
            wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            wb = P.wbuf;
            asm volatile("" : "+s"(wb));
            float* sl = slab;
            asm volatile("" : "+s"(sl));
            float* const A0 = sl;
            float* const A1 = A0 + (R10 + 16) * 132;
            float* const D1 = A1 + (R10 + 16) * 132;
            float* const D2 = D1 + (R17 + 16) * 36;
            __syncthreads();
            if (tid < NB * EDIM) {
                float e = srow[4 + tid % EDIM];
                if (P.cond_emb) e += P.cond_emb[(size_t)b_of(tid / EDIM) * EDIM + tid % EDIM];
                SE[tid] = e / (1.f + expf(-e));
            }
            if (sidx > 1) {      // this step's noise, one thread per (frame, joint pair): the same Philox keys as score_kernel
                const int k = P.ns - sidx;
                for (int gi = tid; gi < NB * T * 9; gi += NTHREADS) {
                    const int i = gi / (T * 9), t = (gi / 9) % T, v0 = (gi % 9) * 2;
                    const int b = b_of(i), s = s_of(i);
                    const unsigned fixed = fixed_of(i);
                    float z[4] = {0.f, 0.f, 0.f, 0.f};
                    if (!((fixed >> t) & 1u)) {
                        const int tx = tx_of(fixed, t);
                        if (P.noise) {
                            const float* zp = P.noise + ((size_t)(s * K + k) * P.B + b) * per + tx * 17 + v0;
                            z[0] = zp[0]; z[1] = zp[Tx * 17];
                            if (v0 + 1 < 17) { z[2] = zp[1]; z[3] = zp[Tx * 17 + 1]; }
                        } else {
                            philox_normal4(P.seed, (unsigned)(tx * 9 + (v0 >> 1)), (unsigned)k, (unsigned)s, (unsigned)(P.first_window + b), z);
                        }
                    }
                    float* zo = ZN + ((i * TP + t) * 17 + v0) * C0;
                    zo[0] = z[0]; zo[1] = z[1];
                    if (v0 + 1 < 17) { zo[2] = z[2]; zo[3] = z[3]; }
                }
            }
            __syncthreads();
            for (int u = tid; u < NB * EMB_TOTAL; u += NTHREADS) {
                const int i = u / EMB_TOTAL, o = u % EMB_TOTAL;
                const float* we = wb + N.we + o * EDIM;
                float a = wb[N.be + o];
#pragma unroll
                for (int k = 0; k < EDIM; ++k) a = fmaf(we[k], SE[i * EDIM + k], a);
                EMB[i * EMBS + o] = a;
            }
            __syncthreads();

            // ---- one mix-first ST-GCN layer: xin (slab or XT) -> xout (slab).  Per 32-channel half of the input: X of ALL frames
            // -> LDS, mix -> z in LDS (never in the slab), the half's share of the channel GEMM into register accumulators
            // (z . W_t and x . W_r / + x); epilogue -> slab after the last half.  The layers at 17 joints (one half) take their
            // frames in two groups, z holding one group at a time.
            // RSI >= 0: the layer's input is joint resampler RSI applied to `xin` (+ `skip`): each 32-channel part of X is built in
            // LDS from the resampler's input rows, chunk of frames by chunk -- the resampled tensor never exists in the slab
            auto layer = [&](auto lc, auto rsc, const float* xin, bool xin_lds, float* xout, const float* skip) {
                constexpr int L = decltype(lc)::value, RSI = decltype(rsc)::value;
                constexpr int VIN = RSI == 0 ? 17 : RSI == 2 ? 10 : 12;      // joints of the resampler's input (down1, down2, up3, up2)
                // LDS hand-over between two plain layers at the same joint count (3 -> 4, 5 -> 6, 7 -> 8): the first layer's epilogue
                // writes output channels 0 .. 31 straight into the X region -- the second layer's first 32-channel part, which then
                // never touches the slab (its load was the one nothing could hide: stores -> barrier -> loads, 3 - 4 us a layer)
                constexpr bool HO = L == 3 || L == 5 || L == 7, HI = L == 4 || L == 6 || L == 8;
                // ... and between the layers at 17 joints (0 -> 1 -> 2, 9 -> 10; one 32-channel part, two frame groups): the whole output
                // goes to the X region, group 0's once group 1's time mix has read the old rows (its accumulators wait in registers)
                constexpr bool HO17 = L == 0 || L == 1 || L == 9, HI17 = L == 1 || L == 2;
                // ... and into a fused resampler (4 -> down2 -> 5, 6 -> up3 -> 7): output channels 0 .. 31 go where the next layer's
                // resampler takes its input chunks from (that layer's z region), all frames at once
                constexpr bool HOR = L == 4 || L == 6, HIR = L == 5 || L == 7;
                constexpr int TNEXT = (TF * (L == 4 ? 10 : L == 8 ? 17 : 12) + 16) * 36;     // offset of the next layer's z region
                // ... or only the resampler's FIRST chunk where all of it does not fit (2 -> down1 -> 3, 8 -> up2 -> 9); layer 2 keeps
                // group 0's accumulators until both groups are through (the chunk's place is still its own X rows before)
                constexpr bool HOC = L == 2 || L == 8, HIC = L == 3 || L == 9;
                constexpr LDesc D = layer_desc(L);
                constexpr int CIN = D.cin, COUT = D.cout, V = D.V, CSI = cs_of(CIN), CSO = cs_of(COUT);
                constexpr bool RES = D.res != 0;
                constexpr int CINV = CIN >= 32 ? 32 : 16, NH = CIN / CINV, CSZ = cs_of(CINV);
                constexpr int CSV = L == 0 ? 4 : L == 1 ? 36 : CSZ;    // layer 0 reads the chain state in place (see score_kernel); layer 1's
                                                                       // X is layer 0's output, written with the 32-channel row stride
                constexpr int ROWS = TF * V, FS = V == 17 ? 2 : 1, ROWSG = ROWS / FS;
                static_assert(FS == 1 || NH == 1, "frame groups and channel halves are not combined");
                static_assert(!HOR || TNEXT + TF * V * 36 <= RA_F, "the handed-over resampler input fits behind the next layer's X");
                static_assert(!HOC || TNEXT + TL_FC * V * 36 <= RA_F, "the handed-over first chunk fits behind the next layer's X");
                static_assert(COUT % 16 == 0 && ROWSG % (NB > 1 ? 1 : 1) == 0, "");
                constexpr int MT = COUT / 16, NT = ceil16(ROWSG) / 16, KH = CINV / 16;
                using TI = Tiling<MT, NT>;
                static_assert(FS == tl_ngrp(V), "the packed time-mix tiles follow the frame groups");
                // (thread / wave ids opaque per LAYER: the per-lane addresses of a layer's copies and tiles are invariant across
                // its loops, and hoisted to the top of the pass for all eleven layers at once they spill)
                int tid = tid0;
                asm volatile("" : "+v"(tid));
                const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                float* const XA = RA;                                          // [ROWS + 16][CSV]
                float* const ZA = RA + (ROWS + 16) * (L <= 1 ? 36 : CSZ);      // [ROWSG + 16][CSZ]  (layer 0: behind the next layer's X)
                TlStage<ROWS, CINV> sx;                // plain input: a 32-channel part of all frames; resampled input: the skip rows
                constexpr int IR = TL_FC * VIN, OR = TL_FC * V;
                TlStage<RSI >= 0 ? IR : 1, 32> si;     // resampled input: a chunk of the resampler's input rows
                RsCoef<32, VIN, V, TL_FC, 1, false> rc;
                if constexpr (RSI >= 0) {
                    static_assert(!HIR || NH >= 2, "");
                    static_assert(!HIC || (NH == 1 && NFC == 2), "");
                    if constexpr (HIC) si.issue(tid, xin + (size_t)IR * CSI, CSI, 0);      // (chunk 0 is in the z region already)
                    else si.issue(tid, xin, CSI, HIR ? CINV : 0);   // (HIR: part 0 is in the z region already, all chunks of it)
                    rc.load(wb + N.rsw[RSI], wb + N.rsw[RSI] + ((V + 15) / 16) * ((VIN + 3) / 4) * 64, lane);
                    if (skip) sx.issue(tid, skip, CSI, 0);
                } else if (!xin_lds && !HI17) {
                    static_assert(!HI || (RSI < 0 && NH >= 2), "");
                    sx.issue(tid, xin, CSI, HI ? CINV : 0);
                }
                float tqa[TP / 4], aja[(V + 15) / 16][(V + 3) / 4];     // the first units' mix coefficients, a stage ahead
                tl_time_fetch<V, TP, NB, FS>(tqa, wb + N.tqm[L], wave, lane, 0);
                // weight fragments of the wave's m-tile: all of them up front, or (128 input channels) a quarter at a time
                constexpr bool AQ = NH > 2;
                constexpr int KQA = (CIN / 16) * (RES ? 2 : 1);
                const int mt = wave % MT, ng = MT > NWAVES ? 0 : wave / MT, c0 = mt * 16 + 4 * (lane >> 4);
                LayerAfr<AQ ? 1 : KQA> A;
                float4 aq[AQ ? 2 * KH : 1];
                const float* wfr = wb + N.wp[L] + ((size_t)mt * KQA * 64 + lane) * 4;
                if constexpr (!AQ) {
                    LayerW lw;
                    lw.wp = N.wp[L]; lw.bias = N.bias[L];
                    A.template load<MT>(wb, lw, wave, lane);
                } else {
                    A.bcur = load_global4(wb + N.bias[L] + c0);
                }
                const float slope = N.slope[L], pinf = prelu_bound(slope);
                f32x4 acc[TI::MAXN];
                static_for<NH>([&](auto hh) {
                    constexpr int h = decltype(hh)::value;
                    const float* Xl = xin;
                    if constexpr (RSI >= 0) {
                        static_assert(RSI < 0 || (CINV == 32 && L != 0), "");
                        static_assert(RSI < 0 || IR <= ROWSG + 16, "the resampler's input chunk fits the z region");
                        float nosk[1] = {0.f};
                        if constexpr (HIR && h == 0) {
#pragma unroll
                            for (int fc = 0; fc < NFC; ++fc)
                                resample_stage<32, VIN, V, TL_FC, 1, false, false, true>(ZA + fc * IR * CSZ, CSZ, XA + fc * OR * CSV, CSV, rc, nosk, wave, lane);
                        } else {
#pragma unroll
                            for (int fc = 0; fc < NFC; ++fc) {
                                if (!(HIC && fc == 0)) {
                                    __syncthreads();      // (the previous stage / part / chunk is done with XA and the chunk region)
                                    si.commit(tid, ZA, CSZ);
                                    __syncthreads();
                                    if (!HIC && fc + 1 < NFC) si.issue(tid, xin + (size_t)(fc + 1) * IR * CSI, CSI, h * CINV);
                                    else if constexpr (h + 1 < NH) si.issue(tid, xin, CSI, (h + 1) * CINV);
                                }
                                resample_stage<32, VIN, V, TL_FC, 1, false, false, true>(ZA, CSZ, XA + fc * OR * CSV, CSV, rc, nosk, wave, lane);
                            }
                        }
                        if (skip) {                       // + the U-Net skip (d2 / d1), this part's channels
                            __syncthreads();
#pragma unroll
                            for (int i = 0; i < decltype(sx)::N; ++i) {
                                const int u = tid + i * NTHREADS;
                                if (u < ROWS * 8) {
                                    float4* xp = reinterpret_cast<float4*>(XA + (u >> 3) * CSV + (u & 7) * 4);
                                    const float4 a = *xp, b = sx.v[i];
                                    *xp = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
                                }
                            }
                            if constexpr (h + 1 < NH) sx.issue(tid, skip, CSI, (h + 1) * CINV);
                        }
                        Xl = XA;
                    } else if (HI17) {
                        Xl = XA;                          // (the previous layer's epilogue left it there)
                    } else if (!xin_lds) {
                        if constexpr (!(HI && h == 0)) {  // (HI: part 0 is in XA already, part 1 on its way)
                            __syncthreads();              // (the previous stage / half is done with XA)
                            sx.commit(tid, XA, CSV);
                            if constexpr (h + 1 < NH) sx.issue(tid, xin, CSI, (h + 1) * CINV);
                        }
                        Xl = XA;
                    }
                    if constexpr (AQ) {
                        static_assert(!AQ || RES, "");
#pragma unroll
                        for (int k = 0; k < KH; ++k) {
                            aq[k] = load_global4(wfr + (h * KH + k) * 256);
                            aq[KH + k] = load_global4(wfr + (CIN / 16 + h * KH + k) * 256);
                        }
                    }
                    __syncthreads();
                    TLMARK(4 * L);
                    auto gemm_fg = [&](int fg, f32x4 (&ac)[TI::MAXN]) {
                        const float* xg = Xl + fg * ROWSG * CSV;
                        if (h == 0) {
#pragma unroll
                            for (int i = 0; i < TI::MAXN; ++i) ac[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                        if constexpr (AQ) gemm_part<MT, NT, KH, KH, 0, KH>(aq, ZA, CSZ, xg, CSV, wave, lane, ac);
                        else if constexpr (RES) gemm_part<MT, NT, KH, KH, h * KH, CIN / 16 + h * KH>(A.a, ZA, CSZ, xg, CSV, wave, lane, ac);
                        else gemm_part<MT, NT, KH, 0, h * KH, 0>(A.a, ZA, CSZ, xg, CSV, wave, lane, ac);
                        if constexpr (!RES) {             // identity residual: the tile's own 4 channels of x, when they lie in this half
                            if ((mt * 16) / CINV == h) {
                                const float* xr = xg + __mul24(ng * 16 + (lane & 15), CSV) + c0 - h * CINV;
                                static_for<TI::MAXN>([&](auto ii) {
                                    constexpr int i = decltype(ii)::value;
                                    if (ng + i * TI::NG < NT) {
                                        const float4 r = *reinterpret_cast<const float4*>(xr + i * TI::NG * 16 * CSV);
                                        ac[i] += f32x4{r.x, r.y, r.z, r.w};
                                    }
                                });
                            }
                        }
                    };
                    // epilogue: bias, PReLU, embedding -> slab, or -> the X region (the next layer's input, row stride 36)
                    auto epi_fg = [&](int fg, const f32x4 (&ac)[TI::MAXN]) {
                        const float4 bcur = A.bcur;
                        static_for<TI::MAXN>([&](auto ii) {
                            constexpr int i = decltype(ii)::value;
                            const int col = ng * 16 + (lane & 15) + i * TI::NG * 16;
                            if (ng + i * TI::NG < NT && col < ROWSG) {
                                const int gcol = fg * ROWSG + col;
                                const float4 e = *reinterpret_cast<const float4*>(EMB + (NB > 1 ? gcol / (TP * V) : 0) * EMBS + emb_off(L) + c0);
                                const f32x2 t0 = f32x2{ac[i][0] + bcur.x, ac[i][1] + bcur.y}, t1 = f32x2{ac[i][2] + bcur.z, ac[i][3] + bcur.w};
                                const f32x2 m0 = t0 * slope, m1 = t1 * slope;
                                const float4 o = make_float4(__builtin_amdgcn_fmed3f(t0[0], m0[0], pinf) + e.x, __builtin_amdgcn_fmed3f(t0[1], m0[1], pinf) + e.y,
                                                             __builtin_amdgcn_fmed3f(t1[0], m1[0], pinf) + e.z, __builtin_amdgcn_fmed3f(t1[1], m1[1], pinf) + e.w);
                                if (HO17 || (HO && mt < 2)) *reinterpret_cast<float4*>(XA + gcol * 36 + c0) = o;
                                else *reinterpret_cast<float4*>(xout + (size_t)gcol * CSO + c0) = o;      // (layer 4's is the skip d2 as well)
                                if (HOR && mt < 2) *reinterpret_cast<float4*>(RA + TNEXT + gcol * 36 + c0) = o;
                                if (HOC && gcol < TL_FC * V) *reinterpret_cast<float4*>(RA + TNEXT + gcol * 36 + c0) = o;
                            }
                        });
                    };
                    if constexpr (HO17 || L == 2) {
                        static_assert(!(HO17 || L == 2) || (FS == 2 && NH == 1 && COUT <= 32), "");
                        f32x4 acc0[TI::MAXN];
                        tl_joint_fetch<V, TP, NB, FS>(aja, wb + N.am[L], wave, lane, 0);
                        tl_time_mix<CINV, V, TP, NB, FS>(Xl, CSV, ZA, CSZ, wb + N.tqm[L], wave, lane, 0, tqa);
                        tl_time_fetch<V, TP, NB, FS>(tqa, wb + N.tqm[L], wave, lane, 1);
                        __syncthreads();
                        tl_joint_mix<CINV, V, TP, NB, FS>(ZA, CSZ, wb + N.am[L], wave, lane, 0, aja);
                        TLMARK(4 * L + 1);
                        __syncthreads();
                        TLMARK(4 * L + 2);
                        gemm_fg(0, acc0);
                        tl_joint_fetch<V, TP, NB, FS>(aja, wb + N.am[L], wave, lane, 1);
                        TLMARK(4 * L + 3);
                        __syncthreads();                  // (z is free)
                        tl_time_mix<CINV, V, TP, NB, FS>(Xl, CSV, ZA, CSZ, wb + N.tqm[L], wave, lane, 1, tqa);
                        __syncthreads();                  // (nobody reads the X rows of group 0 any more)
                        if constexpr (HO17) epi_fg(0, acc0);
                        tl_joint_mix<CINV, V, TP, NB, FS>(ZA, CSZ, wb + N.am[L], wave, lane, 1, aja);
                        TLMARK(4 * L + 1);
                        __syncthreads();
                        TLMARK(4 * L + 2);
                        gemm_fg(1, acc);
                        __syncthreads();                  // (... nor those of group 1)
                        if constexpr (!HO17) epi_fg(0, acc0);
                        epi_fg(1, acc);
                        TLMARK(4 * L + 3);
                    } else {
#pragma unroll
                        for (int fg = 0; fg < FS; ++fg) {
                            tl_joint_fetch<V, TP, NB, FS>(aja, wb + N.am[L], wave, lane, fg);
                            tl_time_mix<CINV, V, TP, NB, FS>(Xl, CSV, ZA, CSZ, wb + N.tqm[L], wave, lane, fg, tqa);
                            if (fg + 1 < FS || h + 1 < NH) tl_time_fetch<V, TP, NB, FS>(tqa, wb + N.tqm[L], wave, lane, fg + 1 < FS ? fg + 1 : 0);
                            __syncthreads();
                            tl_joint_mix<CINV, V, TP, NB, FS>(ZA, CSZ, wb + N.am[L], wave, lane, fg, aja);
                            TLMARK(4 * L + 1);
                            __syncthreads();
                            TLMARK(4 * L + 2);
                            gemm_fg(fg, acc);
                            if constexpr (h == NH - 1) {
                                static_assert(!(HO || HOR) || (FS == 1 && CSV == 36), "");
                                if constexpr (HO || HOR || L == 8) __syncthreads();    // (every wave is done with XA / z: the m-tiles 0, 1 go there)
                                epi_fg(fg, acc);
                            }
                            if (fg + 1 < FS) __syncthreads();  // (the next group's mix overwrites z)
                            TLMARK(4 * L + 3);
                        }
                    }
                });
                __syncthreads();
            };
#define TL_C(x) std::integral_constant<int, x>{}
            TLMARK(60);                                                     // pass prologue (noise, embeddings)
#define TL_NORS std::integral_constant<int, -1>{}
            layer(TL_C(0), TL_NORS, XT, true, A0, nullptr);
            layer(TL_C(1), TL_NORS, A0, false, A1, nullptr);
            layer(TL_C(2), TL_NORS, A1, false, D1, nullptr);                // -> d1
            layer(TL_C(3), TL_C(0), D1, false, A0, nullptr);                // down1 on the way in
            layer(TL_C(4), TL_NORS, A0, false, D2, nullptr);                // -> d2
            layer(TL_C(5), TL_C(1), D2, false, A0, nullptr);                // down2 on the way in; 64 -> 128
            layer(TL_C(6), TL_NORS, A0, false, A1, nullptr);                // 128 -> 64, mix-first here (four 32-channel quarters)
            layer(TL_C(7), TL_C(2), A1, false, A0, D2);                     // up3 + d2 on the way in
            layer(TL_C(8), TL_NORS, A0, false, A1, nullptr);
            layer(TL_C(9), TL_C(3), A1, false, A0, D1);                     // up2 + d1 on the way in
            TLMARK(61);
            {   // ---- layer 10 (32 -> 2) W-first on plain FMAs: P4[col][r] = sum_k W4[r][k] X[col][k]  (P_t 0,1 ; P_r 2,3)
                int tid = tid0;
                asm volatile("" : "+v"(tid));
                const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                MixLongCoef<16, 17, TP, NB> mc10;      // (the mix's first coefficients: in flight behind the product)
                mc10.load(wb + N.tq[10], wb + N.am[10], wave, lane);
                const float* w4 = wb + N.wp[10];     // [4][32], read with wave-uniform addresses (scalar loads)
                for (int col = tid; col < R17; col += NTHREADS) {
                    const float* xp = RA + col * 36;          // layer 9's output, handed over in LDS
                    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 x = *reinterpret_cast<const float4*>(xp + 4 * q);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            a[r] = fmaf(w4[r * 32 + 4 * q + 0], x.x, a[r]); a[r] = fmaf(w4[r * 32 + 4 * q + 1], x.y, a[r]);
                            a[r] = fmaf(w4[r * 32 + 4 * q + 2], x.z, a[r]); a[r] = fmaf(w4[r * 32 + 4 * q + 3], x.w, a[r]);
                        }
                    }
                    *reinterpret_cast<float4*>(P4 + col * 4) = make_float4(a[0], a[1], a[2], a[3]);
                }
                __syncthreads();
                // its 2-channel mix (16-channel block view of P4: channels 2..15 are the next columns' values, never stored)
                mix_long<16, 17, TP, NB>(P4, 4, mc10, wb + N.tq[10], wb + N.am[10], wave, lane, ZeroInitL{},
                                     [&](int q, int w0, int c, auto v) {
                                         if (c < C0) {
                                             float* zp = ZO + ((q * 17 + w0)) * C0 + c;
                                             if constexpr (std::is_same_v<decltype(v), f32x4>) {
#pragma unroll
                                                 for (int r = 0; r < 4; ++r)
                                                     if (w0 + r < 17) zp[r * C0] = v[r];
                                             } else {
                                                 *zp = v;
                                             }
                                         }
                                     });
                __syncthreads();
                // eps = layer 10 + x; DDPM update of the frame each prediction drives (mocodad.py:172-178,829-838)
                const float slope10 = N.slope[10], ca = srow[0], cb = srow[1], csg = srow[2];
                const bool zadd = sidx > 1;
                constexpr int NIT = (C0 * TF * 17 + NTHREADS - 1) / NTHREADS;
                float xn[NIT];
                int dst[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int u = tid + it * NTHREADS;
                    dst[it] = -1; xn[it] = 0.f;
                    if (u < TF * 17 * C0) {
                        const int c = u % C0, col = u / C0, f = col / 17, i = f / TP, t = f % TP, v = col % 17;
                        const float l10 = prelu(ZO[u] + P4[col * 4 + C0 + c] + wb[N.bias[10] + c], slope10) + EMB[i * EMBS + emb_off(10) + c];
                        const float eps = l10 + XT[col * 4 + c];
                        const int k = t >= T ? -1 : P.win_mask ? (((fixed_of(i) >> t) & 1u) ? -1 : 0) : M.upd_of[t];
                        if (k >= 0) {
                            const int tp = P.win_mask ? t : M.pos_of[k];
                            const int colp = (i * TP + tp) * 17 + v;
                            xn[it] = ca * (XT[colp * 4 + c] - cb * eps) + csg * (zadd ? ZN[colp * C0 + c] : 0.f);
                            dst[it] = colp * 4 + c;
                        }
                    }
                }
                __syncthreads();
#pragma unroll
                for (int it = 0; it < NIT; ++it)
                    if (dst[it] >= 0) XT[dst[it]] = xn[it];
                TLMARK(54);                            // layer 10 + DDPM update
            }
        }
        __syncthreads();
        // ---- loss over the corrupt frames (mocodad.py:484)
        for (int i = 0; i < NB; ++i) {
            if (grp * NB + i >= P.n_chains) break;
            const long long chain = grp * NB + i;
            const int b = b_of(i), s = s_of(i);
            const unsigned fixed = fixed_of(i);
            float part = 0.f;
            for (int e = tid; e < per; e += NTHREADS) {
                const int c = e / (Tx * 17), tx = (e / 17) % Tx, v = e % 17;
                int tu = M.pos_of[tx];
                if (P.win_mask) { int cnt = 0; for (int t = 0; t < T; ++t) if (!((fixed >> t) & 1u)) { if (cnt == tx) tu = t; ++cnt; } }
                const float x0 = XT[((i * TP + tu) * 17 + v) * 4 + c];
                const float gt = load_coord(P.dv, b, c, src_of(tu), v, P.seg_len);
                part += loss_elem(x0, gt, P.loss_fn);
                if (P.pose_out) P.pose_out[(size_t)(b * P.S + s) * per + e] = x0;
            }
            RED[tid] = part;
            __syncthreads();
            for (int o = NTHREADS / 2; o > 0; o >>= 1) { if (tid < o) RED[tid] += RED[tid + o]; __syncthreads(); }
            if (tid == 0) P.loss_out[chain] = RED[0] / (float)per;
            __syncthreads();
        }
    }
}

// 'E_unet' condition encoder at any frame count (the U-Net's down path without embeddings + to_time_dim), same scratch scheme
struct GenCond { GLayer L[7]; int rs_w[2], rs_b[2], lw, lb; };
__global__ __launch_bounds__(GEN_THREADS) void cond_unet_generic_kernel(const float* wb, const GenCond N, const DataView dv, const FrameIdx fi,
                                                                        int seg_len, int T, int B, float* __restrict__ emb_out,
                                                                        float* __restrict__ scratch) {
    __shared__ float RED[GEN_THREADS];
    const int TV = T * 17, tid = threadIdx.x;
    float* slab = scratch + (size_t)blockIdx.x * GEN_SLAB * T;
    float* A = slab;
    float* Bb = A + GEN_BUF * T;
    float* Zb = Bb + GEN_BUF * T;
    float* D1 = Zb + GEN_BUF * T;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        for (int u = tid; u < C0 * TV; u += GEN_THREADS) {
            const int c = u / TV, t = (u % TV) / 17, v = u % 17;
            D1[u] = load_coord(dv, b, c, fi.idx[t], v, seg_len);
        }
        __syncthreads();
        g_layer(wb, N.L[0], T, D1, A, Zb, A, nullptr);
        g_layer(wb, N.L[1], T, A, Bb, Zb, Bb, nullptr);
        g_layer(wb, N.L[2], T, Bb, A, Zb, A, nullptr);
        g_resample(wb, N.rs_w[0], N.rs_b[0], 32, T, 17, 12, A, Bb, nullptr);
        g_layer(wb, N.L[3], T, Bb, A, Zb, A, nullptr);
        g_layer(wb, N.L[4], T, A, Bb, Zb, Bb, nullptr);
        g_resample(wb, N.rs_w[1], N.rs_b[1], 64, T, 12, 10, Bb, A, nullptr);
        g_layer(wb, N.L[5], T, A, Bb, Zb, Bb, nullptr);
        g_layer(wb, N.L[6], T, Bb, A, Zb, A, nullptr);            // -> A [6][T][10]
        const int F = CU_OUT * T * 10;
        for (int jo = 0; jo < EDIM; ++jo) {
            float a = 0.f;
            for (int k = tid; k < F; k += GEN_THREADS) a = fmaf(wb[N.lw + (size_t)jo * F + k], A[k], a);
            RED[tid] = a;
            __syncthreads();
            for (int o = GEN_THREADS / 2; o > 0; o >>= 1) { if (tid < o) RED[tid] += RED[tid + o]; __syncthreads(); }
            if (tid == 0) emb_out[(size_t)b * EDIM + jo] = RED[0] + wb[N.lb + jo];
            __syncthreads();
        }
    }
}

// scatter-max of window scores to frames (mocodad.py:392-393 + eval_utils.py:27-34); scores >= 0
__global__ void scatter_max_kernel(const float* __restrict__ scores, const int* __restrict__ frames,
                                   const int* __restrict__ row, long long n, int seg_len, int n_frames,
                                   float* __restrict__ out) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n * seg_len) return;
    const long long i = u / seg_len;
    const int f = frames[u] - 1;
    if (f < 0 || f >= n_frames) return;
    // non-negative floats order like their bit patterns
    atomicMax(reinterpret_cast<int*>(out + (size_t)row[i] * n_frames + f), __float_as_int(fmaxf(scores[i], 0.f)));
}


// ------------------------------------------------------------------------------------------------
// Frame-score assembly after the path (mocodad.py:362-425; eval_utils.py:27-34,100-106,133-149), on device, in float64
// like the reference's NumPy code.
//   frame_scatter_kernel: window score -> max over the windows covering each frame of its (transform, clip, person) row.
//   frame_scores_kernel : one workgroup per clip; for every transform: per person pad_scores, then
//                         mean_p + (max_p - min_p) of log1p over the persons present, HR-mask compaction, shift,
//                         gaussian_filter1d (scipy defaults: truncate 4 sigma, 'reflect'), accumulated over the transforms
//                         and divided by their number.
// Rows are dense: row = (transform * n_clips + clip) * P + person id; `used[row]` marks persons that have windows.
// ------------------------------------------------------------------------------------------------
struct FrameParams {
    const float* scores; const long long* trans; const long long* meta; const int* frames;
    const long long* clip_keys;     // (n_clips,) sorted (scene << 32 | clip)
    const int* clip_n;              // (n_clips,) frames of the clip = len(gt)
    const int* dst;                 // per clip F entries: position of the frame after the HR masks, -1 = dropped
    const int* out_len;             // (n_clips,) frames kept
    const long long* out_off;       // (n_clips,) offset of the clip in the concatenated output
    const double* gauss;            // (2 radius + 1,) normalised weights
    float* mat; int* used; double* out;
    long long n;
    int seg_len, n_clips, num_transform, P, F, pad, shift, radius;
};

__global__ void frame_scatter_kernel(const FrameParams Q) {
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= Q.n * Q.seg_len) return;
    const long long i = u / Q.seg_len;
    const long long tr = Q.trans[i];
    if (tr < 0 || tr >= Q.num_transform) return;
    const long long key = (Q.meta[i * 4 + 0] << 32) | (Q.meta[i * 4 + 1] & 0xffffffffll);
    int lo = 0, hi = Q.n_clips;                     // lower bound in the sorted clip keys
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (Q.clip_keys[mid] < key) lo = mid + 1; else hi = mid; }
    if (lo >= Q.n_clips || Q.clip_keys[lo] != key) return;        // a clip without a ground-truth file is not evaluated
    const long long person = Q.meta[i * 4 + 2];
    if (person < 0 || person >= Q.P) return;
    const int f = Q.frames[u] - 1;
    if (f < 0 || f >= Q.clip_n[lo]) return;
    const long long row = ((long long)tr * Q.n_clips + lo) * Q.P + person;
    Q.used[row] = 1;
    // non-negative floats order like their bit patterns (np.nanmax over the windows covering the frame; 0 = absent)
    atomicMax(reinterpret_cast<int*>(Q.mat + row * Q.F + f), __float_as_int(fmaxf(Q.scores[i], 0.f)));
}

__global__ __launch_bounds__(256) void frame_scores_kernel(const FrameParams Q) {
    extern __shared__ __attribute__((aligned(16))) double fsm[];
    const int ci = blockIdx.x, n = Q.clip_n[ci], m = Q.out_len[ci];
    double* cs = fsm;                 // [m] compacted clip score of the current transform
    double* acc = fsm + Q.F;          // [m] sum over the transforms
    const int* dst = Q.dst + (size_t)ci * Q.F;
    for (int j = threadIdx.x; j < m; j += blockDim.x) acc[j] = 0.0;
    for (int tr = 0; tr < Q.num_transform; ++tr) {
        const size_t row0 = ((size_t)tr * Q.n_clips + ci) * Q.P;
        __syncthreads();
        for (int f = threadIdx.x; f < n; f += blockDim.x) {
            double sum = 0.0, lmax = 0.0, lmin = 0.0;
            int cnt = 0;
            for (int p = 0; p < Q.P; ++p) {
                if (!Q.used[row0 + p]) continue;
                const float* r = Q.mat + (row0 + p) * Q.F;
                float v = r[f];
                if (Q.pad >= 0 && v != 0.f) {
                    // pad_scores (eval_utils.py:133-149): zero `pad` frames before and pad-1 frames after every interval of
                    // absence inside frames [0, n-2]; an interval touching frame 0 / frame n-2 is not extended on that side
                    bool z = false;
                    for (int d = 1; d <= Q.pad && !z; ++d) z = (f + d <= n - 2) && r[f + d] == 0.f;
                    if (!z) {
                        // backwards: for the last frame, the run of absence that ends at frame n-2 does not count
                        bool in_tail = (f == n - 1);
                        for (int d = 1; d <= Q.pad - 1 && f - d >= 0 && !z; ++d) {
                            const bool zero = r[f - d] == 0.f;
                            if (in_tail) { if (!zero) in_tail = false; }
                            else z = zero;
                        }
                    }
                    if (z) v = 0.f;
                }
                const double dv = (double)v, lg = log1p(dv);
                sum += dv;
                if (cnt == 0) { lmax = lg; lmin = lg; } else { lmax = fmax(lmax, lg); lmin = fmin(lmin, lg); }
                ++cnt;
            }
            const int j = dst[f];
            // a (transform, clip) block without any person: NaN (the reference fails on np.stack of an empty list; the host
            // wrapper turns the NaN into that error)
            if (j >= 0) cs[j] = cnt > 0 ? sum / (double)cnt + (lmax - lmin) : (double)NAN;
        }
        __syncthreads();
        // score_process (eval_utils.py:100-106): shift by `shift` frames (zeros enter), then correlate with the Gaussian
        // in scipy's symmetric form: in[c] w[c] + sum_{i=1..radius} (in[c-i] + in[c+i]) w[c-i], outermost pair first
        for (int j = threadIdx.x; j < m; j += blockDim.x) {
            auto at = [&](int k) -> double {          // shifted, 'reflect'-extended (d c b a | a b c d | d c b a)
                const int per = 2 * m;
                k %= per; if (k < 0) k += per;
                if (k >= m) k = per - 1 - k;
                return k >= Q.shift ? cs[k - Q.shift] : 0.0;
            };
            double t = at(j) * Q.gauss[Q.radius];
            for (int i = Q.radius; i >= 1; --i) t += (at(j - i) + at(j + i)) * Q.gauss[Q.radius - i];
            acc[j] += t;
        }
    }
    __syncthreads();
    double* o = Q.out + Q.out_off[ci];
    for (int j = threadIdx.x; j < m; j += blockDim.x) o[j] = acc[j] / (double)Q.num_transform;
}

// ================================================================================================
// host side
// ================================================================================================
thread_local std::string g_err;
int fail(int code, const std::string& m) { g_err = m; return code; }

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MCD_EDEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct TensorMap {
    std::unordered_map<std::string, std::pair<const float*, int64_t>> m;
    std::string missing;
    const float* get(const std::string& name, int64_t numel) {
        auto it = m.find(name);
        if (it == m.end()) { if (missing.empty()) missing = "missing tensor " + name; return nullptr; }
        if (it->second.second != numel) {
            if (missing.empty()) missing = "tensor " + name + " has " + std::to_string(it->second.second) + " elements, expected " + std::to_string(numel);
            return nullptr;
        }
        return it->second.first;
    }
    bool has(const std::string& name) const { return m.count(name) != 0; }
};

struct Folded { std::vector<double> w, b; };  // BN-folded 1x1 conv: w[cout][cin], b[cout]

// conv (cout,cin,1,1)+bias followed by eval BatchNorm2d (eps 1e-5): W' = s W, b' = s (b - mu) + beta
bool fold_conv_bn(TensorMap& tm, const std::string& conv, const std::string& bn, int cout, int cin, Folded& f) {
    const float* w = tm.get(conv + ".weight", (int64_t)cout * cin);
    const float* b = tm.get(conv + ".bias", cout);
    const float* g = tm.get(bn + ".weight", cout);
    const float* be = tm.get(bn + ".bias", cout);
    const float* mu = tm.get(bn + ".running_mean", cout);
    const float* var = tm.get(bn + ".running_var", cout);
    if (!w || !b || !g || !be || !mu || !var) return false;
    f.w.resize((size_t)cout * cin); f.b.resize(cout);
    for (int o = 0; o < cout; ++o) {
        const double s = (double)g[o] / sqrt((double)var[o] + 1e-5);
        for (int i = 0; i < cin; ++i) f.w[(size_t)o * cin + i] = s * (double)w[(size_t)o * cin + i];
        f.b[o] = s * ((double)b[o] - (double)mu[o]) + (double)be[o];
    }
    return true;
}

struct Builder {
    std::vector<float> buf;
    int alloc(size_t n) { size_t o = (buf.size() + 3) & ~size_t(3); buf.resize(o + n, 0.f); return (int)o; }
};

// Tq[q][v][t] = T[v][t][q]; A copied
bool pack_mix(TensorMap& tm, const std::string& p, int T, int V, Builder& B, int& tq, int& am) {
    const float* Tm = tm.get(p + ".gcn.T", (int64_t)V * T * T);
    const float* A = tm.get(p + ".gcn.A", (int64_t)T * V * V);
    if (!Tm || !A) return false;
    tq = B.alloc((size_t)T * V * T);
    for (int q = 0; q < T; ++q) for (int v = 0; v < V; ++v) for (int t = 0; t < T; ++t)
        B.buf[tq + (q * V + v) * T + t] = Tm[(v * T + t) * T + q];
    am = B.alloc((size_t)T * V * V);
    memcpy(&B.buf[am], A, sizeof(float) * T * V * V);
    return true;
}

// fragment-order coefficients for the MFMA mix (see mix_stage)
// (TP > T: the tables of a frame count padded to TP -- score_tiled_kernel -- with zero coefficients for the pad frames)
bool pack_mix_mfma(TensorMap& tm, const std::string& p, int T, int V, Builder& B, int& tqf, int& af, int TP = 0) {
    const float* Tm = tm.get(p + ".gcn.T", (int64_t)V * T * T);
    const float* A = tm.get(p + ".gcn.A", (int64_t)T * V * V);
    if (!Tm || !A) return false;
    if (TP < T) TP = T;
    const int KS = (V + 3) / 4, MT = (V + 15) / 16;
    const int NR = (KS * TP + 15) / 16;
    tqf = B.alloc((size_t)TP * NR * 64);
    af = B.alloc((size_t)TP * MT * KS * 64);
    for (int q = 0; q < T; ++q) for (int r = 0; r < NR; ++r) for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, g = lane >> 4, idx = r * 16 + i, s = idx / TP, t = idx % TP, v = mix_vmap(V, s, g);
        B.buf[tqf + (q * NR + r) * 64 + lane] = (idx < KS * TP && v < V && t < T) ? Tm[(v * T + t) * T + q] : 0.f;
    }
    for (int q = 0; q < T; ++q) for (int s = 0; s < KS; ++s) for (int lane = 0; lane < 64; ++lane) {
        const int j = lane & 15, g = lane >> 4, v = mix_vmap(V, s, g);
        for (int mt = 0; mt < MT; ++mt) {
            // m-tile 0: MFMA A fragment (output joint 16mt + j).  V = 17: the one joint beyond it is mixed on the VALU
            // (mix_stage), its coefficient A_q[v][16] replicated over the 16 lanes of the group
            const int w = (V == 17 && mt == 1) ? 16 : mt * 16 + j;
            B.buf[af + ((q * MT + mt) * KS + s) * 64 + lane] = (v < V && w < V) ? A[(q * V + v) * V + w] : 0.f;
        }
    }
    return true;
}

// time-mix coefficients of one layer as the A fragments of tl_time_mix: [joint v][frame tile of a chain][k-step][lane], lane
// (i, g) = gcn.T[v][t = 4 ks + g][q], q = row i of the tile (tiles follow the layer's frame groups, TlGroups)
int pack_time_mfma(const float* Tm, int T, int V, int TP, int NB, Builder& B) {
    const int ngrp = tl_ngrp(V), nch = NB >= ngrp ? NB / ngrp : 1, fgc = NB * TP / ngrp / nch;
    const int mtg = (fgc + 15) / 16, ntc = mtg * (TP / fgc), kt = TP / 4;
    const int off = B.alloc((size_t)V * ntc * kt * 64);
    for (int v = 0; v < V; ++v) for (int tile = 0; tile < ntc; ++tile) for (int ks = 0; ks < kt; ++ks) for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 15, g = lane >> 4, t = 4 * ks + g, r = (tile % mtg) * 16 + i, q = (tile / mtg) * fgc + r;
        B.buf[off + ((size_t)(v * ntc + tile) * kt + ks) * 64 + lane] = (r < fgc && q < T && t < T) ? Tm[((size_t)v * T + t) * T + q] : 0.f;
    }
    return off;
}

// MFMA A-operand fragment order of a logical [M][K] matrix (M, K multiples of 16) with the K permutation that lets
// one ds_read_b128 of the B operand feed four k-steps (see gemm_tiles): element e of lane (i, g) in group kq is
// W[16 mt + i][16 kq + 4 g + e]  (k-step e of the group covers channels {16 kq + 4 g + e : g = 0..3}).
// round-to-nearest-even float -> bf16 bits (what v_cvt_pk_bf16_f32 does for finite values)
static inline unsigned short bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_to_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
// split-bf16 fragments of gemm_tiles_bf3: [m-tile][K/32][hi, lo][lane] x 8 bf16 (4 floats), lane (row, g): k = 32 ch + 8 g + i
template <class F>
int pack_gemm_frags_bf3(Builder& B, int M, int K, F&& w) {
    const int MTn = M / 16, CH = K / 32;
    const int off = B.alloc((size_t)MTn * CH * 2 * 64 * 4);
    for (int mt = 0; mt < MTn; ++mt) for (int ch = 0; ch < CH; ++ch) for (int lane = 0; lane < 64; ++lane) {
        const int row = mt * 16 + (lane & 15), g = lane >> 4;
        unsigned short hi[8], lo[8];
        for (int i = 0; i < 8; ++i) {
            const float v = (float)w(row, ch * 32 + 8 * g + i);
            hi[i] = bf16_rne(v);
            lo[i] = bf16_rne(v - bf16_to_f(hi[i]));
        }
        memcpy(&B.buf[off + ((size_t)((mt * CH + ch) * 2 + 0) * 64 + lane) * 4], hi, 16);
        memcpy(&B.buf[off + ((size_t)((mt * CH + ch) * 2 + 1) * 64 + lane) * 4], lo, 16);
    }
    return off;
}
template <class F>
int pack_gemm_frags(Builder& B, int M, int K, F&& w) {
    const int MTn = M / 16, KQ = K / 16;
    const int off = B.alloc((size_t)MTn * KQ * 64 * 4);
    for (int mt = 0; mt < MTn; ++mt) for (int kq = 0; kq < KQ; ++kq) for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 4; ++e) {
            const int row = mt * 16 + (lane & 15), g = lane >> 4;
            B.buf[off + ((size_t)(mt * KQ + kq) * 64 + lane) * 4 + e] = (float)w(row, kq * 16 + 4 * g + e);
        }
    return off;
}

}  // namespace

struct mcd_weights {
    mcd_model_cfg_t cfg;
    int device;
    float* dbuf;
    size_t n_floats;
    CondW cond;
    bool has_cond;
    bool cond_fast;   // shipped condition-encoder architecture -> cond_fast_kernel
    bool cond_unet;   // 'E_unet' condition encoder -> cond_unet_kernel
    bool fast_unet;   // a specialised score_kernel<T,...> exists for cfg.t_unet (1 .. 12); otherwise the slab-tiled (13 .. 32) or the runtime-shape kernel
    TiledNet tiled;   // tables of score_tiled_kernel (12 < t_unet <= 32), frame count padded to tiled_tp
    int tiled_tp;     // 16, 24 or 32; 0 = none
    GenNet gen;       // plain (unpacked) folded weights of the U-Net for score_generic_kernel
    GenCond gcond;    // ... and of the 'E_unet' condition encoder
    int zero_row;     // offset (floats) of 32 zero words in dbuf: an all-zero step_table row for mcd_layer_forward
    int opt[MCD_OPT_COUNT];   // mcd_set_option values (plain ints: set before the calls they affect, like any other argument)
};

namespace {

// Raise a kernel's dynamic-LDS limit once per device.  `done` is the kernel's own device bitmask; two host threads racing
// here both make the (idempotent) call, nobody launches before it has been made on its device.
int ensure_lds_limit(const void* fn, size_t bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return MCD_OK;
    HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return MCD_OK;
}
#define LDS_LIMIT(kernel_expr, bytes) do { static std::atomic<unsigned long long> done_{0}; \
    int rc_ = ensure_lds_limit(reinterpret_cast<const void*>(kernel_expr), (bytes), done_); if (rc_ != MCD_OK) return rc_; } while (0)

// Workgroup slots of the device for a kernel (resident workgroups per CU x CUs), asked once per (kernel, device).
int wg_slots(const void* fn, size_t lds, std::atomic<int> (&cache)[64]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, NTHREADS, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    v = per_cu * cus;
    cache[dev].store(v, std::memory_order_relaxed);
    return v;
}
// How a scoring call is cut into workgroups.  A workgroup owns NB windows and runs S / split of their samples in sequence.
// split = 1 (window-major) lets the condition encoder and the aggregation run inside the workgroup -- ONE launch per call,
// no workspace -- split = S (chain-major) is one trajectory per workgroup with the encoder and the aggregation as their own
// small launches.
int choose_split(int n_groups, int S, int slots) {
    // estimated makespan in units of one trajectory: rounds of workgroups x trajectories per workgroup; chain-major pays its
    // two extra launches (~0.05 trajectories).  (Round 2 charged window-major 2 %: what it lost was the tail of the YOUNGER
    // of the two co-resident workgroups, which the alternating wave priority of score_kernel removes -- the one-launch form is
    // now the faster one at equal rounds, profiles/r03s_prio_slices.txt.)
    auto rounds = [&](long long wgs) { return (double)((wgs + slots - 1) / slots); };
    const double window_major = rounds(n_groups) * S;
    const double chain_major = rounds((long long)n_groups * S) + 0.05;
    return S > 1 && chain_major < window_major ? S : 1;
}

template <int T, int NB, int MINW, bool BF3 = false, bool LT = false>
int launch_score_t(ScoreParams& P, hipStream_t st, bool* fused) {
    using PL = Plan<T, NB>;
    LDS_LIMIT((&score_kernel<T, NB, MINW, BF3, LT>), PL::BYTES);
    static std::atomic<int> slots_cache[64];
    const int groups = (P.B + NB - 1) / NB;
    const int slots = wg_slots(reinterpret_cast<const void*>(&score_kernel<T, NB, MINW, BF3, LT>), PL::BYTES, slots_cache);
    if (P.plan_only || P.mode != 0) {
        P.split = P.mode == 0 ? choose_split(groups, P.S, slots) : 1;
        if (P.mode == 0 && P.force_split > 0) P.split = P.force_split < P.S ? P.force_split : P.S;
        if (P.plan_only) return MCD_OK;
    }
    // the in-kernel condition encoder / aggregation need the workgroup to see all samples of its windows (and S <= 64)
    const bool whole = P.split == 1 && P.mode == 0;
    if (!(whole && P.S <= 64)) P.loss_agg = nullptr;
    if (fused) *fused = P.loss_agg != nullptr;
    if (P.loss_agg && P.loss_out_optional) P.loss_out = nullptr;
    if (P.mode == 0 && !P.loss_agg && !P.loss_out) return fail(MCD_EINVAL, "workspace required (mcd_score_workspace_bytes): per-sample losses of an unfused aggregation");
    P.prio_shift = 0;
    if (MINW >= 4 && P.mode == 0 && P.phase != -1) {
        // priority time slice of the co-resident workgroups (see score_kernel): about 1/6 of the launch's expected duration --
        // rounds of workgroups x trajectories per workgroup x passes x ~7.5 us per (chain, frame) of a pass
        const double rounds = (double)(((long long)groups * P.split + slots - 1) / slots);
        const double traj = (double)((P.S + P.split - 1) / P.split);
        const double ticks = rounds * traj * (double)(P.ns > 2 ? P.ns - 1 : 1) * 7.5 * NB * T * 100.0;
        int sh = (int)floor(log2(ticks / 6.0) + 0.5);
        P.prio_shift = sh < 10 ? 10 : (sh > 26 ? 26 : sh);
    }
    hipLaunchKernelGGL((score_kernel<T, NB, MINW, BF3, LT>), dim3(groups * P.split), dim3(NTHREADS), PL::BYTES, st, P);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}

int launch_score(const mcd_weights* w, int T, ScoreParams& P, hipStream_t st, bool* fused = nullptr) {
    P.force_split = w->opt[MCD_OPT_SPLIT];
    P.phase = w->opt[MCD_OPT_PHASE];
    const int variant = w->opt[MCD_OPT_VARIANT];          // tuning experiments only
    // opt-in split-bf16 channel GEMMs (layers 2..9) for 3, 6 and 12 U-Net frames (see gemm_tiles_bf3); everything measured and
    // reported by bench.py uses the fp32 path
    const bool bf3 = w->opt[MCD_OPT_BF16X3] != 0;
#if defined(MCD_FAST_T)     // developer builds (-DMCD_FAST_T=3|6|12 [-DMCD_FAST_NB= -DMCD_FAST_MINW=]): one instantiation only
#ifndef MCD_FAST_NB
#define MCD_FAST_NB (MCD_FAST_T == 3 ? 2 : 1)
#endif
#ifndef MCD_FAST_MINW
#define MCD_FAST_MINW (MCD_FAST_T >= 8 ? 2 : 4)
#endif
    if (T != MCD_FAST_T) return fail(MCD_EUNSUPPORTED, "fast build");
    return launch_score_t<MCD_FAST_T, MCD_FAST_NB, MCD_FAST_MINW>(P, st, fused);
#elif defined(MCD_FAST_BUILD)   // developer builds: only the two default-shape instantiations
    if (T == 3 && bf3) return launch_score_t<3, 2, 4, true>(P, st, fused);
    if (T == 3) return variant == 2 ? launch_score_t<3, 2, 2>(P, st, fused) : launch_score_t<3, 2, 4>(P, st, fused);
    return fail(MCD_EUNSUPPORTED, "fast build");
#else
    switch (T) {
        case 3:
            if (bf3) return launch_score_t<3, 2, 4, true>(P, st, fused);
            if (variant == 0) return launch_score_t<3, 2, 4>(P, st, fused);   // default: 2 chains / WG, 2 WGs per CU (<=128 VGPR)
            if (variant == 1) return launch_score_t<3, 4, (NWAVES == 16 ? 4 : 2)>(P, st, fused);   // 4 chains / WG, 1 WG per CU
            if (variant == 3) return launch_score_t<3, 1, 4>(P, st, fused);   // 1 chain / WG (tuning experiment with MCD_NWAVES=4)
            return launch_score_t<3, 2, 2>(P, st, fused);
        case 6:
            if (bf3) return launch_score_t<6, 1, 4, true>(P, st, fused);
            if (variant == 1) return launch_score_t<6, 2, 2>(P, st, fused);   // 2 chains / WG, 1 WG per CU (no register cap)
            return launch_score_t<6, 1, 4>(P, st, fused);                     // 1 chain / WG, 2 WGs per CU
        case 12: return bf3 ? launch_score_t<12, 1, 2, true>(P, st, fused) : launch_score_t<12, 1, 2>(P, st, fused);
        case 4: return launch_score_t<4, 1, 4>(P, st, fused);                 // e.g. seg_len 8 split in halves
        case 5: return launch_score_t<5, 2, 2>(P, st, fused);                 // e.g. seg_len 10 split in halves (2 chains / WG, 1 WG per CU)
        case 8: return launch_score_t<8, 1, 2>(P, st, fused);                 // e.g. seg_len 8 concat / seg_len 12 with 4 condition frames
        case 10: return launch_score_t<10, 1, 2>(P, st, fused);               // e.g. seg_len 20 split in halves / seg_len 10 concat
        case 7: return launch_score_t<7, 1, 2>(P, st, fused);                 // odd frame counts: one output frame per mix unit
        case 9: return launch_score_t<9, 1, 2>(P, st, fused);
        case 11: return launch_score_t<11, 1, 2>(P, st, fused);
        case 1: return launch_score_t<1, 4, 4>(P, st, fused);                 // (4 chains / WG, 2 WGs per CU)
        case 2: return launch_score_t<2, 3, 4>(P, st, fused);                 // e.g. seg_len 4 split in halves (3 chains / WG, 2 WGs per CU)
        default: return fail(MCD_EUNSUPPORTED, "U-Net frame count " + std::to_string(T) + " not instantiated (supported: 1 .. 12)");
    }
#endif
}

}  // namespace

namespace {
template <int T, int NB>
int launch_cond_fast_t(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, hipStream_t st) {
    constexpr int P17 = ceil16(NB * T * 17);
    constexpr size_t bytes = (size_t)P17 * (2 * 20 + 2 * 36) * 4;
    LDS_LIMIT((&cond_fast_kernel<T, NB>), bytes);
    hipLaunchKernelGGL((cond_fast_kernel<T, NB>), dim3((B + NB - 1) / NB), dim3(NTHREADS), bytes, st, w->dbuf, data, fi, seg_len, emb, B);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}
int launch_cond_fast(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, hipStream_t st) {
    switch (w->cond.Tc) {
        case 3: return launch_cond_fast_t<3, 2>(w, data, fi, seg_len, emb, B, st);
        case 5: return launch_cond_fast_t<5, 2>(w, data, fi, seg_len, emb, B, st);
        case 6: return launch_cond_fast_t<6, 2>(w, data, fi, seg_len, emb, B, st);
        case 10: return launch_cond_fast_t<10, 1>(w, data, fi, seg_len, emb, B, st);
        case 4: return launch_cond_fast_t<4, 2>(w, data, fi, seg_len, emb, B, st);
        case 7: return launch_cond_fast_t<7, 1>(w, data, fi, seg_len, emb, B, st);
        case 8: return launch_cond_fast_t<8, 1>(w, data, fi, seg_len, emb, B, st);
        case 9: return launch_cond_fast_t<9, 1>(w, data, fi, seg_len, emb, B, st);
        case 11: return launch_cond_fast_t<11, 1>(w, data, fi, seg_len, emb, B, st);
        case 1: return launch_cond_fast_t<1, 4>(w, data, fi, seg_len, emb, B, st);
        case 2: return launch_cond_fast_t<2, 3>(w, data, fi, seg_len, emb, B, st);
        case 12: return launch_cond_fast_t<12, 1>(w, data, fi, seg_len, emb, B, st);
        default: return fail(MCD_EUNSUPPORTED, "cond_fast: frame count not instantiated");
    }
}
template <int T, int NB>
int launch_cond_unet_t(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, hipStream_t st) {
    constexpr size_t lds = (size_t)CondUnetLds<T, NB>::FLOATS * 4;
    LDS_LIMIT((&cond_unet_kernel<T, NB>), lds);
    hipLaunchKernelGGL((cond_unet_kernel<T, NB>), dim3((B + NB - 1) / NB), dim3(NTHREADS), lds, st, w->dbuf, data, fi, seg_len, emb, B);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}
// frame counts the MFMA 'E_unet' encoder is instantiated for (the trajectory kernel's LDS plans)
bool cond_unet_has_kernel(int Tc) {
#ifdef MCD_FAST_T
    return Tc == 3 || Tc == 6 || Tc == 12;
#else
    return Tc >= 1 && Tc <= 12;
#endif
}
int launch_cond_unet(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, hipStream_t st) {
    switch (w->cond.Tc) {
        case 3: return launch_cond_unet_t<3, 2>(w, data, fi, seg_len, emb, B, st);
        case 6: return launch_cond_unet_t<6, 1>(w, data, fi, seg_len, emb, B, st);
        case 12: return launch_cond_unet_t<12, 1>(w, data, fi, seg_len, emb, B, st);
#ifndef MCD_FAST_T
        case 4: return launch_cond_unet_t<4, 2>(w, data, fi, seg_len, emb, B, st);
        case 5: return launch_cond_unet_t<5, 2>(w, data, fi, seg_len, emb, B, st);
        case 8: return launch_cond_unet_t<8, 1>(w, data, fi, seg_len, emb, B, st);
        case 10: return launch_cond_unet_t<10, 1>(w, data, fi, seg_len, emb, B, st);
        case 7: return launch_cond_unet_t<7, 1>(w, data, fi, seg_len, emb, B, st);
        case 9: return launch_cond_unet_t<9, 1>(w, data, fi, seg_len, emb, B, st);
        case 11: return launch_cond_unet_t<11, 1>(w, data, fi, seg_len, emb, B, st);
        case 1: return launch_cond_unet_t<1, 4>(w, data, fi, seg_len, emb, B, st);
        case 2: return launch_cond_unet_t<2, 3>(w, data, fi, seg_len, emb, B, st);
#endif
        default: return fail(MCD_EUNSUPPORTED, "E_unet condition encoder: frame count not instantiated");
    }
}
constexpr int GEN_MAX_WGS = 2048;       // persistent grid of the runtime-shape kernels (8 workgroups of 4 waves per CU)
int64_t gen_scratch_bytes(int64_t units, int T) {
    const int64_t wgs = units < GEN_MAX_WGS ? units : GEN_MAX_WGS;
    return wgs * (int64_t)GEN_SLAB * T * 4;
}
int launch_score_generic(const mcd_weights* w, const ScoreParams& P, const FrameMaps& M, float* scratch, hipStream_t st) {
    const int T = w->cfg.t_unet;
    const int wgs = P.n_chains < GEN_MAX_WGS ? P.n_chains : GEN_MAX_WGS;
    const size_t lds = ((size_t)3 * C0 * T * 17 + EMB_TOTAL + 4 + EDIM + GEN_THREADS) * 4;
    hipLaunchKernelGGL(score_generic_kernel, dim3(wgs), dim3(GEN_THREADS), lds, st, P, M, w->gen, T, scratch);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}
// MFMA kernel of the long windows (12 < T <= 32); slabs: tl_slab_floats(TP) floats per workgroup
// chains per workgroup of the slab-tiled kernel: two 16-frame chains share one (see score_tiled_kernel)
constexpr int tl_nb(int TP) { return TP <= 16 ? 2 : 1; }
template <int TP, int NB>
int launch_score_tiled_t(const mcd_weights* w, const ScoreParams& P, const FrameMaps& M, float* scratch, int wgs, hipStream_t st) {
    constexpr int TF = TP * NB;
    constexpr size_t lds = ((size_t)tl_ra_floats(TF) + (TF * 17 + 16) * 4 + NB * (EMB_TOTAL + 4 + EDIM) + TF * 17 * 2 * 2 + (TF * 17 + 16) * 4 + NTHREADS) * 4;
    LDS_LIMIT((&score_tiled_kernel<TP, NB>), lds);
    hipLaunchKernelGGL((score_tiled_kernel<TP, NB>), dim3(wgs), dim3(NTHREADS), lds, st, P, M, w->tiled, w->cfg.t_unet, scratch);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}
int tiled_wgs(int64_t chains, int TP) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int64_t units = (chains + tl_nb(TP) - 1) / tl_nb(TP);
    return (int)(units < cus ? units : cus);         // one workgroup per CU (110 - 135 KB of LDS), persistent over the chains
}
int64_t tiled_scratch_bytes(int64_t chains, int TP) { return (int64_t)tiled_wgs(chains, TP) * tl_slab_floats(TP * tl_nb(TP)) * 4; }
int launch_score_tiled(const mcd_weights* w, const ScoreParams& P, const FrameMaps& M, float* scratch, hipStream_t st) {
    const int wgs = tiled_wgs(P.n_chains, w->tiled_tp);
    switch (w->tiled_tp) {
        case 16: return launch_score_tiled_t<16, tl_nb(16)>(w, P, M, scratch, wgs, st);
        case 24: return launch_score_tiled_t<24, tl_nb(24)>(w, P, M, scratch, wgs, st);
        case 32: return launch_score_tiled_t<32, tl_nb(32)>(w, P, M, scratch, wgs, st);
        default: return fail(MCD_EUNSUPPORTED, "tiled kernel: frame count");
    }
}
// plain condition encoder (any channel list; 13 .. 31 condition frames of the shipped one).  scratch: cond_plain_scratch_bytes()
// of global memory when three LDS buffers do not fit (W.gmode), else unused
constexpr int CE_MAX_WGS = 512;
int64_t cond_plain_scratch_bytes(const mcd_weights* w, int64_t B) {
    if (!w->has_cond || w->cond_unet || !w->cond.gmode) return 0;
    return (B < CE_MAX_WGS ? B : CE_MAX_WGS) * (int64_t)w->cond.cmax * w->cond.Tc * 17 * 4;
}
int launch_cond_plain(const mcd_weights* w, const float* cond_data, int B, float* emb, float* scratch, hipStream_t st) {
    const bool g = w->cond.gmode != 0;
    if (g && !scratch) return fail(MCD_EINVAL, "workspace required (mcd_score_workspace_bytes) for this many condition frames");
    const size_t lds = ((size_t)(g ? 2 : 3) * w->cond.cmax * w->cond.Tc * 17 + CE_THREADS) * 4;
    LDS_LIMIT(&cond_encode_kernel, (size_t)160 * 1024);
    const int wgs = g && B > CE_MAX_WGS ? CE_MAX_WGS : B;
    hipLaunchKernelGGL(cond_encode_kernel, dim3(wgs), dim3(CE_THREADS), lds, st, w->cond, cond_data, emb, B, g ? scratch : nullptr);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}
// the condition encoders that read the condition frames straight from the window view: the MFMA kernels for the frame
// counts they are instantiated for, the runtime-shape 'E_unet' kernel otherwise (scratch: gen_scratch_bytes(B, Tc))
int launch_cond_mfma(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, float* scratch,
                     hipStream_t st) {
    if (!w->cond_unet) return launch_cond_fast(w, data, fi, seg_len, emb, B, st);
    const int Tc = w->cond.Tc;
    if (cond_unet_has_kernel(Tc) && !w->opt[MCD_OPT_COND_GENERIC]) return launch_cond_unet(w, data, fi, seg_len, emb, B, st);
    if (!scratch) return fail(MCD_EINVAL, "workspace required (mcd_score_workspace_bytes) for the runtime-shape condition encoder");
    const int wgs = B < GEN_MAX_WGS ? B : GEN_MAX_WGS;
    hipLaunchKernelGGL(cond_unet_generic_kernel, dim3(wgs), dim3(GEN_THREADS), 0, st, w->dbuf, w->gcond, data, fi, seg_len, Tc, B, emb, scratch);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}
}  // namespace

static unsigned long long* g_prof = nullptr;  // MCD_PROFILE builds: device buffer of 32 accumulators

// test aid (mcd_debug_poison_lds): every CU's LDS filled with signalling garbage (NaN bit patterns), so that a kernel reading
// shared memory it never wrote produces NaNs instead of depending on what the previous kernel happened to leave there
__global__ __launch_bounds__(NTHREADS) void poison_lds_kernel(unsigned* sink, int words) {
    extern __shared__ unsigned psm[];
    for (int u = threadIdx.x; u < words; u += NTHREADS) psm[u] = 0x7fc00000u | (unsigned)u;
    __syncthreads();
    if (threadIdx.x == 0 && sink) atomicOr(sink, psm[(blockIdx.x * 7919) % words] & 1u);      // (keeps the stores alive)
    __builtin_amdgcn_s_sleep(64);
}

extern "C" {

void mcd_debug_set_prof(void* p) { g_prof = reinterpret_cast<unsigned long long*>(p); }

int mcd_debug_poison_lds(void* stream) {
    constexpr size_t lds = 160 * 1024;
    LDS_LIMIT(poison_lds_kernel, lds);
    // one 160 KB workgroup per CU at a time; several waves of them so that every CU of every XCD takes at least one
    hipLaunchKernelGGL(poison_lds_kernel, dim3(4096), dim3(NTHREADS), lds, static_cast<hipStream_t>(stream), (unsigned*)nullptr, (int)(lds / 4));
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}

const char* mcd_last_error(void) { return g_err.c_str(); }
int32_t mcd_abi_version(void) { return MCD_ABI_VERSION; }

int mcd_pack_weights(const mcd_tensor_t* tensors, int32_t n_tensors, const mcd_model_cfg_t* cfg, int32_t device,
                     mcd_weights_t** out) {
    if (!tensors || !cfg || !out) return fail(MCD_EINVAL, "null argument");
    if (cfg->num_coords != C0) return fail(MCD_EUNSUPPORTED, "num_coords must be 2");
    if (cfg->n_joints != 17) return fail(MCD_EUNSUPPORTED, "n_joints must be 17 (the reference U-Net hard-wires 17/12/10 joints)");
    if (cfg->emb_dim != EDIM) return fail(MCD_EUNSUPPORTED, "embedding_dim must be 16");
    const int T = cfg->t_unet;
    if (T < 1 || T > MCD_MAX_FRAMES) return fail(MCD_EUNSUPPORTED, "U-Net frame count must be in 1.." + std::to_string(MCD_MAX_FRAMES));
    const bool fast_unet = T >= 1 && T <= 12;     // the instantiated score_kernel<T,...>
    GenNet G;
    memset(&G, 0, sizeof(G));
    GenCond GC;
    memset(&GC, 0, sizeof(GC));
    TensorMap tm;
    for (int i = 0; i < n_tensors; ++i) tm.m[tensors[i].name] = {tensors[i].data, tensors[i].numel};

    Builder B;
    struct HostLayer { int tq, am, wp, bias; float slope; int wpb; };
    struct { HostLayer L[NLAYERS]; int we, be, rs_w[4], rs_b[4]; } U;
    memset(&U, 0, sizeof(U));
    B.alloc(TAB_FLOATS);  // offset table lives at the start of the buffer
    static const char* names[NLAYERS] = {"st_gcnnsp1a.0", "st_gcnnsd1.0", "st_gcnnsd1.1", "st_gcnnsd2.0", "st_gcnnsd2.1",
                                         "st_gcnnsd3.0", "st_gcnnsd3.1", "st_gcnnsu4.0", "st_gcnnsu4.1", "st_gcnnsu3.0",
                                         "st_gcnnsu3.1"};
    U.we = B.alloc((size_t)EMB_TOTAL * EDIM);
    U.be = B.alloc(EMB_TOTAL + 28);
    for (int l = 0; l < NLAYERS; ++l) {
        const LDesc D = layer_desc(l);
        const std::string p = std::string("model.") + names[l];
        if (!pack_mix_mfma(tm, p, T, D.V, B, U.L[l].tq, U.L[l].am)) return fail(MCD_EMISSING, tm.missing);
        const int cin = l == 0 ? C0 : D.cin;   // real input channels (layer 0 is zero-padded to one 16-channel block)
        Folded ft, fr;
        if (!fold_conv_bn(tm, p + ".tcn.0", p + ".tcn.1", D.cout, cin, ft)) return fail(MCD_EMISSING, tm.missing);
        if (D.res && !fold_conv_bn(tm, p + ".residual.0", p + ".residual.1", D.cout, cin, fr)) return fail(MCD_EMISSING, tm.missing);
        const float* sl = tm.get(p + ".prelu.weight", 1);
        const float* we = tm.get(p + ".emb_layer.1.weight", (int64_t)D.cout * EDIM);
        const float* be = tm.get(p + ".emb_layer.1.bias", D.cout);
        if (!sl || !we || !be) return fail(MCD_EMISSING, tm.missing);
        U.L[l].slope = sl[0];
        memcpy(&B.buf[U.we + (size_t)emb_off(l) * EDIM], we, sizeof(float) * D.cout * EDIM);
        memcpy(&B.buf[U.be + emb_off(l)], be, sizeof(float) * D.cout);
        {   // plain layout for the runtime-shape kernel
            GLayer& g = G.L[l];
            g.cin = cin; g.cout = D.cout; g.V = D.V; g.slope = sl[0]; g.embo = emb_off(l);
            if (!pack_mix(tm, p, T, D.V, B, g.tq, g.am)) return fail(MCD_EMISSING, tm.missing);
            g.wt = B.alloc(ft.w.size());
            for (size_t i = 0; i < ft.w.size(); ++i) B.buf[g.wt + i] = (float)ft.w[i];
            g.wr = -1;
            if (D.res) { g.wr = B.alloc(fr.w.size()); for (size_t i = 0; i < fr.w.size(); ++i) B.buf[g.wr + i] = (float)fr.w[i]; }
            g.bias = B.alloc(D.cout);
            for (int o = 0; o < D.cout; ++o) B.buf[g.bias + o] = (float)(ft.b[o] + (D.res ? fr.b[o] : 0.0));
        }
        const int mpad = ceil16(D.cout);
        U.L[l].bias = B.alloc(mpad);
        for (int o = 0; o < D.cout; ++o) B.buf[U.L[l].bias + o] = (float)(ft.b[o] + (D.res ? fr.b[o] : 0.0));
        // MFMA fragment order.  Logical matrix Wcat[M][K] (cinp = input channels padded to 16):
        //   mix-first layers: M = cout, K = cinp (W_t') + cinp (W_r', when the layer has a residual conv)
        //   W-first layers 6 and 10: M = [W_t' ; W_r'] stacked (layer 10: rows 0,1 / 2,3 of one 16-row tile), K = cinp
        const bool wfirst = (l == 6 || l == 10);
        const int cinp = D.cin;
        const int M = l == 6 ? 2 * D.cout : mpad;
        const int Kc = wfirst ? cinp : cinp * (D.res ? 2 : 1);
        auto wt = [&](int r, int k) -> double { return (r < D.cout && k < cin) ? ft.w[(size_t)r * cin + k] : 0.0; };
        auto wr = [&](int r, int k) -> double { return (r < D.cout && k < cin) ? fr.w[(size_t)r * cin + k] : 0.0; };
        auto wcat = [&](int r, int k) -> double {
            if (wfirst) return r < D.cout ? wt(r, k) : (r < 2 * D.cout ? wr(r - D.cout, k) : 0.0);
            return k < cinp ? wt(r, k) : wr(r, k - cinp);
        };
        if (l == 10) {
            // layer 10's W-first product has 4 useful rows ([W_t' ; W_r'], 2 + 2): kept as plain rows for the FMA path
            U.L[l].wp = B.alloc(4 * 32);
            for (int r = 0; r < 4; ++r) for (int k = 0; k < 32; ++k) B.buf[U.L[l].wp + r * 32 + k] = (float)wcat(r, k);
        } else {
            U.L[l].wp = pack_gemm_frags(B, M, Kc, wcat);
            if (l >= 2 && l <= 9) U.L[l].wpb = pack_gemm_frags_bf3(B, M, Kc, wcat);
        }
    }
    static const char* rs_names[4] = {"down1", "down2", "up3", "up2"};
    static const int rs_in[4] = {17, 12, 10, 12}, rs_out[4] = {12, 10, 12, 17};
    for (int r = 0; r < 4; ++r) {
        Folded f;
        const std::string p = std::string("model.") + rs_names[r];
        if (!fold_conv_bn(tm, p + ".block.0", p + ".block.1", rs_out[r], rs_in[r], f)) return fail(MCD_EMISSING, tm.missing);
        const int vin = rs_in[r], vout = rs_out[r];
        const bool capture = r < 2;   // the down-samplers capture the skip tensors (see resample_stage)
        const int KS = capture ? (vin > 16 ? 5 : 4) : (vin + 3) / 4, MTr = (vout + 15) / 16;
        U.rs_w[r] = B.alloc((size_t)MTr * KS * 64);
        U.rs_b[r] = B.alloc(32);
        for (int mt = 0; mt < MTr; ++mt) for (int ks = 0; ks < KS; ++ks) for (int lane = 0; lane < 64; ++lane) {
            // vout = 17: the second fragment holds joint 16's weights replicated over each lane group (VALU path)
            const int vo = (vout == 17 && mt == 1) ? 16 : mt * 16 + (lane & 15), v = rs_vmap(capture, vin, ks, lane >> 4);
            B.buf[U.rs_w[r] + (mt * KS + ks) * 64 + lane] = (vo < vout && v < vin) ? (float)f.w[(size_t)vo * vin + v] : 0.f;
        }
        for (int vo = 0; vo < vout; ++vo) B.buf[U.rs_b[r] + vo] = (float)f.b[vo];
        G.rs_w[r] = B.alloc((size_t)vout * vin);
        for (size_t i = 0; i < f.w.size(); ++i) B.buf[G.rs_w[r] + i] = (float)f.w[i];
        G.rs_b[r] = B.alloc(vout);
        for (int vo = 0; vo < vout; ++vo) B.buf[G.rs_b[r] + vo] = (float)f.b[vo];
    }
    G.we = U.we; G.be = U.be;
    // tables of score_tiled_kernel (12 < T <= 32): mix coefficients for the padded frame count, non-capture resampler packs;
    // GEMM fragments, biases, slopes and the embedding Linear are the specialised kernels' own
    TiledNet TN;
    memset(&TN, 0, sizeof(TN));
    const int tiled_tp = (T > 12 && T <= 32) ? (T <= 16 ? 16 : T <= 24 ? 24 : 32) : 0;
    if (tiled_tp) {
        for (int l = 0; l < NLAYERS; ++l) {
            const LDesc D = layer_desc(l);
            if (!pack_mix_mfma(tm, std::string("model.") + names[l], T, D.V, B, TN.tq[l], TN.am[l], tiled_tp)) return fail(MCD_EMISSING, tm.missing);
            TN.tqm[l] = pack_time_mfma(tm.get(std::string("model.") + names[l] + ".gcn.T", (int64_t)D.V * T * T), T, D.V, tiled_tp, tl_nb(tiled_tp), B);
            TN.wp[l] = U.L[l].wp; TN.bias[l] = U.L[l].bias; TN.slope[l] = U.L[l].slope;
            if (l == 6) {    // this kernel runs layer 6 mix-first like the others: [W_t' | W_r'] fragments (the specialised kernels' are W-first)
                Folded ft, fr;
                const std::string p6 = std::string("model.") + names[l];
                if (!fold_conv_bn(tm, p6 + ".tcn.0", p6 + ".tcn.1", D.cout, D.cin, ft) || !fold_conv_bn(tm, p6 + ".residual.0", p6 + ".residual.1", D.cout, D.cin, fr))
                    return fail(MCD_EMISSING, tm.missing);
                TN.wp[l] = pack_gemm_frags(B, D.cout, 2 * D.cin, [&](int r, int k) -> double {
                    return k < D.cin ? ft.w[(size_t)r * D.cin + k] : fr.w[(size_t)r * D.cin + k - D.cin]; });
            }
        }
        for (int r = 0; r < 4; ++r) {
            Folded f;
            if (!fold_conv_bn(tm, std::string("model.") + rs_names[r] + ".block.0", std::string("model.") + rs_names[r] + ".block.1", rs_out[r], rs_in[r], f))
                return fail(MCD_EMISSING, tm.missing);
            const int vin = rs_in[r], vout = rs_out[r], KS = (vin + 3) / 4, MTr = (vout + 15) / 16;
            const int wf = B.alloc((size_t)MTr * KS * 64 + 32);
            for (int mt = 0; mt < MTr; ++mt) for (int ks = 0; ks < KS; ++ks) for (int lane = 0; lane < 64; ++lane) {
                const int vo = (vout == 17 && mt == 1) ? 16 : mt * 16 + (lane & 15), v = rs_vmap(false, vin, ks, lane >> 4);
                B.buf[wf + (mt * KS + ks) * 64 + lane] = (vo < vout && v < vin) ? (float)f.w[(size_t)vo * vin + v] : 0.f;
            }
            for (int vo = 0; vo < vout; ++vo) B.buf[wf + MTr * KS * 64 + vo] = (float)f.b[vo];
            TN.rsw[r] = wf;
        }
        TN.we = U.we; TN.be = U.be;
    }
    // condition encoder
    CondW Cw;
    memset(&Cw, 0, sizeof(Cw));
    bool cond_fast = false;
    int ctab[4][F_STRIDE] = {{0}};
    const bool has_cond = cfg->strategy == MCD_STRATEGY_INJECT;
    const bool cond_unet = has_cond && cfg->cond_layers == MCD_COND_UNET;
    int utab[TABC_ULB + 1] = {0};   // cond table of the 'E_unet' encoder: 7 layers, 2 resamplers, Linear
    if (cond_unet) {
        const int Tc = cfg->t_cond;
        if (Tc < 1 || Tc > MCD_MAX_FRAMES) return fail(MCD_EUNSUPPORTED, "condition frames must be in 1.." + std::to_string(MCD_MAX_FRAMES));
        Cw.Tc = Tc; Cw.latent = EDIM;
        static const char* unames[7] = {"st_gcnnsp1a.0", "st_gcnnsd1.0", "st_gcnnsd1.1", "st_gcnnsd2.0", "st_gcnnsd2.1", "st_gcnnsd3.0", "st_gcnnsd3.1"};
        static const int ucin[7] = {C0, 16, 32, 32, 64, 64, 128}, ucout[7] = {16, 32, 32, 64, 64, 128, CU_OUT}, uv[7] = {17, 17, 17, 12, 12, 10, 10};
        for (int l = 0; l < 7; ++l) {
            const int cinr = ucin[l], cout = ucout[l], cinp = cinr < 16 ? 16 : cinr;
            const std::string p = std::string("condition_encoder.") + unames[l];
            Folded ft, fr;
            const bool res = cinr != cout;
            if (!fold_conv_bn(tm, p + ".tcn.0", p + ".tcn.1", cout, cinr, ft)) return fail(MCD_EMISSING, tm.missing);
            if (res && !fold_conv_bn(tm, p + ".residual.0", p + ".residual.1", cout, cinr, fr)) return fail(MCD_EMISSING, tm.missing);
            const float* sl = tm.get(p + ".prelu.weight", 1);
            if (!sl) return fail(MCD_EMISSING, tm.missing);
            int tq = 0, am = 0;
            if (!pack_mix_mfma(tm, p, Tc, uv[l], B, tq, am)) return fail(MCD_EMISSING, tm.missing);
            const int wp = pack_gemm_frags(B, ceil16(cout), cinp * (res ? 2 : 1), [&](int r, int k) -> double {
                const bool second = k >= cinp;
                const int kk = second ? k - cinp : k;
                if (r >= cout || kk >= cinr) return 0.0;
                return second ? fr.w[(size_t)r * cinr + kk] : ft.w[(size_t)r * cinr + kk];
            });
            const int bias = B.alloc(ceil16(cout));
            for (int o = 0; o < cout; ++o) B.buf[bias + o] = (float)(ft.b[o] + (res ? fr.b[o] : 0.0));
            utab[l * F_STRIDE + F_TQ] = tq; utab[l * F_STRIDE + F_AM] = am; utab[l * F_STRIDE + F_WP] = wp; utab[l * F_STRIDE + F_BIAS] = bias;
            memcpy(&utab[l * F_STRIDE + F_SLOPE], &sl[0], sizeof(float));
            {   // plain layout for cond_unet_generic_kernel
                GLayer& g = GC.L[l];
                g.cin = cinr; g.cout = cout; g.V = uv[l]; g.slope = sl[0]; g.embo = -1;
                if (!pack_mix(tm, p, Tc, uv[l], B, g.tq, g.am)) return fail(MCD_EMISSING, tm.missing);
                g.wt = B.alloc(ft.w.size());
                for (size_t i = 0; i < ft.w.size(); ++i) B.buf[g.wt + i] = (float)ft.w[i];
                g.wr = -1;
                if (res) { g.wr = B.alloc(fr.w.size()); for (size_t i = 0; i < fr.w.size(); ++i) B.buf[g.wr + i] = (float)fr.w[i]; }
                g.bias = B.alloc(cout);
                for (int o = 0; o < cout; ++o) B.buf[g.bias + o] = (float)(ft.b[o] + (res ? fr.b[o] : 0.0));
            }
        }
        static const char* urs[2] = {"down1", "down2"};
        static const int urin[2] = {17, 12}, urout[2] = {12, 10};
        for (int r = 0; r < 2; ++r) {
            Folded f;
            const std::string p = std::string("condition_encoder.") + urs[r];
            if (!fold_conv_bn(tm, p + ".block.0", p + ".block.1", urout[r], urin[r], f)) return fail(MCD_EMISSING, tm.missing);
            const int vin = urin[r], vout = urout[r], KS = (vin + 3) / 4, MTr = (vout + 15) / 16;
            const int wf = B.alloc((size_t)MTr * KS * 64), bo = B.alloc(32);
            for (int mt = 0; mt < MTr; ++mt) for (int ks = 0; ks < KS; ++ks) for (int lane = 0; lane < 64; ++lane) {
                const int vo = mt * 16 + (lane & 15), v = rs_vmap(false, vin, ks, lane >> 4);
                B.buf[wf + (mt * KS + ks) * 64 + lane] = (vo < vout && v < vin) ? (float)f.w[(size_t)vo * vin + v] : 0.f;
            }
            for (int vo = 0; vo < vout; ++vo) B.buf[bo + vo] = (float)f.b[vo];
            utab[TABC_URS + 2 * r] = wf; utab[TABC_URS + 2 * r + 1] = bo;
            GC.rs_w[r] = B.alloc((size_t)vout * vin);
            for (size_t i = 0; i < f.w.size(); ++i) B.buf[GC.rs_w[r] + i] = (float)f.w[i];
            GC.rs_b[r] = B.alloc(vout);
            for (int vo = 0; vo < vout; ++vo) B.buf[GC.rs_b[r] + vo] = (float)f.b[vo];
        }
        const int64_t F = (int64_t)CU_OUT * Tc * 10;
        const float* lw = tm.get("condition_encoder.to_time_dim.weight", F * EDIM);
        const float* lb = tm.get("condition_encoder.to_time_dim.bias", EDIM);
        if (!lw || !lb) return fail(MCD_EMISSING, tm.missing);
        utab[TABC_ULW] = B.alloc(F * EDIM); memcpy(&B.buf[utab[TABC_ULW]], lw, sizeof(float) * F * EDIM);
        utab[TABC_ULB] = B.alloc(EDIM); memcpy(&B.buf[utab[TABC_ULB]], lb, sizeof(float) * EDIM);
        GC.lw = utab[TABC_ULW]; GC.lb = utab[TABC_ULB];
    } else if (has_cond) {
        if (cfg->cond_layers < 1 || cfg->cond_layers > MCD_MAX_COND_LAYERS) return fail(MCD_EINVAL, "bad cond_layers");
        if (cfg->t_cond < 1 || cfg->t_cond > MCD_MAX_FRAMES) return fail(MCD_EUNSUPPORTED, "condition frames must be in 1.." + std::to_string(MCD_MAX_FRAMES));
        Cw.n_layers = cfg->cond_layers; Cw.Tc = cfg->t_cond; Cw.latent = EDIM; Cw.cmax = C0;
        int cin = C0;
        for (int l = 0; l < Cw.n_layers; ++l) {
            const int cout = cfg->cond_channels[l];
            if (cout < 1 || cout > 128) return fail(MCD_EUNSUPPORTED, "condition-encoder channels must be in 1..128");
            const std::string p = "condition_encoder.encoder.model_layers." + std::to_string(l);
            Cw.cin[l] = cin; Cw.cout[l] = cout; if (cout > Cw.cmax) Cw.cmax = cout;
            if (!pack_mix(tm, p, Cw.Tc, 17, B, Cw.tq[l], Cw.am[l])) return fail(MCD_EMISSING, tm.missing);
            Folded ft, fr;
            if (!fold_conv_bn(tm, p + ".tcn.0", p + ".tcn.1", cout, cin, ft)) return fail(MCD_EMISSING, tm.missing);
            const bool res = cin != cout;
            if (res && !fold_conv_bn(tm, p + ".residual.0", p + ".residual.1", cout, cin, fr)) return fail(MCD_EMISSING, tm.missing);
            const float* sl = tm.get(p + ".prelu.weight", 1);
            if (!sl) return fail(MCD_EMISSING, tm.missing);
            Cw.slope[l] = sl[0];
            Cw.wt[l] = B.alloc(ft.w.size());
            for (size_t i = 0; i < ft.w.size(); ++i) B.buf[Cw.wt[l] + i] = (float)ft.w[i];
            Cw.wr[l] = -1;
            if (res) { Cw.wr[l] = B.alloc(fr.w.size()); for (size_t i = 0; i < fr.w.size(); ++i) B.buf[Cw.wr[l] + i] = (float)fr.w[i]; }
            Cw.bias[l] = B.alloc(cout);
            for (int o = 0; o < cout; ++o) B.buf[Cw.bias[l] + o] = (float)(ft.b[o] + (res ? fr.b[o] : 0.0));
            cin = cout;
        }
        const int64_t F = (int64_t)cin * Cw.Tc * 17;
        const float* lw = tm.get("condition_encoder.btlnk.weight", F * EDIM);
        const float* lb = tm.get("condition_encoder.btlnk.bias", EDIM);
        if (!lw || !lb) return fail(MCD_EMISSING, tm.missing);
        Cw.lw = B.alloc(F * EDIM); memcpy(&B.buf[Cw.lw], lw, sizeof(float) * F * EDIM);
        Cw.lb = B.alloc(EDIM); memcpy(&B.buf[Cw.lb], lb, sizeof(float) * EDIM);
        // fast path (cond_fast_kernel): the shipped architecture at a frame count the MFMA stages are instantiated for
        cond_fast = Cw.n_layers == 4 && Cw.cout[0] == 32 && Cw.cout[1] == 16 && Cw.cout[2] == 32 && Cw.cout[3] == 32 &&
                    Cw.Tc >= 1 && Cw.Tc <= 12;
        if (cond_fast) {
            int cinr = C0;
            for (int l = 0; l < 4; ++l) {
                const int cout = Cw.cout[l], cinp = l == 0 ? 16 : cinr;
                const std::string p = "condition_encoder.encoder.model_layers." + std::to_string(l);
                Folded ft, fr;
                const bool res = cinr != cout;
                fold_conv_bn(tm, p + ".tcn.0", p + ".tcn.1", cout, cinr, ft);
                if (res) fold_conv_bn(tm, p + ".residual.0", p + ".residual.1", cout, cinr, fr);
                int tq = 0, am = 0;
                pack_mix_mfma(tm, p, Cw.Tc, 17, B, tq, am);
                const int wp = pack_gemm_frags(B, ceil16(cout), cinp * (res ? 2 : 1), [&](int r, int k) -> double {
                    const bool second = k >= cinp;
                    const int kk = second ? k - cinp : k;
                    if (r >= cout || kk >= cinr) return 0.0;
                    return second ? fr.w[(size_t)r * cinr + kk] : ft.w[(size_t)r * cinr + kk];
                });
                const int bias = B.alloc(ceil16(cout));
                for (int o = 0; o < cout; ++o) B.buf[bias + o] = (float)(ft.b[o] + (res ? fr.b[o] : 0.0));
                ctab[l][F_TQ] = tq; ctab[l][F_AM] = am; ctab[l][F_WP] = wp; ctab[l][F_BIAS] = bias;
                memcpy(&ctab[l][F_SLOPE], &Cw.slope[l], sizeof(float));
                cinr = cout;
            }
        }
        const size_t lds = ((size_t)3 * Cw.cmax * Cw.Tc * 17 + CE_THREADS) * 4;
        Cw.gmode = lds > 160 * 1024;
        if (((size_t)2 * Cw.cmax * Cw.Tc * 17 + CE_THREADS) * 4 > 160 * 1024) return fail(MCD_EUNSUPPORTED, "condition encoder activations exceed LDS");
    }
    {
        int* tab = reinterpret_cast<int*>(B.buf.data());
        for (int l = 0; l < NLAYERS; ++l) {
            tab[l * F_STRIDE + F_TQ] = U.L[l].tq; tab[l * F_STRIDE + F_AM] = U.L[l].am;
            tab[l * F_STRIDE + F_WP] = U.L[l].wp; tab[l * F_STRIDE + F_BIAS] = U.L[l].bias;
            memcpy(&tab[l * F_STRIDE + F_SLOPE], &U.L[l].slope, sizeof(float));
            tab[l * F_STRIDE + F_WPB] = U.L[l].wpb;
        }
        tab[TAB_WE] = U.we; tab[TAB_BE] = U.be;
        if (cond_unet) for (int i = 0; i <= TABC_ULB; ++i) tab[TABC + i] = utab[i];
        if (cond_fast) {
            for (int l = 0; l < 4; ++l) for (int f = 0; f < F_STRIDE; ++f) tab[TABC + l * F_STRIDE + f] = ctab[l][f];
            tab[TABC + TABC_LW] = Cw.lw; tab[TABC + TABC_LB] = Cw.lb;
        }
        for (int r = 0; r < 4; ++r) { tab[TAB_RSW + r] = U.rs_w[r]; tab[TAB_RSB + r] = U.rs_b[r]; }
    }
    const int zero_row = B.alloc(32);
    // upload on `device`, leaving the calling thread's current device as it was
    int prev_dev = 0;
    HIP_TRY(hipGetDevice(&prev_dev));
    HIP_TRY(hipSetDevice(device));
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev_dev};
    mcd_weights* w = new mcd_weights();
    memset(w->opt, 0, sizeof(w->opt));
    w->zero_row = zero_row; w->fast_unet = fast_unet; w->gen = G; w->gcond = GC; w->tiled = TN; w->tiled_tp = tiled_tp;
    w->cfg = *cfg; w->device = device; w->n_floats = B.buf.size(); w->has_cond = has_cond; w->cond_fast = cond_fast; w->cond_unet = cond_unet;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&w->dbuf), B.buf.size() * sizeof(float));
    if (e != hipSuccess) { delete w; return fail(MCD_EDEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemcpy(w->dbuf, B.buf.data(), B.buf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(w->dbuf); delete w; return fail(MCD_EDEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
    Cw.base = w->dbuf;
    w->cond = Cw;
    *out = w;
    return MCD_OK;
}

int mcd_set_option(mcd_weights_t* w, int32_t option, int32_t value) {
    if (!w) return fail(MCD_EINVAL, "null argument");
    if (option < 0 || option >= MCD_OPT_COUNT) return fail(MCD_EINVAL, "unknown option " + std::to_string(option));
    w->opt[option] = value;
    return MCD_OK;
}

void mcd_free_weights(mcd_weights_t* w) {
    if (!w) return;
    if (w->dbuf) (void)hipFree(w->dbuf);
    delete w;
}

int mcd_cond_encode(const mcd_weights_t* w, const float* cond_data, int32_t n_windows, float* emb_out, void* stream) {
    if (!w) return fail(MCD_EINVAL, "null argument");
    if (!w->has_cond) return fail(MCD_EINVAL, "model has no condition encoder");
    if (n_windows <= 0) return MCD_OK;
    if (!cond_data || !emb_out) return fail(MCD_EINVAL, "null argument");
    if (w->cond_unet || (w->cond_fast && !w->opt[MCD_OPT_COND_GENERIC])) {
        FrameIdx fi;
        for (int k = 0; k < MCD_MAX_FRAMES; ++k) fi.idx[k] = k;
        DataView dv;
        memset(&dv, 0, sizeof(dv));
        dv.data = cond_data;
        if (w->cond_unet && !cond_unet_has_kernel(w->cond.Tc))
            return fail(MCD_EUNSUPPORTED, "mcd_cond_encode: the 'E_unet' encoder at this frame count needs scratch memory; use mcd_score");
        return launch_cond_mfma(w, dv, fi, w->cond.Tc, emb_out, n_windows, nullptr, (hipStream_t)stream);
    }
    if (w->cond.gmode) return fail(MCD_EUNSUPPORTED, "mcd_cond_encode: this many condition frames need scratch memory; use mcd_score");
    return launch_cond_plain(w, cond_data, n_windows, emb_out, nullptr, (hipStream_t)stream);
}

int mcd_unet_forward(const mcd_weights_t* w, const float* x, const float* cond, const float* step_table, int32_t t,
                     int32_t n_windows, float* eps_out, void* stream) {
    if (!w) return fail(MCD_EINVAL, "null argument");
    if (n_windows <= 0) return MCD_OK;
    if (!x || !step_table || !eps_out) return fail(MCD_EINVAL, "null argument");
    ScoreParams P;
    memset(&P, 0, sizeof(P));
    P.wbuf = w->dbuf; P.x_in = x; P.cond_emb = cond; P.step_table = step_table; P.eps_out = eps_out;
    P.B = n_windows; P.S = 1; P.ns = t + 1; P.seg_len = w->cfg.t_unet; P.n_corrupt = w->cfg.t_unet; P.fixed_mask = 0;
    P.mode = 1; P.step_single = t; P.n_chains = n_windows;
    if (w->fast_unet && !w->opt[MCD_OPT_GENERIC_UNET]) return launch_score(w, w->cfg.t_unet, P, (hipStream_t)stream);
    // runtime-shape kernel (a test entry here): its scratch slabs come from the stream-ordered allocator
    hipStream_t st = (hipStream_t)stream;
    float* scratch = nullptr;
    HIP_TRY(hipMallocAsync(reinterpret_cast<void**>(&scratch), (size_t)gen_scratch_bytes(n_windows, w->cfg.t_unet), st));
    FrameMaps M;
    memset(&M, 0, sizeof(M));
    const int rc = launch_score_generic(w, P, M, scratch, st);
    HIP_TRY(hipFreeAsync(scratch, st));
    return rc;
}

int mcd_layer_forward(const mcd_weights_t* w, int32_t stage, const float* x, const float* emb, int32_t n_windows, float* out,
                      void* stream) {
    if (!w) return fail(MCD_EINVAL, "null argument");
    if (stage < 0 || stage > 14) return fail(MCD_EINVAL, "stage must be 0..10 (ST-GCN layers) or 11..14 (down1, down2, up3, up2)");
    if (n_windows <= 0) return MCD_OK;
    if (!x || !out || !emb) return fail(MCD_EINVAL, "null argument");
    ScoreParams P;
    memset(&P, 0, sizeof(P));
    P.wbuf = w->dbuf; P.cond_emb = emb; P.step_table = w->dbuf + w->zero_row;   // pe = 0: the layers see SiLU(emb)
    P.B = n_windows; P.S = 1; P.ns = 1; P.seg_len = w->cfg.t_unet; P.n_corrupt = w->cfg.t_unet;
    P.mode = 1; P.step_single = 0; P.n_chains = n_windows;
    P.lt_stage = stage; P.lt_in = x; P.lt_out = out;
    P.x_in = stage == 0 ? x : nullptr;     // layer 0 reads the chain state itself; the other stages start from x = 0
#if defined(MCD_FAST_BUILD) || defined(MCD_FAST_T)
    return fail(MCD_EUNSUPPORTED, "fast build");
#else
    switch (w->cfg.t_unet) {
        case 3: return launch_score_t<3, 2, 4, false, true>(P, (hipStream_t)stream, nullptr);
        case 6: return launch_score_t<6, 1, 4, false, true>(P, (hipStream_t)stream, nullptr);
        case 12: return launch_score_t<12, 1, 2, false, true>(P, (hipStream_t)stream, nullptr);
        default: return fail(MCD_EUNSUPPORTED, "mcd_layer_forward is instantiated for 3, 6 and 12 U-Net frames (the fixtures' shapes)");
    }
#endif
}

__global__ void philox_noise_kernel(unsigned long long seed, long long first_window, int B, int S, int K, int Tx, float* __restrict__ out) {
    // one thread per (s, k, b, tx, joint pair): exactly the draws of score_kernel (x_T: one call per element keyed
    // (element, 0, s, window); step k >= 1: one call per joint pair keyed (tx * 9 + pair, k, s, window))
    const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)S * K * B * Tx * 9;
    if (u >= total) return;
    const int jp = (int)(u % 9), tx = (int)((u / 9) % Tx);
    const int b = (int)((u / (9 * Tx)) % B), k = (int)((u / ((long long)9 * Tx * B)) % K), s = (int)(u / ((long long)9 * Tx * B * K));
    const int v0 = 2 * jp, CTV = C0 * Tx * 17;
    float* o = out + ((size_t)(s * K + k) * B + b) * CTV;
    float z[4];
    if (k == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = i & 1, v = v0 + (i >> 1);
            z[i] = v < 17 ? philox_normal(seed, (unsigned)((c * Tx + tx) * 17 + v), 0u, (unsigned)s, (unsigned)(first_window + b)) : 0.f;
        }
    } else {
        philox_normal4(seed, (unsigned)(tx * 9 + jp), (unsigned)k, (unsigned)s, (unsigned)(first_window + b), z);
    }
    o[tx * 17 + v0] = z[0];
    o[Tx * 17 + tx * 17 + v0] = z[1];
    if (v0 + 1 < 17) { o[tx * 17 + v0 + 1] = z[2]; o[Tx * 17 + tx * 17 + v0 + 1] = z[3]; }
}

int mcd_philox_noise(uint64_t seed, int64_t first_window_id, int32_t n_windows, int32_t n_samples, int32_t noise_steps,
                     int32_t n_corrupt, float* noise_out, void* stream) {
    if (n_windows <= 0) return MCD_OK;
    if (!noise_out) return fail(MCD_EINVAL, "null argument");
    if (n_samples < 1 || noise_steps < 2 || n_corrupt < 1 || n_corrupt > MCD_MAX_FRAMES) return fail(MCD_EINVAL, "bad sizes");
    const int K = noise_steps > 2 ? noise_steps - 1 : 1;
    const long long total = (long long)n_samples * K * n_windows * n_corrupt * 9;
    hipLaunchKernelGGL(philox_noise_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned long long)seed, (long long)first_window_id, n_windows, n_samples, K, n_corrupt, noise_out);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}

static int64_t ws_loss_bytes(int64_t B, int64_t S) { return (B * S * 4 + 255) / 256 * 256; }
static int64_t ws_cond_bytes(const mcd_weights* w, int64_t B) {
    const int64_t raw = B * (EDIM + C0 * (w->cfg.t_cond > 0 ? w->cfg.t_cond : 0) * 17) * 4 + 256;
    return (raw + 255) / 256 * 256;
}
int32_t mcd_plan_split(const mcd_weights_t* w, const mcd_score_cfg_t* cfg) {
    if (!w || !cfg) return fail(MCD_EINVAL, "null argument");
    if (cfg->n_windows <= 0) return 1;
    if (!w->fast_unet || w->opt[MCD_OPT_GENERIC_UNET]) return 0;
    ScoreParams P;
    memset(&P, 0, sizeof(P));
    P.B = cfg->n_windows; P.S = cfg->n_samples; P.ns = cfg->noise_steps; P.mode = 0; P.plan_only = 1;
    const int rc = launch_score(w, w->cfg.t_unet, P, nullptr);
    return rc != MCD_OK ? rc : P.split;
}

int64_t mcd_score_workspace_bytes(const mcd_weights_t* w, const mcd_score_cfg_t* cfg) {
    if (!w || !cfg) return 0;
    // condition embeddings (B,16) + gathered condition frames (B,C,Tc,V); then the scratch slabs of the runtime-shape kernels
    // (frame counts without a specialised instantiation, or MCD_OPT_GENERIC_UNET / MCD_OPT_COND_GENERIC)
    int64_t gen = 0;
    if (!w->fast_unet || w->opt[MCD_OPT_GENERIC_UNET]) gen = gen_scratch_bytes((int64_t)cfg->n_windows * cfg->n_samples, w->cfg.t_unet);
    if (!w->fast_unet && w->tiled_tp) {
        const int64_t g3 = tiled_scratch_bytes((int64_t)cfg->n_windows * cfg->n_samples, w->tiled_tp);
        if (g3 > gen) gen = g3;
    }
    if (w->cond_unet) { const int64_t g2 = gen_scratch_bytes(cfg->n_windows, w->cond.Tc); if (g2 > gen) gen = g2; }
    { const int64_t g4 = cond_plain_scratch_bytes(w, cfg->n_windows); if (g4 > gen) gen = g4; }
    return ws_cond_bytes(w, cfg->n_windows) + ws_loss_bytes(cfg->n_windows, cfg->n_samples) + gen;
}

__global__ void gather_frames_kernel(const DataView dv, float* __restrict__ out, int B, int C, int T, int V, int n,
                                     const FrameIdx fi) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= B * C * n * V) return;
    const int v = u % V, k = (u / V) % n, c = (u / (V * n)) % C, b = u / (V * n * C);
    out[u] = load_coord(dv, b, c, fi.idx[k], v, T);
}

static int launch_aggregate(const AggrParams& A, hipStream_t st) {
    hipLaunchKernelGGL(aggregate_kernel, dim3((A.B + 63) / 64), dim3(64), 0, st, A);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}

// One scoring call.  aggr = 0: per-sample losses only (loss_all required).  aggr = a loss-based MCD_AGGR_* strategy: loss_agg
// (B,) is produced too -- inside the trajectory kernel when its workgroups see all samples of their windows (one launch per
// call), by aggregate_kernel otherwise.
static int score_impl(const mcd_weights_t* w, const mcd_score_cfg_t* cfg, const float* data, const mcd_window_view_t* view,
                      const float* noise, uint64_t seed, int64_t first_window_id, const float* step_table, void* workspace,
                      int aggr, float quantile, float* loss_agg, float* loss_all, float* pose_out, void* stream) {
    if (!w || !cfg) return fail(MCD_EINVAL, "null argument");
    const int B = cfg->n_windows, S = cfg->n_samples;
    if (B <= 0) return MCD_OK;
    if (!data || !step_table) return fail(MCD_EINVAL, "null argument");
    if (aggr == 0 && !loss_all) return fail(MCD_EINVAL, "null argument");
    if (aggr != 0) {
        if (!loss_agg) return fail(MCD_EINVAL, "null argument");
        if (aggr != MCD_AGGR_BEST && aggr != MCD_AGGR_WORST && aggr != MCD_AGGR_MEAN && aggr != MCD_AGGR_MEDIAN && aggr != MCD_AGGR_QUANTILE)
            return fail(MCD_EINVAL, "mcd_score_fused aggregates losses (best, worst, mean, median, quantile); the *_pose strategies go through mcd_score + mcd_aggregate");
        if (S > 64) return fail(MCD_EUNSUPPORTED, "aggregation supports n_generated_samples <= 64");
        if (aggr == MCD_AGGR_QUANTILE && !(quantile >= 0.f && quantile <= 1.f))       // (also rejects NaN; torch.quantile raises)
            return fail(MCD_EINVAL, "quantile must be in [0, 1]");
    }
    if (S < 1 || cfg->noise_steps < 2) return fail(MCD_EINVAL, "need n_samples >= 1 and noise_steps >= 2");
    if (cfg->n_corrupt < 1 || cfg->n_cond + cfg->n_corrupt != cfg->seg_len || cfg->seg_len > MCD_MAX_FRAMES)
        return fail(MCD_EINVAL, "cond/corrupt index lists do not partition seg_len");
    const int strat = w->cfg.strategy;
    const int Tu = w->cfg.t_unet;
    const bool rnd = strat == MCD_STRATEGY_RANDOM_IMP;
    const bool keeps_cond = strat == MCD_STRATEGY_CONCAT || strat == MCD_STRATEGY_INBETWEEN_IMP || rnd;   // condition frames are U-Net input
    if (rnd && !(view && view->cond_mask)) return fail(MCD_EINVAL, "random_imp needs mcd_window_view_t.cond_mask");
    const int tf = keeps_cond ? cfg->n_cond : 0;
    if (tf + cfg->n_corrupt != Tu) return fail(MCD_EINVAL, "frame split does not match the packed U-Net (t_unet)");
    const bool generic = !w->fast_unet || w->opt[MCD_OPT_GENERIC_UNET] != 0;      // runtime-shape kernel
    if (strat == MCD_STRATEGY_INJECT && cfg->n_cond != w->cfg.t_cond) return fail(MCD_EINVAL, "n_cond does not match the packed condition encoder");
    hipStream_t st = (hipStream_t)stream;
    ScoreParams P;
    memset(&P, 0, sizeof(P));
    P.wbuf = w->dbuf; P.prof = g_prof; P.dv.data = data;
    if (view) {
        if (view->trans && !view->affine) return fail(MCD_EINVAL, "window view: trans given without an affine table");
        P.dv.base = reinterpret_cast<const long long*>(view->base); P.dv.sc = view->stride_c; P.dv.st = view->stride_t;
        P.dv.trans = view->trans; P.dv.aff = view->affine;
        if (rnd) P.win_mask = view->cond_mask;
        if (!view->base) { P.dv.sc = (long long)cfg->seg_len * 17; P.dv.st = 17; }
    }
    P.noise = noise; P.step_table = step_table; P.pose_out = pose_out;
    P.seed = seed; P.first_window = first_window_id;
    P.B = B; P.S = S; P.ns = cfg->noise_steps; P.seg_len = cfg->seg_len; P.n_corrupt = cfg->n_corrupt;
    P.loss_fn = cfg->loss_fn; P.mode = 0; P.n_chains = B * S; P.split = 1;
    P.aggr = aggr; P.aggr_q = quantile; P.loss_agg = aggr ? loss_agg : nullptr;
    // U-Net frame layout: concat = condition frames first (mocodad.py:668), imputation = natural frame order
    // (mocodad.py:672-683), inject / no_condition = the corrupt frames only
    FrameMaps M;
    memset(&M, 0, sizeof(M));
    for (int k = 0; k < tf && !rnd; ++k) {
        const int t = strat == MCD_STRATEGY_INBETWEEN_IMP ? cfg->cond_idx[k] : k;
        if (t < 0 || t >= Tu || ((P.fixed_mask >> t) & 1)) return fail(MCD_EINVAL, "bad cond_idx");
        P.fixed_mask |= 1 << t;
        M.src_frame[t] = cfg->cond_idx[k];
    }
    for (int k = 0; k < cfg->n_corrupt && !rnd; ++k) {
        const int t = strat == MCD_STRATEGY_INBETWEEN_IMP ? cfg->corrupt_idx[k] : tf + k;
        if (t < 0 || t >= Tu || ((P.fixed_mask >> t) & 1)) return fail(MCD_EINVAL, "bad corrupt_idx");
        M.src_frame[t] = cfg->corrupt_idx[k];
        M.tx_of[t] = k;
        M.pos_of[k] = t;
    }
    for (int t = 0; t < MCD_MAX_FRAMES; ++t) M.upd_of[t] = -1;
    for (int k = 0; k < cfg->n_corrupt && !rnd; ++k) {
        const int t = keeps_cond ? cfg->corrupt_idx[k] : k;     // mocodad.py:829-838: mask built from corrupt_idxs
        if (t < 0 || t >= Tu || M.upd_of[t] >= 0) return fail(MCD_EINVAL, "bad corrupt_idx");
        M.upd_of[t] = k;
        if (M.pos_of[k] != t) P.upd_shift = 1;
    }
    for (int t = 0; t < 12; ++t) {      // the specialised kernels (<= 12 frames) carry the maps in their parameter block
        P.src_frame[t] = M.src_frame[t]; P.tx_of[t] = M.tx_of[t]; P.pos_of[t] = M.pos_of[t]; P.upd_of[t] = M.upd_of[t];
    }
    // workspace: [condition embeddings | gathered condition frames][per-sample losses (B,S)][scratch slabs of the runtime-shape kernels]
    char* wsb = reinterpret_cast<char*>(workspace);
    float* ws_loss = wsb ? reinterpret_cast<float*>(wsb + ws_cond_bytes(w, B)) : nullptr;
    float* gen_scratch = wsb ? reinterpret_cast<float*>(wsb + ws_cond_bytes(w, B) + ws_loss_bytes(B, S)) : nullptr;
    P.loss_out = loss_all ? loss_all : ws_loss;       // (skipped by a fused launch when the caller did not ask for it)
    if (!generic) {           // how the call is cut into workgroups (decides where the condition encoder runs)
        P.plan_only = 1;
        const int rc = launch_score(w, Tu, P, st);
        if (rc != MCD_OK) return rc;
        P.plan_only = 0;
    }
    auto score = [&]() -> int {
        bool fused = false;
        int rc;
        if (!generic) {
            P.loss_out_optional = loss_all == nullptr;
            rc = launch_score(w, Tu, P, st, &fused);
        } else {
            if (!workspace) return fail(MCD_EINVAL, "workspace required (mcd_score_workspace_bytes) for the runtime-shape kernel");
            // 12 < T <= 32: the MFMA kernel over an L2-resident slab; everything else (and MCD_OPT_GENERIC_UNET): plain FMAs
            if (w->tiled_tp && !w->opt[MCD_OPT_GENERIC_UNET]) rc = launch_score_tiled(w, P, M, gen_scratch, st);
            else rc = launch_score_generic(w, P, M, gen_scratch, st);
        }
        if (rc != MCD_OK || aggr == 0 || fused) return rc;
        AggrParams A;        // the workgroups did not see all samples of their windows: aggregate the (B,S) losses afterwards
        memset(&A, 0, sizeof(A));
        A.loss_all = P.loss_out; A.loss_agg = loss_agg; A.B = B; A.S = S; A.C = C0; A.Tx = cfg->n_corrupt; A.V = 17;
        A.seg_len = cfg->seg_len; A.strategy = aggr; A.loss_fn = cfg->loss_fn; A.q = quantile;
        return launch_aggregate(A, st);
    };
    if (strat == MCD_STRATEGY_INJECT) {
        // the shipped encoder with as many condition frames as the U-Net has frames runs inside the trajectory kernel when
        // its workgroups own whole windows (otherwise every workgroup of a window would repeat it: its own launch then)
        if (!generic && P.split == 1 && w->cond_fast && !w->opt[MCD_OPT_COND_GENERIC] && cfg->n_cond == Tu && Tu <= 12) {
            P.cond_inkernel = 1;
            for (int k = 0; k < Tu; ++k) P.cond_idx[k] = cfg->cond_idx[k];
            return score();
        }
        if (!workspace) return fail(MCD_EINVAL, "workspace required for this condition encoder");
        float* emb = reinterpret_cast<float*>(workspace);
        float* cbuf = emb + (size_t)B * EDIM + 16;
        const int Tc = cfg->n_cond;
        if (w->cond_unet || (w->cond_fast && !w->opt[MCD_OPT_COND_GENERIC])) {
            FrameIdx fi;
            for (int k = 0; k < MCD_MAX_FRAMES; ++k) fi.idx[k] = cfg->cond_idx[k];
            int rc = launch_cond_mfma(w, P.dv, fi, cfg->seg_len, emb, B, gen_scratch, st);
            if (rc != MCD_OK) return rc;
            P.cond_emb = emb;
            return score();
        }
        const int total = B * C0 * Tc * 17;
        FrameIdx fi;
        for (int k = 0; k < MCD_MAX_FRAMES; ++k) fi.idx[k] = cfg->cond_idx[k];
        hipLaunchKernelGGL(gather_frames_kernel, dim3((total + 255) / 256), dim3(256), 0, st, P.dv, cbuf, B, C0, cfg->seg_len,
                           17, Tc, fi);
        HIP_TRY(hipGetLastError());
        int rc = launch_cond_plain(w, cbuf, B, emb, gen_scratch, st);
        if (rc != MCD_OK) return rc;
        P.cond_emb = emb;
    }
    return score();
}

int mcd_score(const mcd_weights_t* w, const mcd_score_cfg_t* cfg, const float* data, const float* noise, uint64_t seed,
              int64_t first_window_id, const float* step_table, void* workspace, float* loss_out, float* pose_out,
              void* stream) {
    return score_impl(w, cfg, data, nullptr, noise, seed, first_window_id, step_table, workspace, 0, 0.f, nullptr, loss_out, pose_out, stream);
}

int mcd_score_view(const mcd_weights_t* w, const mcd_score_cfg_t* cfg, const float* data, const mcd_window_view_t* view,
                   const float* noise, uint64_t seed, int64_t first_window_id, const float* step_table, void* workspace,
                   float* loss_out, float* pose_out, void* stream) {
    return score_impl(w, cfg, data, view, noise, seed, first_window_id, step_table, workspace, 0, 0.f, nullptr, loss_out, pose_out, stream);
}

int mcd_score_fused(const mcd_weights_t* w, const mcd_score_cfg_t* cfg, const float* data, const mcd_window_view_t* view,
                    const float* noise, uint64_t seed, int64_t first_window_id, const float* step_table, void* workspace,
                    int32_t aggregation, float quantile, float* loss_agg, float* loss_all, float* pose_out, void* stream) {
    if (aggregation == MCD_AGGR_ALL) return fail(MCD_EINVAL, "MCD_AGGR_ALL is mcd_score");
    return score_impl(w, cfg, data, view, noise, seed, first_window_id, step_table, workspace, aggregation, quantile, loss_agg, loss_all,
                      pose_out, stream);
}

int mcd_aggregate(const mcd_score_cfg_t* cfg, int32_t num_coords, int32_t n_joints, int32_t strategy, float quantile,
                  const float* loss_all, const float* pose_all, const float* data, float* loss_agg, float* pose_agg,
                  void* stream) {
    if (!cfg) return fail(MCD_EINVAL, "null argument");
    if (cfg->n_windows <= 0) return MCD_OK;
    if (!loss_all || !loss_agg) return fail(MCD_EINVAL, "null argument");
    if (cfg->n_samples > 64) return fail(MCD_EUNSUPPORTED, "aggregation supports n_generated_samples <= 64");
    if (strategy < MCD_AGGR_BEST || strategy > MCD_AGGR_QUANTILE) return fail(MCD_EINVAL, "unknown aggregation strategy");
    if (strategy == MCD_AGGR_QUANTILE && !(quantile >= 0.f && quantile <= 1.f)) return fail(MCD_EINVAL, "quantile must be in [0, 1]");
    const bool need_pose = strategy == MCD_AGGR_MEAN_POSE || strategy == MCD_AGGR_MEDIAN_POSE;
    if (need_pose && (!pose_all || !data)) return fail(MCD_EINVAL, "pose strategies need pose_all and data");
    if (pose_agg && !pose_all) return fail(MCD_EINVAL, "pose_agg requested without pose_all");
    if (cfg->n_windows <= 0) return MCD_OK;
    AggrParams P;
    memset(&P, 0, sizeof(P));
    P.loss_all = loss_all; P.pose_all = pose_all; P.data = data; P.loss_agg = loss_agg; P.pose_agg = pose_agg;
    P.B = cfg->n_windows; P.S = cfg->n_samples; P.C = num_coords; P.Tx = cfg->n_corrupt; P.V = n_joints;
    P.seg_len = cfg->seg_len; P.strategy = strategy; P.loss_fn = cfg->loss_fn; P.q = quantile;
    for (int t = 0; t < cfg->n_corrupt && t < MCD_MAX_FRAMES; ++t) P.corrupt_idx[t] = cfg->corrupt_idx[t];
    return launch_aggregate(P, (hipStream_t)stream);
}

int mcd_scatter_max(const float* scores, const int32_t* frames, const int32_t* row, int64_t n, int32_t seg_len,
                    int32_t n_rows, int32_t n_frames, float* out, void* stream) {
    if (!scores || !frames || !row || !out) return fail(MCD_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    HIP_TRY(hipMemsetAsync(out, 0, (size_t)n_rows * n_frames * sizeof(float), st));
    if (n <= 0) return MCD_OK;
    const long long total = (long long)n * seg_len;
    hipLaunchKernelGGL(scatter_max_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, scores, frames, row,
                       (long long)n, seg_len, n_frames, out);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}


int64_t mcd_frame_scores_workspace_bytes(const mcd_frame_cfg_t* c) {
    if (!c || c->n_clips <= 0 || c->num_transform <= 0 || c->n_persons <= 0 || c->max_frames <= 0) return 0;
    const int64_t rows = (int64_t)c->num_transform * c->n_clips * c->n_persons;
    return rows * c->max_frames * 4 + (rows * 4 + 255) / 256 * 256;
}

int mcd_frame_scores(const mcd_frame_cfg_t* c, const float* scores, const int64_t* trans, const int64_t* meta,
                     const int32_t* frames, int64_t n_windows, int32_t seg_len, void* workspace, double* out, void* stream) {
    if (!c || !workspace || !out) return fail(MCD_EINVAL, "null argument");
    if (c->n_clips <= 0 || c->num_transform <= 0 || c->n_persons <= 0 || c->max_frames <= 0) return fail(MCD_EINVAL, "bad sizes");
    if (!c->clip_keys || !c->clip_n_frames || !c->frame_dst || !c->clip_out_len || !c->clip_out_off || !c->gauss_weights)
        return fail(MCD_EINVAL, "null table");
    if (n_windows > 0 && (!scores || !trans || !meta || !frames)) return fail(MCD_EINVAL, "null argument");
    if (c->frames_shift < 1) return fail(MCD_EINVAL, "frames_shift must be >= 1 (the reference's score[:-shift] is empty for 0)");
    if (c->gauss_radius < 0) return fail(MCD_EINVAL, "bad gauss_radius");
    const size_t lds = (size_t)2 * c->max_frames * sizeof(double);
    if (lds > 150 * 1024) return fail(MCD_EUNSUPPORTED, "clips longer than 9600 frames");
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)c->num_transform * c->n_clips * c->n_persons;
    FrameParams Q;
    memset(&Q, 0, sizeof(Q));
    Q.scores = scores; Q.trans = reinterpret_cast<const long long*>(trans); Q.meta = reinterpret_cast<const long long*>(meta);
    Q.frames = frames; Q.clip_keys = reinterpret_cast<const long long*>(c->clip_keys); Q.clip_n = c->clip_n_frames;
    Q.dst = c->frame_dst; Q.out_len = c->clip_out_len; Q.out_off = reinterpret_cast<const long long*>(c->clip_out_off);
    Q.gauss = c->gauss_weights;
    Q.mat = reinterpret_cast<float*>(workspace);
    Q.used = reinterpret_cast<int*>(Q.mat + rows * c->max_frames);
    Q.out = out; Q.n = n_windows; Q.seg_len = seg_len; Q.n_clips = c->n_clips; Q.num_transform = c->num_transform;
    Q.P = c->n_persons; Q.F = c->max_frames; Q.pad = c->pad_size; Q.shift = c->frames_shift; Q.radius = c->gauss_radius;
    HIP_TRY(hipMemsetAsync(workspace, 0, (size_t)(rows * c->max_frames * 4 + rows * 4), st));
    if (n_windows > 0) {
        const long long total = (long long)n_windows * seg_len;
        hipLaunchKernelGGL(frame_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Q);
        HIP_TRY(hipGetLastError());
    }
    LDS_LIMIT(&frame_scores_kernel, (size_t)150 * 1024);
    hipLaunchKernelGGL(frame_scores_kernel, dim3(c->n_clips), dim3(256), lds, st, Q);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}

}  // extern "C"
