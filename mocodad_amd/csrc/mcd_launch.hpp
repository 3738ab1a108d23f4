// mcd_launch.hpp — host side shared by the translation units of libmocodad_hip.so: the weights handle, error plumbing and the
// launcher templates whose explicit instantiations (mcd_inst_*.hip) hold the kernels.
#pragma once
#include "mcd_device.hpp"
#include "mcd_score_kernel.hpp"
#include "mcd_tiled_kernel.hpp"
#include "mcd_instances.hpp"

namespace mcd {

struct CondW {
    const float* base;
    int n_layers, Tc, latent, cmax;
    int gmode;       // 1: three LDS buffers of cmax x Tc x 17 do not fit (25 .. 31 condition frames of the shipped encoder): the third one lives in global scratch
    int cin[MCD_MAX_COND_LAYERS], cout[MCD_MAX_COND_LAYERS];
    int tq[MCD_MAX_COND_LAYERS], am[MCD_MAX_COND_LAYERS], wt[MCD_MAX_COND_LAYERS], wr[MCD_MAX_COND_LAYERS];
    int bias[MCD_MAX_COND_LAYERS];
    float slope[MCD_MAX_COND_LAYERS];
    int lw, lb;
};

struct GLayer { int cin, cout, V, tq, am, wt, wr, bias, embo; float slope; };    // wr < 0: identity residual; embo < 0: no embedding
struct GenNet { GLayer L[NLAYERS]; int rs_w[4], rs_b[4], we, be; };
struct GenCond { GLayer L[7]; int rs_w[2], rs_b[2], lw, lb; };

}  // namespace mcd

struct mcd_weights {
    mcd_model_cfg_t cfg;
    int device;
    float* dbuf;
    size_t n_floats;
    mcd::CondW cond;
    bool has_cond;
    bool cond_fast;   // shipped condition-encoder architecture -> cond_fast_kernel
    bool cond_unet;   // 'E_unet' condition encoder -> cond_unet_kernel
    bool fast_unet;   // a specialised score_kernel<T,...> exists for cfg.t_unet (1 .. 12); otherwise the slab-tiled (13 .. 32) or the runtime-shape kernel
    mcd::TiledNet tiled;   // tables of score_tiled_kernel (12 < t_unet <= 32), frame count padded to tiled_tp
    int tiled_tp;     // 16, 24 or 32; 0 = none
    mcd::TiledNet tiled_cond;   // ... of its COND form: the 'E_unet' condition encoder at 13 .. 32 condition frames
    int tiled_cond_tp;
    mcd::GenNet gen;       // plain (unpacked) folded weights of the U-Net for score_generic_kernel
    mcd::GenCond gcond;    // ... and of the 'E_unet' condition encoder
    int zero_row;     // offset (floats) of 32 zero words in dbuf: an all-zero step_table row for mcd_layer_forward
    int* tune;        // 4 device words: the trajectory kernel's own measurement of its previous launch (ScoreParams::tune)
    int opt[MCD_OPT_COUNT];   // mcd_set_option values (plain ints: set before the calls they affect, like any other argument)
};

namespace mcd {

inline thread_local std::string g_err;
inline int fail(int code, const std::string& m) { g_err = m; return code; }

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(MCD_EDEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)


// Raise a kernel's dynamic-LDS limit once per device.  `done` is the kernel's own device bitmask; two host threads racing
// here both make the (idempotent) call, nobody launches before it has been made on its device.
inline int ensure_lds_limit(const void* fn, size_t bytes, std::atomic<unsigned long long>& done) {
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
inline int wg_slots(const void* fn, size_t lds, std::atomic<int> (&cache)[64], int threads) {      // (threads: the caller's NTHREADS -- a translation-unit constant)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v > 0) return v;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    v = per_cu * cus;
    cache[dev].store(v, std::memory_order_relaxed);
    return v;
}
// How a scoring call is cut into workgroups.  A workgroup owns NB windows and runs S / split of their samples in sequence.
// split = 1 (window-major) lets the condition encoder and the aggregation run inside the workgroup -- ONE launch per call,
// no workspace -- split = S (chain-major) is one trajectory per workgroup with the encoder and the aggregation as their own
// small launches.
inline int choose_split(int n_groups, int S, int slots) {
    // estimated makespan in units of one trajectory: rounds of workgroups x trajectories per workgroup; chain-major pays its
    // two extra launches (~0.05 trajectories).  (Round 2 charged window-major 2 %: what it lost was the tail of the YOUNGER
    // of the two co-resident workgroups, which the alternating wave priority of score_kernel removes -- the one-launch form is
    // now the faster one at equal rounds, profiles/r03s_prio_slices.txt.)
    auto rounds = [&](long long wgs) { return (double)((wgs + slots - 1) / slots); };
    const double window_major = rounds(n_groups) * S;
    const double chain_major = rounds((long long)n_groups * S) + 0.05;
    return S > 1 && chain_major < window_major ? S : 1;
}

// NWAVES / NTHREADS are constants of the TRANSLATION UNIT (-DMCD_NWAVES of a unit's flags, mcd_instances.hpp), and Plan, MixCfg,
// Tiling, TlStage and the kernels depend on them: every kernel and launcher is instantiated in exactly one unit (the others see
// `extern template`), the launchers below refuse to be instantiated with another wave count than the shipped one of their
// frame count, and mcd_api.hip -- which is built with the default and only CALLS launchers -- poisons the two names.
constexpr int shipped_waves_score(int t) { return t >= 9 && t <= 12 ? 12 : 8; }
constexpr int shipped_waves_tiled(int tp) { return tp > 16 ? 12 : 8; }
#if defined(MCD_FAST_T) || defined(MCD_ANY_NWAVES)      // developer builds measure other wave counts
#define MCD_WAVES_CHECK(expected) static_assert(true, "")
#else
#define MCD_WAVES_CHECK(expected) static_assert(NWAVES == (expected), "this unit's MCD_NWAVES is not the wave count the kernel ships with (MCD_UNIT_FLAGS_<n> in mcd_instances.hpp)")
#endif

template <int T, int NB, int MINW, bool LT = false>
int launch_score_t(ScoreParams& P, hipStream_t st, bool* fused) {
    MCD_WAVES_CHECK(shipped_waves_score(T));
    using PL = Plan<T, NB>;
    LDS_LIMIT((&score_kernel<T, NB, MINW, LT>), PL::BYTES);
    static std::atomic<int> slots_cache[64];
    const int groups = (P.B + NB - 1) / NB;
    const int slots = wg_slots(reinterpret_cast<const void*>(&score_kernel<T, NB, MINW, LT>), PL::BYTES, slots_cache, NTHREADS);
    if (P.plan_only || P.mode != 0) {
        P.split = P.mode == 0 ? choose_split(groups, P.S, slots) : 1;
        if (P.mode == 0 && P.force_split > 0) P.split = P.force_split < P.S ? P.force_split : P.S;
        if (P.plan_only) return MCD_OK;
    }
    // the in-kernel condition encoder / aggregation need the workgroup to see all samples of its windows; its LDS holds 64
    // per-sample losses per window: more samples are aggregated by aggregate_kernel from the (B,S) losses
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
        P.prio_rounds = (int)rounds;
    }
    hipLaunchKernelGGL((score_kernel<T, NB, MINW, LT>), dim3(groups * P.split), dim3(NTHREADS), PL::BYTES, st, P);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}


// largest frame count of the MFMA encoder of the shipped architecture: its four 17-joint layers need 112 floats of LDS per column
constexpr int MCD_COND_FAST_MAX_T = 20;
template <int T, int NB>
int launch_cond_fast_t(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, hipStream_t st) {
    constexpr int P17 = ceil16(NB * T * 17);
    constexpr size_t bytes = (size_t)P17 * (2 * 20 + 2 * 36) * 4;
    LDS_LIMIT((&cond_fast_kernel<T, NB>), bytes);
    hipLaunchKernelGGL((cond_fast_kernel<T, NB>), dim3((B + NB - 1) / NB), dim3(NTHREADS), bytes, st, w->dbuf, data, fi, seg_len, emb, B);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}

template <int T, int NB>
int launch_cond_unet_t(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, hipStream_t st) {
    constexpr size_t lds = (size_t)CondUnetLds<T, NB>::FLOATS * 4;
    LDS_LIMIT((&cond_unet_kernel<T, NB>), lds);
    hipLaunchKernelGGL((cond_unet_kernel<T, NB>), dim3((B + NB - 1) / NB), dim3(NTHREADS), lds, st, w->dbuf, data, fi, seg_len, emb, B);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}

// MFMA kernel of the long windows (12 < T <= 32); slabs: tl_slab_floats(TP) floats per workgroup
// chains per workgroup of the slab-tiled kernel: two 16-frame chains share one (see score_tiled_kernel)
#ifndef MCD_TL16_NB
#define MCD_TL16_NB 1
#endif
constexpr int tl_nb(int TP) { return TP <= 16 ? MCD_TL16_NB : 1; }
// workgroups per CU: a single 16-frame chain (-DMCD_TL16_NB=1) leaves room for two (75 KB of LDS each, 128 registers)
constexpr int tl_wgs_per_cu(int TP) { return TP * tl_nb(TP) <= 16 ? 2 : 1; }
template <int TP, int NB, bool LT = false, bool COND = false>
int launch_score_tiled_t(const mcd_weights* w, const ScoreParams& P, const FrameMaps& M, float* scratch, int wgs, hipStream_t st) {
    MCD_WAVES_CHECK(COND ? NWAVES : shipped_waves_tiled(TP));      // (the COND form -- the 'E_unet' encoder on the tiled stages -- runs on eight or twelve)
    constexpr int TF = TP * NB;
    constexpr size_t lds = ((size_t)tl_ra_floats(TF) + (TF * 17 + 16) * 4 + NB * (EMB_TOTAL + 4 + EDIM) + TF * 17 * 2 * 2 + (TF * 17 + 16) * 4 + NTHREADS + 128 + 32 + tl_ex_floats(TF)) * 4;
    LDS_LIMIT((&score_tiled_kernel<TP, NB, LT, COND>), lds);
    ScoreParams Q = P;
    Q.prio_shift = 0;
    const int phase = w->opt[MCD_OPT_PHASE];      // (-1: no slices; -(16 + shift): a given slice length, tuning)
    if (TF <= 16 && !LT && !COND && P.mode == 0 && phase != -1 && wgs > 0) {
        // priority time slice of the two co-resident workgroups: about 1/6 of the launch -- chains per workgroup x passes x
        // ~190 us per pass of a 16-frame chain sharing its CU (2.4 GHz), in ticks of the 100 MHz clock
        const double per_wg = (double)((P.n_chains + (long long)wgs * NB - 1) / ((long long)wgs * NB));
        const double ticks = per_wg * (double)(P.ns > 2 ? P.ns - 1 : 1) * 190.0 * 100.0;
        const int sh = (int)floor(log2(ticks / 6.0) + 0.5);
        Q.prio_shift = sh < 10 ? 10 : (sh > 26 ? 26 : sh);
        if (phase < -15) Q.prio_shift = -phase - 16 > 26 ? 26 : (-phase - 16 < 10 ? 10 : -phase - 16);
    }
    hipLaunchKernelGGL((score_tiled_kernel<TP, NB, LT, COND>), dim3(wgs), dim3(NTHREADS), lds, st, Q, M, COND ? w->tiled_cond : w->tiled,
                       COND ? w->cond.Tc : w->cfg.t_unet, scratch);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}


// developer builds (-DMCD_FAST_T=3|6|12 [-DMCD_FAST_NB= -DMCD_FAST_MINW= -DMCD_FAST_TILED=16|24|32 -DMCD_FAST_TILED_COND=16|24|32]): one trajectory kernel only
#ifdef MCD_FAST_T
#ifndef MCD_FAST_NB
#define MCD_FAST_NB (MCD_FAST_T == 3 ? 2 : 1)
#endif
#ifndef MCD_FAST_MINW
#define MCD_FAST_MINW (MCD_NWAVES == 12 ? 3 : MCD_FAST_T >= 7 ? 2 : 4)
#endif
#endif

// Every instantiation lives in exactly one unit of mcd_inst.hip; everywhere else it is only declared.
#ifdef MCD_FAST_T
extern template int launch_score_t<MCD_FAST_T, MCD_FAST_NB, MCD_FAST_MINW, false>(ScoreParams&, hipStream_t, bool*);
extern template int launch_cond_fast_t<MCD_FAST_T, MCD_FAST_NB>(const mcd_weights*, const DataView&, const FrameIdx&, int, float*, int, hipStream_t);
extern template int launch_cond_unet_t<MCD_FAST_T, MCD_FAST_NB>(const mcd_weights*, const DataView&, const FrameIdx&, int, float*, int, hipStream_t);
#ifdef MCD_FAST_TILED
extern template int launch_score_tiled_t<MCD_FAST_TILED, tl_nb(MCD_FAST_TILED), false>(const mcd_weights*, const ScoreParams&, const FrameMaps&, float*, int, hipStream_t);
#endif
#ifdef MCD_FAST_TILED_COND
extern template int launch_score_tiled_t<MCD_FAST_TILED_COND, tl_nb(MCD_FAST_TILED_COND), false, true>(const mcd_weights*, const ScoreParams&, const FrameMaps&, float*, int, hipStream_t);
#endif
#else
#define MCD_DECL_SCORE(unit, T, NB, MINW, LT) extern template int launch_score_t<T, NB, MINW, LT>(ScoreParams&, hipStream_t, bool*);
#define MCD_DECL_COND_FAST(unit, T, NB) \
    extern template int launch_cond_fast_t<T, NB>(const mcd_weights*, const DataView&, const FrameIdx&, int, float*, int, hipStream_t);
#define MCD_DECL_COND_UNET(unit, T, NB) \
    extern template int launch_cond_unet_t<T, NB>(const mcd_weights*, const DataView&, const FrameIdx&, int, float*, int, hipStream_t);
#define MCD_DECL_TILED(unit, TP, NB, LT) \
    extern template int launch_score_tiled_t<TP, NB, LT>(const mcd_weights*, const ScoreParams&, const FrameMaps&, float*, int, hipStream_t);
MCD_SCORE_INSTANCES(MCD_DECL_SCORE)
MCD_COND_FAST_INSTANCES(MCD_DECL_COND_FAST)
MCD_COND_UNET_INSTANCES(MCD_DECL_COND_UNET)
MCD_TILED_INSTANCES(MCD_DECL_TILED)
#define MCD_DECL_TILED_COND(unit, TP, NB) \
    extern template int launch_score_tiled_t<TP, NB, false, true>(const mcd_weights*, const ScoreParams&, const FrameMaps&, float*, int, hipStream_t);
MCD_TILED_COND_INSTANCES(MCD_DECL_TILED_COND)
#endif

}  // namespace mcd
