// mcd_tiled_kernel.hpp — score_tiled_kernel<TP, NB>: the MFMA trajectory kernel of 13 .. 32 U-Net frames.
#pragma once
#include "mcd_device.hpp"

namespace mcd {

// ------------------------------------------------------------------------------------------------
// MFMA path for 12 < T_u <= 32 U-Net frames (concat over 24 frames, 16 + 16, ...): the activations of such a chain do not
// fit LDS (257 KB at layer 5 of a 24-frame chain), so the chain lives in a slab of global memory (L2 / Infinity Cache) and a
// layer is "32 input channels of ALL frames -> LDS -> mix -> channel GEMM -> slab", one 512-thread workgroup per CU:
//   X      one 32-channel part of the layer's input, every frame, staged slab -> registers -> LDS (the next part's loads go out
//          right before this part's channel GEMM -- NOT in front of its mixes: vector-memory loads return in order, and the
//          mixes' in-place coefficient fetches wait with vmcnt(0); round 5, DESIGN.md 2.3 "in-order returns").  The four joint resamplers are not stages of their own: the layer behind one
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
// garbage that no real frame ever reads).  13 .. 16 frames run as <16, 1>: one chain per workgroup, 79 KB of LDS, TWO workgroups
// per CU under the 128-register cap (+10.6 % over two chains in one workgroup, profiles/r04v_tl16_nb1_ab.txt).
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
__host__ __device__ constexpr int tl_fc(int TP) { return TP == 24 ? 12 : TP == 16 ? 8 : 16; }
__host__ __device__ constexpr int tl_ra_floats(int TP) {
    // LDS work region of a layer: X (32 channels of all frames, + pad rows) and z (the same; half the frames at 17 joints).
    // (A fused resampler's input chunk passes through the z region.)
    return cmax((TP * 17 + 16 + TP * 17 / 2 + 16) * 36, 2 * (TP * 12 + 16) * 36);
}
// 24 frames: the LDS has room (29 KB of the 43 KB the plan leaves) for the SECOND chunk of a resampler's input as well, behind
// the layers whose output all 32 channels of the next layer's fused resampler read (2 -> down1 -> 3, 8 -> up2 -> 9): both chunks are
// handed over in LDS and layers 3 / 9 start without the store -> barrier -> load round trip through the slab (MCD_TL_EX)
#ifndef MCD_TL_EX
#define MCD_TL_EX 1
#endif
// (+ 4 pad rows: the 17-joint resampler's last k-step reads joints 16 .. 19 of the chunk's last frame -- zero coefficients on finite values)
__host__ __device__ constexpr int tl_ex_floats(int TF) { return (MCD_TL_EX && TF == 24) ? (tl_fc(TF) * 17 + 4) * 36 : 0; }
__host__ __device__ constexpr int tl_qc(int TP) { return TP % 3 == 0 ? 3 : 4; }     // output frames per mix unit (6 at 24 frames: 108 coefficient registers, spills)
__host__ __device__ constexpr long long tl_slab_floats(int TP) {
    // A0, A1 (ping-pong, up to 128 ch x 10 joints), the skips D1, D2 -- each with 16 rows of padding behind it
    return (long long)2 * (TP * 10 + 16) * 132 + (long long)(TP * 17 + 16) * 36 + (long long)(TP * 12 + 16) * 68;
}

#ifndef MCD_TL_DENSE
#define MCD_TL_DENSE 0     // 1: slab rows of exactly C floats (every 64-byte chunk of a row aligned); 0: C + 4 like the LDS rows
#endif
// row stride of a C-channel tensor in the SLAB.  The + 4 of the LDS row strides (cs_of) serves the LDS banks; in global memory it
// puts a 32-channel part of a row (128 bytes) across two 128-byte lines.  Dense rows were measured: 24 frames +0.1 %, 32 frames
// -2.2 % (power-of-two row strides; profiles/r05y_tiled_dense_rows_ab.txt) -- the padded stride stays
__host__ __device__ constexpr int ss_of(int c) { return MCD_TL_DENSE ? c : c + 4; }
// cross-layer prefetches in front of a layer's last channel GEMM (see `layer`), per frame-count class: bit 0 the next layer's first
// time-mix fragments, bit 1 the fragments of the resampler fused into the next layer, bit 2 the skip rows of layers 7 / 9.
// Measured per shape (profiles/r05z_tiled24_*_ab.txt, r05zb_tiled_pre_shapes_ab.txt): they hold registers across the GEMM
#ifndef MCD_TL_PRE16
#define MCD_TL_PRE16 7
#endif
#ifndef MCD_TL_PRE24
#define MCD_TL_PRE24 7
#endif
#ifndef MCD_TL_PRE32
#define MCD_TL_PRE32 7      // (32 frames: -0.8 .. -2 % while layer 5 held 80 accumulators per lane; +1.5 % since it runs on twelve waves, profiles/r05zj_tiled32_pre_w2_ab.txt)
#endif
__host__ __device__ constexpr int tl_pre(int TF) { return TF <= 16 ? MCD_TL_PRE16 : TF == 24 ? MCD_TL_PRE24 : MCD_TL_PRE32; }
#ifndef MCD_TL_L5_W2
#define MCD_TL_L5_W2 1
#endif
template <class T> struct TlType { using type = T; };
#ifndef MCD_TL_L10_MFMA
#define MCD_TL_L10_MFMA 1
#endif
#ifndef MCD_TL_LATE
#define MCD_TL_LATE 1      // the next part's slab loads go out right before the current part's channel GEMM (see `layer`)
#endif
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
    // (unconditional: a slot past the end re-reads the last one.  Conditional loads make the NUMBER of loads in flight depend on
    // the path, and every later wait for an older load -- a weight fragment fetched before them -- then has to assume none were
    // issued: s_waitcnt vmcnt(0), i.e. it waits for these slab loads as well)
    __device__ __forceinline__ void issue(int tid, const float* src, int ss, int ch0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int u0 = tid + i * NTHREADS, u = (MCD_TL_LATE && u0 >= ROWS * Q) ? ROWS * Q - 1 : u0;
            if (MCD_TL_LATE || u < ROWS * Q) { const int r = u / Q, c = (u - r * Q) * 4; v[i] = load_global4(src + (size_t)r * ss + ch0 + c); }
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
template <int CINV, int V, int TP, int NB = 1, int QCO = 0>
struct MixLongCoef {      // time-mix rows + joint-mix fragments of one unit's QC output frames (NB chains of TP frames each)
    static constexpr int QC = QCO > 0 ? QCO : tl_qc(TP), KS = (V + 3) / 4, MT = (V + 15) / 16, CB = CINV / 16, NQ = TP / QC;
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
template <int CINV, int V, int TP, int NB, int NGRP = 1, int QCO = 0, class Init, class Store>
__device__ __forceinline__ void mix_long(const float* __restrict__ X, int cs, const MixLongCoef<CINV, V, TP, NB, QCO>& first,
                                         const float* __restrict__ tqd, const float* __restrict__ af,
                                         int wave, int lane, Init&& init, Store&& store, int grp = 0) {
    using MC = MixLongCoef<CINV, V, TP, NB, QCO>;
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
        // the wave's later units, their MFMA chains interleaved.  Only the LAST round can lie past the end (a wave-uniform test):
        // such a wave used to recompute the last unit unstored -- free on 8 waves' idle SIMD slots, but with 12 waves per workgroup
        // those MFMAs (7 of 24 slots at 17 joints) take the matrix pipe from the SIMD's other two waves
        auto later = [&](auto nn) {
            constexpr int N = decltype(nn)::value;
            if constexpr (N > 0) {
                f32x4 accr[N][CB];
#pragma unroll
                for (int i = 0; i < N; ++i)
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) accr[i][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KT; ++ks)
#pragma unroll
                    for (int i = 0; i < N; ++i)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb) accr[i][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[i][ks], br[i][cb][ks], accr[i][cb], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < N; ++i) write_y(i + 1, ur[i], accr[i]);
            }
        };
        if (UNITS % NWAVES == 0 || wave + (PER - 1) * NWAVES < UNITS) later(std::integral_constant<int, PER - 1>{});
        else later(std::integral_constant<int, PER - 2>{});
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
    // (a wave whose round lies past the end of the units skips it -- see tl_time_mix)
    if (UNITS >= NWAVES || wave < UNITS) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, f1);
    if constexpr (PER > 1) {
        if (UNITS % NWAVES == 0 || wave + (PER - 1) * NWAVES < UNITS) run(std::integral_constant<int, 1>{}, std::integral_constant<int, PER - 1>{}, ar);
        else if constexpr (PER > 2) run(std::integral_constant<int, 1>{}, std::integral_constant<int, PER - 2>{}, ar);
    }
}

// partial channel GEMM of a layer of the slab-tiled kernel: acc[i] += A[O1 ..] . B1 (+ A[O2 ..] . B2) over this wave's n-tiles
// (Tiling<MT, NT>), K = 16 KQ1 (+ 16 KQ2) channels of the LDS operands b1 / b2 ([col][ch]); two tiles' MFMA chains in flight
// with their B fragments one read ahead, as in gemm_tiles.  The accumulators stay with the caller: a layer with 64 or 128
// input channels sums its 32-channel halves into them and runs its epilogue once.
// FIRST: the accumulators start here, from zero (the first MFMA of a tile takes the literal 0 as its C operand).  Zeroed by the
// caller instead, the accumulator of a tile slot this wave does not own was a phi of "zero" and "unchanged" at every wave-dependent
// branch below, and the register allocator materialised those zeros: 3 - 5 moves per slot and part, a quarter of the GEMM
// stages' non-MFMA instructions.  Left undefined, a slot the wave does not own is simply never read.
template <int MT, int NT, int KQ1, int KQ2, int O1, int O2, bool FIRST = false, int NA>
__device__ __forceinline__ void gemm_part(const float4 (&a)[NA], const float* __restrict__ b1, int cs1, const float* __restrict__ b2,
                                          int cs2, int wave, int lane, f32x4 (&acc)[Tiling<MT, NT>::MAXN]) {
    constexpr int NG = Tiling<MT, NT>::NG, MAXN = Tiling<MT, NT>::MAXN, KQ = KQ1 + KQ2;
    const int ng = MT > NWAVES ? 0 : (wave / MT < NG ? wave / MT : NT);      // (wave counts that MT does not divide: the waves past NG x MT get no tile)
    const int j = lane & 15, g = lane >> 4;
    // (a wave without a tile -- ng = NT -- reads its never-used read-ahead fragments from the rows of tile 0: inside the operand,
    // not behind it.  ADVICE r5: out-of-range LDS reads do return 0 on this hardware, but nothing should rest on that.)
    const int ngr = ng < NT ? ng : 0;
    const float* const p1b = b1 + __mul24(ngr * 16 + j, cs1) + 4 * g;
    const float* const p2b = b2 + __mul24(ngr * 16 + j, cs2) + 4 * g;
    constexpr int NP = (MAXN + 1) / 2;
    // first B fragment (k-group 0) of tile slot i -- read a pair ahead: the last k-group of a pair fetches the next pair's
    auto rd0 = [&](int i) { return *reinterpret_cast<const float4*>((KQ1 > 0 ? p1b + i * NG * 16 * cs1 : p2b + i * NG * 16 * cs2)); };
    float4 nxt[2];
    auto prime = [&](auto pp) {
        constexpr int p = decltype(pp)::value;
        if constexpr (p < NP) {
            constexpr int i0 = 2 * p, i1 = i0 + 1 < MAXN ? i0 + 1 : i0;
            // (unconditional: a slot past the wave's last tile reads rows nobody uses -- at most NG - 1 tiles of rows behind the
            // operand, which the LDS plan's regions behind it cover; a read past the allocation itself returns 0 --;
            // conditional reads made `nxt` a phi of every wave-dependent branch below and cost 2 - 8 register-pair moves per pair)
            nxt[0] = rd0(i0);
            if (i1 != i0) nxt[1] = rd0(i1);
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
            if constexpr (FIRST && kq == 0) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, u[0].x, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if constexpr (N == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, u[1].x, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            } else {
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, u[0].x, c0, 0, 0, 0);
                if constexpr (N == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, u[1].x, c1, 0, 0, 0);
            }
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
#ifndef MCD_TL_ONECHUNK
#define MCD_TL_ONECHUNK 1
#endif
#ifdef MCD_PROFILE
// ... and lane 0 of EVERY wave of workgroup 0 stamps the events of ONE pass (the third of its first chain): event e of wave w at
// slot 4096 + 16 e + w, the event's id at 4096 + 8192 + e (TLTR: a trace-only event)
#define TLTR(id) do { if (tl_tron) { P.prof[4096 + 16 * tl_ev + (tid0 >> 6)] = __builtin_readcyclecounter(); \
    if (tid0 == 0) P.prof[4096 + 8192 + tl_ev] = (id); } ++tl_ev; } while (0)
#define TLMARK(id) do { if (tl_prof) { const unsigned long long t_ = __builtin_readcyclecounter(); \
    atomicAdd(P.prof + 2048 + (tid0 ? 64 : 0) + (id), t_ - tl_last); tl_last = t_; } TLTR(id); } while (0)
#else
#define TLMARK(id) do { } while (0)
#define TLTR(id) do { } while (0)
#endif
// the kernel's argument list as the kernarg segment lays it out (arguments in order at their natural alignment = C struct
// layout): the per-layer table offsets of TiledNet are read through a pointer into the segment that is made opaque per step and
// per layer -- read as plain kernel arguments they are loop-invariant, the compiler hoists all ~80 of them above the step loop,
// and they came back as 441 v_writelane + 985 v_readlane of SGPR spills (round 3: 427 spilled SGPRs at 32 frames)
constexpr int CU_OUT_TL = 6;               // unet_down_channels[6] of STSE_Unet (CU_OUT of cond_unet_kernel)
struct TiledKernArgs { ScoreParams P; FrameMaps M; TiledNet N; int T; float* slabs; };
typedef const TiledNet __attribute__((address_space(4))) KTiledNet;

// P.mode 1 (mcd_unet_forward): ONE pass at step P.step_single from the caller's x_in, eps-prediction to eps_out, no update.
// LT = true (layer test, mcd_layer_forward): a single pass in which only stage P.lt_stage runs -- its input tensor lt_in is put
// where the PREVIOUS layer's epilogue would have left it (the slab and / or the LDS hand-over regions, see `layer` below), its
// output is copied to lt_out from where the stage's own epilogue puts it.  Stages 3, 5, 7, 9 are the fused (joint resampler +
// layer) stages of this kernel: their input is the resampler's input (+ lt_skip, the U-Net skip tensor added behind it).
// COND = true: the 'E_unet' condition encoder at 13 .. 32 condition frames (STSE_Unet, stsae_unet.py:62-146,182-251) on the same
// stages -- one "chain" per window, its condition frames M.src_frame[0 .. T) as input, layers 0 .. 6 with the ENCODER's tables
// (N; layer 6 is 128 -> 6 channels there, padded to one 16-row m-tile), no embeddings (t = None), then
// to_time_dim: Linear(6 * T * 10 -> 16) over the (c, t, v) flattening -> P.eps_out[window][16].
template <int TP, int NB, bool LT = false, bool COND = false>
__global__ __launch_bounds__(NTHREADS, (TP * NB <= 16 ? 2 : 1) * NWAVES / 4) void score_tiled_kernel(const ScoreParams P, const FrameMaps M, const TiledNet N, int T,
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
    float* const W4L = RED + NTHREADS;              // [4][32]  layer 10's W-first weights [W_t; W_r] (constant over the launch)
    constexpr int EXF = tl_ex_floats(TF);
    int* const UPDT = reinterpret_cast<int*>(W4L + 128);   // [32]  frame t -> frame whose chain state its prediction updates (pos_of[upd_of[t]]), or -1
    float* const EX = W4L + 128 + 32;                    // [TL_FC * 17][36]  second hand-over chunk of layers 2 -> 3, 8 -> 9 (24 frames only)
    const int tid0 = threadIdx.x;
    int tid = tid0, lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* wb = P.wbuf;
#ifdef MCD_PROFILE
    const bool tl_prof = P.prof && blockIdx.x == 0 && (tid0 == 0 || tid0 == NTHREADS - 64);
    unsigned long long tl_last = __builtin_readcyclecounter();
    bool tl_tron = false;
    int tl_ev = 0, tl_pass = 0;
#endif
    float* slab = slabs + (size_t)blockIdx.x * tl_slab_floats(TF);
    const int Tx = P.n_corrupt, K = P.ns > 2 ? P.ns - 1 : 1, per = C0 * Tx * 17;
    // everything a chain leaves behind that the next chain of this workgroup reads before writing it (pad rows / pad frames meet
    // zero coefficients: they must hold finite values) -- cleared at the start, and again behind a chain that DIVERGED (see below)
    auto clear_state = [&](int t_id) {
        for (int u = t_id; u < (int)tl_slab_floats(TF); u += NTHREADS) slab[u] = 0.f;
        for (int u = t_id; u < (R17 + 16) * 4; u += NTHREADS) { XT[u] = 0.f; P4[u] = 0.f; }
        for (int u = t_id; u < R17 * C0; u += NTHREADS) { ZN[u] = 0.f; ZO[u] = 0.f; }
        for (int u = t_id; u < EXF; u += NTHREADS) EX[u] = 0.f;
        for (int u = t_id; u < RA_F; u += NTHREADS) RA[u] = 0.f;
    };
    for (int u = tid; u < (int)tl_slab_floats(TF); u += NTHREADS) slab[u] = 0.f;      // pad rows / pad frames: finite values
    for (int u = tid; u < (R17 + 16) * 4; u += NTHREADS) XT[u] = 0.f;
    if (tid < 64) P4[R17 * 4 + tid] = 0.f;
    if (!COND && tid < 128) W4L[tid] = P.wbuf[N.wp[10] + tid];
    if (tid < 32) { const int k = tid < T ? M.upd_of[tid] : -1; UPDT[tid] = (k >= 0 && k < MCD_MAX_FRAMES) ? M.pos_of[k] : -1; }
    for (int u = tid; u < EXF; u += NTHREADS) EX[u] = 0.f;              // (its pad rows stay zero; the rest is rewritten every pass)
    for (int u = tid; u < RA_F; u += NTHREADS) RA[u] = 0.f;           // pad rows meet zero coefficients: they must be finite

    for (long long grp = blockIdx.x; grp * NB < P.n_chains; grp += gridDim.x) {
        // chain i of the group (the last group of an odd count runs its last chain twice and writes it once)
        auto chain_of = [&](int i) { const long long c = grp * NB + i; return c < P.n_chains ? c : P.n_chains - 1; };
        // (window, sample, condition-frame mask) of the group's chains: once per group -- they were 64-bit divisions and a global
        // load inside every pass's noise loop and update tail
        int bq[NB], sq[NB];
        unsigned fq[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const long long c = chain_of(i);
            bq[i] = (int)(c / P.S); sq[i] = (int)(c - (long long)bq[i] * P.S);
            fq[i] = (unsigned)(P.win_mask ? P.win_mask[bq[i]] : P.fixed_mask);
        }
        auto b_of = [&](int i) { return NB == 1 ? bq[0] : bq[i]; };
        auto s_of = [&](int i) { return NB == 1 ? sq[0] : sq[i]; };
        auto fixed_of = [&](int i) { return NB == 1 ? fq[0] : fq[i]; };
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
                if constexpr (COND) x = load_coord(P.dv, b, c, M.src_frame[t], v, P.seg_len);
                else if (P.mode == 1) x = P.x_in ? P.x_in[((size_t)(b * C0 + c) * T + t) * 17 + v] : 0.f;
                else if ((fixed >> t) & 1u) x = load_coord(P.dv, b, c, src_of(t), v, P.seg_len);
                else {
                    const int e = (c * Tx + tx_of(fixed, t)) * 17 + v;
                    x = P.noise ? P.noise[((size_t)(s * K + 0) * P.B + b) * per + e]
                                : philox_normal(P.seed, (unsigned)e, 0u, (unsigned)s, (unsigned)(P.first_window + b));
                }
                XT[((i * TP + t) * 17 + v) * 4 + c] = x;
            }
        }
        const int i_first = COND ? 0 : P.mode == 1 ? P.step_single : P.ns - 1, i_last = COND ? 0 : P.mode == 1 ? P.step_single : 1;
        for (int sidx = i_first; sidx >= i_last; --sidx) {
            const float* srow = P.step_table + sidx * (4 + EDIM);
#ifdef MCD_PROFILE
            tl_tron = P.prof && blockIdx.x == 0 && (tid0 & 63) == 0 && tl_pass == 2;
            tl_ev = 0;
            ++tl_pass;
#endif
            // opaque per step (see score_kernel): otherwise every per-lane address of every stage is hoisted out of the step
            // loop as loop-invariant and the hundreds of resulting registers are spilled
            tid = tid0;
            asm volatile("" : "+v"(tid));
            lane = tid & 63;
            wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            if constexpr (TF <= 16 && !COND) {
                // two workgroups per CU: they take turns at priority 1 by time slices of the 100 MHz clock XOR their slot on the
                // CU, as in score_kernel (the arbiter otherwise serves the older one first and the younger one runs its last
                // chains alone); the host sizes the slice to about a sixth of the launch
                if (P.prio_shift > 0) {
                    unsigned hwid;
                    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                    const unsigned slice = (unsigned)(__builtin_amdgcn_s_memrealtime() >> P.prio_shift);
                    if (((hwid >> 16) ^ slice) & 1u) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
            }
            wb = P.wbuf;
            asm volatile("" : "+s"(wb));
            float* sl = slab;
            asm volatile("" : "+s"(sl));
            KTiledNet* Ns = (KTiledNet*)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(TiledKernArgs, N));
            asm volatile("" : "+s"(Ns));
            float* const A0 = sl;
            float* const A1 = A0 + (R10 + 16) * 132;
            float* const D1 = A1 + (R10 + 16) * 132;
            float* const D2 = D1 + (R17 + 16) * 36;
            __syncthreads();
            if (!COND && tid < NB * EDIM) {
                float e = srow[4 + tid % EDIM];
                if (P.cond_emb) e += P.cond_emb[(size_t)b_of(tid / EDIM) * EDIM + tid % EDIM];
                SE[tid] = e / (1.f + expf(-e));
            }
            if (!COND && sidx > 1 && P.mode == 0) {      // this step's noise, one thread per (frame, joint pair): the same Philox keys as score_kernel
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
            TLMARK(58);                                // pass prologue: SiLU(pe + cond), noise
            for (int u = tid; !COND && u < NB * EMB_TOTAL; u += NTHREADS) {
                const int i = u / EMB_TOTAL, o = u % EMB_TOTAL;
                gfloat* we = as_global(wb + Ns->we + o * EDIM);
                float a = as_global(wb)[Ns->be + o];
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
            // the U-Net skip rows (d2 / d1, first 32 channels) that layers 7 / 9 add behind their fused resampler: written long before
            // (layers 4 / 2), so their loads go out in front of the PREVIOUS layer's last channel GEMM instead of at the layer's start
            // ... and the next layer's first time-mix fragments likewise: fetched behind a layer's own epilogue they wait for the
            // epilogue's slab stores to be acknowledged (stores count in vmcnt too, and everything returns in order)
            constexpr bool TQPRE = (tl_pre(TF) & 1) && !LT;
            float tqx[TP / 4];
            constexpr bool RCPRE = (tl_pre(TF) & 2) && !LT;      // ... and the fragments of the resampler fused into the next layer
            constexpr int FCX = tl_fc(TF);
            RsCoef<32, 17, 12, FCX, 1, false> rcx3;
            RsCoef<32, 12, 10, FCX, 1, false> rcx5;
            RsCoef<32, 10, 12, FCX, 1, false> rcx7;
            RsCoef<32, 12, 17, FCX, 1, false> rcx9;
            constexpr bool SKPRE = (tl_pre(TF) & 4) && !LT && !COND;
            TlStage<SKPRE ? TF * 12 : 1, 32> sk7;
            TlStage<SKPRE ? TF * 17 : 1, 32> sk9;
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
                constexpr bool HOR = L == 4 || (L == 6 && !COND), HIR = L == 5 || L == 7;
                constexpr int TNEXT = (TF * (L == 4 ? 10 : L == 8 ? 17 : 12) + 16) * 36;     // offset of the next layer's z region
                // ... or only the resampler's FIRST chunk where all of it does not fit (2 -> down1 -> 3, 8 -> up2 -> 9); layer 2 keeps
                // group 0's accumulators until both groups are through (the chunk's place is still its own X rows before)
                constexpr bool HOC = L == 2 || L == 8, HIC = L == 3 || L == 9;
                constexpr bool HOC2 = HOC && EXF > 0, HIC2 = HIC && EXF > 0;      // ... and the second chunk as well (EX)
                constexpr LDesc D = (COND && L == 6) ? LDesc{128, 16, 10, 1} : layer_desc(L);
                constexpr int CIN = D.cin, COUT = D.cout, V = D.V, CSI = ss_of(CIN), CSO = ss_of(COUT);
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
                KTiledNet* Nl = Ns;
                asm volatile("" : "+s"(Nl));
                if constexpr (LT) {
                    if (P.lt_stage != L) return;
                    // the stage's input, placed as the previous layer's epilogue (epi_fg of layer L - 1) places its output
                    if constexpr (L > 0) {
                        constexpr int LP = L - 1;
                        constexpr LDesc DP = layer_desc(LP);
                        constexpr bool HO17p = LP == 0 || LP == 1 || LP == 9, HOp = LP == 3 || LP == 5 || LP == 7;
                        constexpr bool HORp = LP == 4 || LP == 6, HOCp = LP == 2 || LP == 8;
                        constexpr int TNEXTp = (TF * (LP == 4 ? 10 : LP == 8 ? 17 : 12) + 16) * 36;
                        constexpr int CI = DP.cout, VI = DP.V, CSP = ss_of(CI);
                        float* xprev = const_cast<float*>(xin);
                        for (int u = tid; u < NB * CI * T * VI; u += NTHREADS) {
                            const int v = u % VI, t = (u / VI) % T, c = (u / (VI * T)) % CI, i = u / (VI * T * CI);
                            const float val = P.lt_in[(((size_t)b_of(i) * CI + c) * T + t) * VI + v];
                            const int gcol = (i * TP + t) * VI + v;
                            if (HO17p || (HOp && c < 32)) RA[gcol * 36 + c] = val;
                            else xprev[(size_t)gcol * CSP + c] = val;
                            if (HORp && c < 32) RA[TNEXTp + gcol * 36 + c] = val;
                            if (HOCp && gcol < TL_FC * VI) RA[TNEXTp + gcol * 36 + c] = val;
                            if (HOCp && EXF > 0 && gcol >= TL_FC * VI && gcol < 2 * TL_FC * VI) EX[(gcol - TL_FC * VI) * 36 + c] = val;
                        }
                        if (skip) {      // the U-Net skip tensor behind the fused resampler (d2 / d1; zeros when the caller gave none)
                            float* sk = const_cast<float*>(skip);
                            for (int u = tid; u < NB * CIN * T * V; u += NTHREADS) {
                                const int v = u % V, t = (u / V) % T, c = (u / (V * T)) % CIN, i = u / (V * T * CIN);
                                sk[(size_t)((i * TP + t) * V + v) * CSI + c] =
                                    P.lt_skip ? P.lt_skip[(((size_t)b_of(i) * CIN + c) * T + t) * V + v] : 0.f;
                            }
                        }
                        __syncthreads();
                    }
                }
                // (profile builds: the X staging of layers 3 .. 9 split further -- slot 40 + 2 (L - 3): the wait for the other waves'
                // previous stage; + 1: the wait for the loads issued a stage ahead + their LDS stores)
#define TLXMARK(k) do { if constexpr (L >= 3) TLMARK(40 + 2 * (L - 3) + (k)); } while (0)
                float* const XA = RA;                                          // [ROWS + 16][CSV]
                float* const ZA = RA + (ROWS + 16) * (L <= 1 ? 36 : CSZ);      // [ROWSG + 16][CSZ]  (layer 0: behind the next layer's X)
                TlStage<ROWS, CINV> sx_own;            // plain input: a 32-channel part of all frames; resampled input: the skip rows
                auto& sx = [&]() -> auto& { if constexpr (SKPRE && L == 7) return sk7; else if constexpr (SKPRE && L == 9) return sk9; else return sx_own; }();
                constexpr int IR = TL_FC * VIN, OR = TL_FC * V;
                // resampled input: a chunk of the resampler's input rows -- or, behind a layer that hands its output over in LDS (HIR: the
                // z region holds the resampler's input of ALL frames), every later 32-channel part whole as well: its loads are issued a
                // whole part ahead, where a part's second chunk was fetched under the first chunk's 300-cycle resampling (its latency
                // to the slab exposed, plus two barriers per chunk)
                constexpr bool ONE = HIR && MCD_TL_ONECHUNK;
                TlStage<RSI >= 0 ? (ONE ? NFC * IR : IR) : 1, 32> si;
                RsCoef<32, VIN, V, TL_FC, 1, false> rc_own;
                auto& rc = [&]() -> auto& {
                    if constexpr (RCPRE && L == 3) return rcx3; else if constexpr (RCPRE && L == 5) return rcx5;
                    else if constexpr (RCPRE && L == 7) return rcx7; else if constexpr (RCPRE && L == 9) return rcx9; else return rc_own;
                }();
                if constexpr (RSI >= 0) {
                    static_assert(!HIR || NH >= 2, "");
                    static_assert(!HIC || (NH == 1 && NFC == 2), "");
                    // Vector-memory loads return IN ORDER: a wait for any load also waits for every older one.  The slab loads are the
                    // slow ones (the slabs of an XCD's workgroups are 13 MB against 4 MB of L2), so nothing that is needed early may
                    // be issued behind them: the coefficient fetches go first, and (MCD_TL_LATE) a later part's slab loads only go out
                    // right before the current part's channel GEMM -- the mix stages fetch their later units' coefficients in place
                    // and wait for them with vmcnt(0); issued in front of the mixes, as they were, the slab loads' whole latency sat
                    // in every time mix.
                    if constexpr (MCD_TL_LATE && !RCPRE) rc.load(wb + Nl->rsw[RSI], wb + Nl->rsw[RSI] + ((V + 15) / 16) * ((VIN + 3) / 4) * 64, lane);
                    static_assert(!HIC2 || (IR + 4) * 36 <= EXF, "");
                    if constexpr (HIC2) { }                                                   // (both chunks are in LDS already)
                    else if constexpr (HIC) si.issue(tid, xin + (size_t)IR * CSI, CSI, 0);      // (chunk 0 is in the z region already)
                    else if constexpr (!(MCD_TL_LATE && HIR)) si.issue(tid, xin, CSI, HIR ? CINV : 0);   // (HIR: part 0 is in the z region already, all chunks of it)
                    if constexpr (!MCD_TL_LATE && !RCPRE) rc.load(wb + Nl->rsw[RSI], wb + Nl->rsw[RSI] + ((V + 15) / 16) * ((VIN + 3) / 4) * 64, lane);
                    static_assert(RSI < 0 || HIC || HIR, "");
                } else if (!xin_lds && !HI17) {
                    static_assert(!HI || (RSI < 0 && NH >= 2), "");
                    static_assert(!MCD_TL_LATE || HI || HI17 || L == 0, "a plain layer's first part comes through LDS (hand-over), the later ones are fetched behind the GEMMs");
                    if constexpr (!MCD_TL_LATE) sx.issue(tid, xin, CSI, HI ? CINV : 0);
                }
                float tqa[TP / 4], aja[(V + 15) / 16][(V + 3) / 4];     // the first units' mix coefficients, a stage ahead
                if constexpr (TQPRE && L > 0) {
#pragma unroll
                    for (int i = 0; i < TP / 4; ++i) tqa[i] = tqx[i];
                } else {
                    tl_time_fetch<V, TP, NB, FS>(tqa, wb + Nl->tqm[L], wave, lane, 0);
                }
                auto next_tq = [&] {      // (called in front of the layer's last channel GEMM)
                    if constexpr (RCPRE && (L == 2 || L == 4 || ((L == 6 || L == 8) && !COND))) {
                        constexpr int RN = L / 2 - 1;                         // the next layer's resampler: down1, down2, up3, up2
                        constexpr int VI = RN == 0 ? 17 : RN == 2 ? 10 : 12, VO = layer_desc(L + 1).V;
                        auto& rn = [&]() -> auto& { if constexpr (L == 2) return rcx3; else if constexpr (L == 4) return rcx5; else if constexpr (L == 6) return rcx7; else return rcx9; }();
                        rn.load(wb + Nl->rsw[RN], wb + Nl->rsw[RN] + ((VO + 15) / 16) * ((VI + 3) / 4) * 64, lane);
                    }
                    if constexpr (TQPRE && L + 1 <= (COND ? 6 : 9)) {
                        constexpr LDesc DN = layer_desc(L + 1);
                        tl_time_fetch<DN.V, TP, NB, tl_ngrp(DN.V)>(tqx, wb + Nl->tqm[L + 1], wave, lane, 0);
                    }
                };
                if constexpr (RSI >= 0 && !(SKPRE && (L == 7 || L == 9))) { if (skip) sx.issue(tid, skip, CSI, 0); }
                // weight fragments of the wave's m-tile: all of them up front, or (128 input channels) a quarter at a time
                constexpr bool AQ = NH > 2;
                constexpr int KQA = (CIN / 16) * (RES ? 2 : 1);
                const int mt = wave % MT, ng = MT > NWAVES ? 0 : (wave / MT < TI::NG ? wave / MT : NT), c0 = mt * 16 + 4 * (lane >> 4);
                // Layer 5 (64 -> 128) on TWELVE waves: its 8 m-tiles gave 8 waves one m-tile x ALL n-tiles each -- 60 / 80 accumulator
                // registers per lane at 24 / 32 frames, held across the staging and the mixes of the second 32-channel part, i.e. parked
                // in scratch there (16 / 111 spilled registers, every reload an in-order load in the middle of a stage), and four
                // waves without a tile.  W2: 4 m-tile pairs x 3 n-thirds -- every wave the m-tiles (wave % 4) and (wave % 4) + 4 on
                // the n-tiles wave / 4 + 3 i: 2 x 5 / 2 x 7 accumulators, the part's weight fragments (2 x 4 float4) fetched per part.
                // 32 frames: 106 -> 58 spilled registers, 0.473 .. 0.517 run to run -> a steady 0.518 .. 0.520 (the spill traffic made
                // the kernel erratic; profiles/r05zi_tiled32_l5w2_ab2.txt).  24 frames: 16 -> 0 spilled registers and -1.1 %: off there.
                constexpr bool W2 = MCD_TL_L5_W2 && TF == 32 && L == 5 && MT == 8 && NWAVES == 12 && FS == 1 && RES && NH == 2;
                using TI2 = Tiling<4, NT>;
                static_assert(!W2 || (TI2::NG == 3 && !HOR && !HOC && !HO17), "");
                float4 aq2[W2 ? 2 : 1][W2 ? 2 * KH : 1];
                float4 bq2[2];
                f32x4 acc2[W2 ? 2 : 1][W2 ? TI2::MAXN : 1];
                const int mp2 = wave & 3, ng2 = wave >> 2;
                LayerAfr<(AQ || W2) ? 1 : KQA> A;
                float4 aq[AQ ? 2 * KH : 1];
                const float* wfr = wb + Nl->wp[L] + ((size_t)mt * KQA * 64 + lane) * 4;
                if constexpr (W2) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) bq2[s2] = load_global4(wb + Nl->bias[L] + (mp2 + 4 * s2) * 16 + 4 * (lane >> 4));
                } else if constexpr (!AQ) {
                    LayerW lw;
                    lw.wp = Nl->wp[L]; lw.bias = Nl->bias[L];
                    A.template load<MT>(wb, lw, wave, lane);
                } else {
                    A.bcur = load_global4(wb + Nl->bias[L] + c0);
                }
                const float slope = Nl->slope[L], pinf = prelu_bound(slope);
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
                        } else if constexpr (ONE) {
                            __syncthreads();              // (the previous part is done with XA and the z region)
                            TLXMARK(0);
                            si.commit(tid, ZA, CSZ);
                            TLXMARK(1);
                            __syncthreads();
                            if constexpr (h + 1 < NH && !MCD_TL_LATE) si.issue(tid, xin, CSI, (h + 1) * CINV);
#pragma unroll
                            for (int fc = 0; fc < NFC; ++fc)
                                resample_stage<32, VIN, V, TL_FC, 1, false, false, true>(ZA + fc * IR * CSZ, CSZ, XA + fc * OR * CSV, CSV, rc, nosk, wave, lane);
                        } else if constexpr (HIC2) {
                            resample_stage<32, VIN, V, TL_FC, 1, false, false, true>(ZA, CSZ, XA, CSV, rc, nosk, wave, lane);
                            resample_stage<32, VIN, V, TL_FC, 1, false, false, true>(EX, 36, XA + OR * CSV, CSV, rc, nosk, wave, lane);
                        } else {
#pragma unroll
                            for (int fc = 0; fc < NFC; ++fc) {
                                if (!(HIC && fc == 0)) {
                                    __syncthreads();      // (the previous stage / part / chunk is done with XA and the chunk region)
                                    TLXMARK(0);
                                    si.commit(tid, ZA, CSZ);
                                    TLXMARK(1);
                                    __syncthreads();
                                    if (!HIC && fc + 1 < NFC) si.issue(tid, xin + (size_t)(fc + 1) * IR * CSI, CSI, h * CINV);
                                    else if constexpr (h + 1 < NH && !MCD_TL_LATE) si.issue(tid, xin, CSI, (h + 1) * CINV);
                                }
                                resample_stage<32, VIN, V, TL_FC, 1, false, false, true>(ZA, CSZ, XA + fc * OR * CSV, CSV, rc, nosk, wave, lane);
                            }
                        }
                        if (skip) {                       // + the U-Net skip (d2 / d1), this part's channels
                            __syncthreads();
                            TLMARK(4 * L);
#pragma unroll
                            for (int i = 0; i < std::remove_reference_t<decltype(sx)>::N; ++i) {
                                const int u = tid + i * NTHREADS;
                                if (u < ROWS * 8) {
                                    float4* xp = reinterpret_cast<float4*>(XA + (u >> 3) * CSV + (u & 7) * 4);
                                    const float4 a = *xp, b = sx.v[i];
                                    *xp = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
                                }
                            }
                            TLXMARK(1);
                            if constexpr (h + 1 < NH && !MCD_TL_LATE) sx.issue(tid, skip, CSI, (h + 1) * CINV);
                        }
                        Xl = XA;
                    } else if (HI17) {
                        Xl = XA;                          // (the previous layer's epilogue left it there)
                    } else if (!xin_lds) {
                        if constexpr (!(HI && h == 0)) {  // (HI: part 0 is in XA already, part 1 on its way)
                            __syncthreads();              // (the previous stage / half is done with XA)
                            TLXMARK(0);
                            sx.commit(tid, XA, CSV);
                            TLXMARK(1);
                            if constexpr (h + 1 < NH && !MCD_TL_LATE) sx.issue(tid, xin, CSI, (h + 1) * CINV);
                        }
                        Xl = XA;
                    }
                    if constexpr (W2) {
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const float* w2 = wb + Nl->wp[L] + ((size_t)(mp2 + 4 * s2) * KQA * 64 + lane) * 4;
#pragma unroll
                            for (int k = 0; k < KH; ++k) {
                                aq2[s2][k] = load_global4(w2 + (h * KH + k) * 256);
                                aq2[s2][KH + k] = load_global4(w2 + (CIN / 16 + h * KH + k) * 256);
                            }
                        }
                    }
                    if constexpr (AQ) {
                        static_assert(!AQ || RES, "");
#pragma unroll
                        for (int k = 0; k < KH; ++k) {
                            aq[k] = load_global4(wfr + (h * KH + k) * 256);
                            aq[KH + k] = load_global4(wfr + (CIN / 16 + h * KH + k) * 256);
                        }
                    }
                    TLTR(180 + L);
                    __syncthreads();
                    TLMARK(4 * L);
                    auto gemm_fg = [&](int fg, f32x4 (&ac)[TI::MAXN]) {
                        const float* xg = Xl + fg * ROWSG * CSV;
                        if constexpr (W2) {
                            gemm_part<4, NT, KH, KH, 0, KH, h == 0>(aq2[0], ZA, CSZ, xg, CSV, wave, lane, acc2[0]);
                            gemm_part<4, NT, KH, KH, 0, KH, h == 0>(aq2[1], ZA, CSZ, xg, CSV, wave, lane, acc2[1]);
                            return;
                        }
                        // (part 0 starts the accumulators: gemm_part's FIRST)
                        if constexpr (AQ) gemm_part<MT, NT, KH, KH, 0, KH, h == 0>(aq, ZA, CSZ, xg, CSV, wave, lane, ac);
                        else if constexpr (RES) gemm_part<MT, NT, KH, KH, h * KH, CIN / 16 + h * KH, h == 0>(A.a, ZA, CSZ, xg, CSV, wave, lane, ac);
                        else gemm_part<MT, NT, KH, 0, h * KH, 0, h == 0>(A.a, ZA, CSZ, xg, CSV, wave, lane, ac);
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
                    auto epi_core = [&](auto tgc, int fg, const int mt, const int ng, const float4 bcur, const auto& ac) {
                        using TG = typename decltype(tgc)::type;
                        const int c0 = mt * 16 + 4 * (lane >> 4);
                        static_for<TG::MAXN>([&](auto ii) {
                            constexpr int i = decltype(ii)::value;
                            const int col = ng * 16 + (lane & 15) + i * TG::NG * 16;
                            if (ng + i * TG::NG < NT && col < ROWSG) {
                                const int gcol = fg * ROWSG + col;
                                float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
                                if constexpr (!COND) e = *reinterpret_cast<const float4*>(EMB + (NB > 1 ? gcol / (TP * V) : 0) * EMBS + emb_off(L) + c0);
                                const f32x2 t0 = f32x2{ac[i][0] + bcur.x, ac[i][1] + bcur.y}, t1 = f32x2{ac[i][2] + bcur.z, ac[i][3] + bcur.w};
                                const f32x2 m0 = t0 * slope, m1 = t1 * slope;
                                const float4 o = make_float4(__builtin_amdgcn_fmed3f(t0[0], m0[0], pinf) + e.x, __builtin_amdgcn_fmed3f(t0[1], m0[1], pinf) + e.y,
                                                             __builtin_amdgcn_fmed3f(t1[0], m1[0], pinf) + e.z, __builtin_amdgcn_fmed3f(t1[1], m1[1], pinf) + e.w);
                                if (HO17 || (HO && mt < 2)) *reinterpret_cast<float4*>(XA + gcol * 36 + c0) = o;
                                else store_global4(xout + (size_t)gcol * CSO + c0, o);      // (layer 4's is the skip d2 as well)
                                if (HOR && mt < 2) *reinterpret_cast<float4*>(RA + TNEXT + gcol * 36 + c0) = o;
                                if (HOC && gcol < TL_FC * V) *reinterpret_cast<float4*>(RA + TNEXT + gcol * 36 + c0) = o;
                                if (HOC2 && gcol >= TL_FC * V) *reinterpret_cast<float4*>(EX + (gcol - TL_FC * V) * 36 + c0) = o;
                            }
                        });
                    };
                    auto epi_fg = [&](int fg, const f32x4 (&ac)[TI::MAXN]) {
                        if constexpr (W2) {
                            epi_core(TlType<TI2>{}, fg, mp2, ng2, bq2[0], acc2[0]);
                            epi_core(TlType<TI2>{}, fg, mp2 + 4, ng2, bq2[1], acc2[1]);
                        } else {
                            epi_core(TlType<TI>{}, fg, mt, ng, A.bcur, ac);
                        }
                    };
                    if constexpr (HO17 || L == 2) {
                        static_assert(!(HO17 || L == 2) || (FS == 2 && NH == 1 && COUT <= 32), "");
                        f32x4 acc0[TI::MAXN];
                        tl_joint_fetch<V, TP, NB, FS>(aja, wb + Nl->am[L], wave, lane, 0);
                        tl_time_mix<CINV, V, TP, NB, FS>(Xl, CSV, ZA, CSZ, wb + Nl->tqm[L], wave, lane, 0, tqa);
                        tl_time_fetch<V, TP, NB, FS>(tqa, wb + Nl->tqm[L], wave, lane, 1);
                        __syncthreads();
                        tl_joint_mix<CINV, V, TP, NB, FS>(ZA, CSZ, wb + Nl->am[L], wave, lane, 0, aja);
                        TLMARK(4 * L + 1);
                        __syncthreads();
                        TLMARK(4 * L + 2);
                        gemm_fg(0, acc0);
                        tl_joint_fetch<V, TP, NB, FS>(aja, wb + Nl->am[L], wave, lane, 1);
                        TLMARK(4 * L + 3);
                        __syncthreads();                  // (z is free)
                        tl_time_mix<CINV, V, TP, NB, FS>(Xl, CSV, ZA, CSZ, wb + Nl->tqm[L], wave, lane, 1, tqa);
                        __syncthreads();                  // (nobody reads the X rows of group 0 any more)
                        if constexpr (HO17) epi_fg(0, acc0);
                        tl_joint_mix<CINV, V, TP, NB, FS>(ZA, CSZ, wb + Nl->am[L], wave, lane, 1, aja);
                        TLMARK(4 * L + 1);
                        __syncthreads();
                        TLMARK(4 * L + 2);
                        next_tq();
                        gemm_fg(1, acc);
                        __syncthreads();                  // (... nor those of group 1)
                        if constexpr (!HO17) epi_fg(0, acc0);
                        epi_fg(1, acc);
                        TLMARK(4 * L + 3);
                    } else {
#pragma unroll
                        for (int fg = 0; fg < FS; ++fg) {
                            tl_joint_fetch<V, TP, NB, FS>(aja, wb + Nl->am[L], wave, lane, fg);
                            tl_time_mix<CINV, V, TP, NB, FS>(Xl, CSV, ZA, CSZ, wb + Nl->tqm[L], wave, lane, fg, tqa);
                            if (fg + 1 < FS || h + 1 < NH) tl_time_fetch<V, TP, NB, FS>(tqa, wb + Nl->tqm[L], wave, lane, fg + 1 < FS ? fg + 1 : 0);
                            TLTR(100 + L);
                            __syncthreads();
                            TLTR(120 + L);
                            tl_joint_mix<CINV, V, TP, NB, FS>(ZA, CSZ, wb + Nl->am[L], wave, lane, fg, aja);
                            TLMARK(4 * L + 1);
                            __syncthreads();
                            TLMARK(4 * L + 2);
                            if constexpr (MCD_TL_LATE && h + 1 < NH) {      // the next part's slab loads: in flight behind this part's GEMM
                                if (fg == FS - 1) {
                                    if constexpr (RSI >= 0) {
                                        si.issue(tid, xin, CSI, (h + 1) * CINV);
                                        if (skip) sx.issue(tid, skip, CSI, (h + 1) * CINV);
                                    } else if (!xin_lds && !HI17) {
                                        sx.issue(tid, xin, CSI, (h + 1) * CINV);
                                    }
                                }
                            }
                            if constexpr (h == NH - 1) { if (fg == FS - 1) next_tq(); }
                            if constexpr (SKPRE && h == NH - 1 && (L == 6 || L == 8)) {
                                if (fg == FS - 1) {
                                    if constexpr (L == 6) sk7.issue(tid, D2, ss_of(64), 0);
                                    else sk9.issue(tid, D1, ss_of(32), 0);
                                }
                            }
                            gemm_fg(fg, acc);
                            if constexpr (h == NH - 1) {
                                static_assert(!(HO || HOR) || (FS == 1 && CSV == 36), "");
                                TLTR(140 + L);
                                if constexpr (HO || HOR || L == 8) __syncthreads();    // (every wave is done with XA / z: the m-tiles 0, 1 go there)
                                TLTR(160 + L);
                                epi_fg(fg, acc);
                            }
                            if (fg + 1 < FS) __syncthreads();  // (the next group's mix overwrites z)
                            TLMARK(4 * L + 3);
                        }
                    }
                });
                __syncthreads();
                if constexpr (LT) {      // the stage's output, from where its epilogue put it
                    for (int u = tid; u < NB * COUT * T * V; u += NTHREADS) {
                        const int v = u % V, t = (u / V) % T, c = (u / (V * T)) % COUT, i = u / (V * T * COUT);
                        const int gcol = (i * TP + t) * V + v;
                        const float val = (HO17 || (HO && c < 32)) ? RA[gcol * 36 + c] : xout[(size_t)gcol * CSO + c];
                        if (grp * NB + i < P.n_chains) P.lt_out[(((size_t)b_of(i) * COUT + c) * T + t) * V + v] = val;
                    }
                    __syncthreads();
                }
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
            if constexpr (COND) {
                // to_time_dim: emb[window][j] = b[j] + sum_k W[j][k] H[k], k = (c * T + t) * 10 + v, H = layer 6's 6 channels in the slab
                __syncthreads();
                int tid = tid0;
                asm volatile("" : "+v"(tid));
                const int F = CU_OUT_TL * T * 10;
                const float* Wl = wb + Ns->we;
                for (int u = tid >> 5; u < NB * EDIM; u += NTHREADS / 32) {        // 32 lanes per (chain, output)
                    const int i = u / EDIM, jo = u % EDIM, part = tid & 31;
                    float a = 0.f;
                    for (int k = part; k < F; k += 32) {
                        const int c = k / (T * 10), r = k % (T * 10);                // r = t * 10 + v
                        a = fmaf(Wl[(size_t)jo * F + k], A1[(size_t)(i * TP * 10 + r) * ss_of(16) + c], a);
                    }
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 32);
                    if (part == 0 && grp * NB + i < P.n_chains) P.eps_out[(size_t)b_of(i) * EDIM + jo] = a + wb[Ns->be + jo];
                }
            } else {
            layer(TL_C(7), TL_C(2), A1, false, A0, D2);                     // up3 + d2 on the way in
            layer(TL_C(8), TL_NORS, A0, false, A1, nullptr);
            layer(TL_C(9), TL_C(3), A1, false, A0, D1);                     // up2 + d1 on the way in
            }
            TLMARK(61);
            if (!COND && (!LT || P.lt_stage == 10)) {   // ---- layer 10 (32 -> 2) W-first on plain FMAs: P4[col][r] = sum_k W4[r][k] X[col][k]  (P_t 0,1 ; P_r 2,3)
                int tid = tid0;
                asm volatile("" : "+v"(tid));
                const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                if constexpr (LT) {      // layer 9's output as its epilogue hands it over: the X region, row stride 36
                    for (int u = tid; u < NB * 32 * T * 17; u += NTHREADS) {
                        const int v = u % 17, t = (u / 17) % T, c = (u / (17 * T)) % 32, i = u / (17 * T * 32);
                        RA[((i * TP + t) * 17 + v) * 36 + c] = P.lt_in[(((size_t)b_of(i) * 32 + c) * T + t) * 17 + v];
                    }
                    __syncthreads();
                }
#ifndef MCD_TL_QC10
#define MCD_TL_QC10 2
#endif
                // (16 frames: two output frames per unit -- 8 units, every wave busy -- instead of four in 4 units)
                // (24 frames on twelve waves: two output frames per unit as well -- 12 units, every wave busy -- instead of three in 8)
#ifndef MCD_TL_QC10_24
#define MCD_TL_QC10_24 0      // (2: 12 units of two frames -- measured 0, profiles/r05z_tiled24_qc10_ab.txt)
#endif
                constexpr int QC10 = TF <= 16 ? MCD_TL_QC10 : (TF == 24 && NWAVES == 12) ? MCD_TL_QC10_24 : 0;
                MixLongCoef<16, 17, TP, NB, QC10> mc10;      // (the mix's first coefficients: in flight behind the product)
                mc10.load(wb + Ns->tq[10], wb + Ns->am[10], wave, lane);
                // P4[col][0 .. 3] = [W_t; W_r] (4 x 32) . X[col][0 .. 31].  Round 4 ran it as plain FMAs per column (a 16-row MFMA tile is 3/4
                // padding): with the weights read through the laundered (generic) weight pointer that was 32 serial flat loads per column,
                // 5 % of a pass; from LDS at wave-uniform addresses it is LDS-bound (40 ds_read_b128 per column on 7 of the 12 waves:
                // 1.6 % of a pass).  On the matrix cores the padding is free -- 26 tiles x 8 MFMAs at 24 frames -- and a tile costs two
                // B-operand reads: A = the weights' rows 0 .. 3 (zero rows below), K in the order (16 jj + 4 g + r) for both operands.
#if MCD_TL_L10_MFMA
                {
                    const int j = lane & 15, g = lane >> 4;
                    float4 af[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        af[jj] = j < 4 ? *reinterpret_cast<const float4*>(W4L + j * 32 + 16 * jj + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
                    constexpr int NT10 = (R17 + 15) / 16;
                    for (int nt = wave; nt < NT10; nt += NWAVES) {
                        const float* xp = RA + (nt * 16 + j) * 36 + 4 * g;      // layer 9's output, handed over in LDS (pad rows: finite)
                        const float4 x0 = *reinterpret_cast<const float4*>(xp), x1 = *reinterpret_cast<const float4*>(xp + 16);
                        f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0].x, x0.x, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0].y, x0.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0].z, x0.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0].w, x0.w, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1].x, x1.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1].y, x1.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1].z, x1.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1].w, x1.w, acc, 0, 0, 0);
                        if (g == 0 && nt * 16 + j < R17) *reinterpret_cast<float4*>(P4 + (nt * 16 + j) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                    }
                }
#else
                for (int col = tid; col < R17; col += NTHREADS) {
                    const float* xp = RA + col * 36;          // layer 9's output, handed over in LDS
                    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const float4 x = *reinterpret_cast<const float4*>(xp + 4 * q);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float4 w = *reinterpret_cast<const float4*>(W4L + r * 32 + 4 * q);
                            a[r] = fmaf(w.x, x.x, a[r]); a[r] = fmaf(w.y, x.y, a[r]);
                            a[r] = fmaf(w.z, x.z, a[r]); a[r] = fmaf(w.w, x.w, a[r]);
                        }
                    }
                    *reinterpret_cast<float4*>(P4 + col * 4) = make_float4(a[0], a[1], a[2], a[3]);
                }
#endif
                __syncthreads();
                TLMARK(55);                            // layer 10: W-first product
                // its 2-channel mix (16-channel block view of P4: channels 2..15 are the next columns' values, never stored)
                mix_long<16, 17, TP, NB, 1, QC10>(P4, 4, mc10, wb + Ns->tq[10], wb + Ns->am[10], wave, lane, ZeroInitL{},
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
                TLMARK(56);                            // layer 10: mix
                // eps = layer 10 + x; DDPM update of the frame each prediction drives (mocodad.py:172-178,829-838)
                const float slope10 = Ns->slope[10], ca = srow[0], cb = srow[1], csg = srow[2];
                const float b10[C0] = {as_global(wb)[Ns->bias[10]], as_global(wb)[Ns->bias[10] + 1]};
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
                        const float l10 = prelu(ZO[u] + P4[col * 4 + C0 + c] + (c ? b10[1] : b10[0]), slope10) + EMB[i * EMBS + emb_off(10) + c];
                        const float eps = l10 + XT[col * 4 + c];
                        if constexpr (LT) {      // layer 10 alone: without the U-Net's residual (+ x)
                            if (t < T && grp * NB + i < P.n_chains) P.lt_out[(((size_t)b_of(i) * C0 + c) * T + t) * 17 + v] = l10;
                        }
                        if (P.mode == 1) {
                            if (t < T && P.eps_out && grp * NB + i < P.n_chains) P.eps_out[(((size_t)b_of(i) * C0 + c) * T + t) * 17 + v] = eps;
                            continue;
                        }
                        const int tpm = UPDT[t];                             // (LDS: M.upd_of[t] -> M.pos_of[k] were two dependent global loads per element)
                        const int k = t >= T ? -1 : P.win_mask ? (((fixed_of(i) >> t) & 1u) ? -1 : 0) : tpm;
                        if (k >= 0) {
                            const int tp = P.win_mask ? t : tpm;
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
        if (COND || P.mode == 1) continue;
        // ---- loss over the corrupt frames (mocodad.py:484)
        bool diverged = false;
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
            // (tree over the next power of two: a workgroup of 12 waves has 768 threads; at 512 the order is the plain halving)
            constexpr int RP2 = NTHREADS <= 512 ? 512 : 1024;
            for (int o = RP2 / 2; o > 0; o >>= 1) { if (tid < o && tid + o < NTHREADS) RED[tid] += RED[tid + o]; __syncthreads(); }
            const float lsum = RED[0];                 // (every thread: the same LDS word)
            if (tid == 0) P.loss_out[chain] = lsum / (float)per;
            diverged |= !(fabsf(lsum) <= 3.0e38f);      // NaN / Inf
            __syncthreads();
        }
        // A diverged chain (NaN / Inf activations) leaves non-finite values in the slab and the LDS regions; the NEXT chain of this
        // workgroup -- another window, or another sample -- reads some of them as pad rows / pad frames against zero coefficients
        // (NaN x 0 = NaN) and would come out NaN as well, where the reference's chains are independent (mocodad.py:155-180).
        // Everything is cleared again then; the common path pays a compare per chain.
        if (diverged) {
            int t_c = tid0;
            asm volatile("" : "+v"(t_c));
            clear_state(t_c);
            __syncthreads();
        }
    }
}


}  // namespace mcd
