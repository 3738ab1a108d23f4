// mcd_score_kernel.hpp — the persistent trajectory kernel score_kernel<T_u, NB, MINW> (1 .. 12 U-Net frames) and the MFMA
// condition encoders cond_fast_kernel / cond_unet_kernel built from the same stage functions (see mcd_device.hpp).
#pragma once
#include "mcd_device.hpp"

// kernels with at most this many waves per SIMD get the latency-side forms (two GEMM tiles in flight, pinned read-ahead, resampler
// units advancing together, coefficient fetches two stages ahead): 2 = the shapes without a register cap; 3 with the 12-wave tuning build
#ifndef MCD_LOWOCC
#define MCD_LOWOCC (MCD_NWAVES == 12 ? 3 : 2)
#endif
#ifndef MCD_NO_EARLY2
#define MCD_NO_EARLY2 0
#endif
#ifndef MCD_RS_ILP
#define MCD_RS_ILP 1
#endif

namespace mcd {

// ------------------------------------------------------------------------------------------------
// condition encoder, fast path for the shipped architecture (channels [32,16,32] + h_dim 32, latent 16):
// the same MFMA mix / GEMM stages as the U-Net, NB windows per 512-thread workgroup, followed by the
// bottleneck Linear over the (c,t,v) flattening (stsae.py:73-89).  Reads the condition frames straight from the
// window tensor (no gather pass).  Other channel lists use cond_encode_kernel below.
// ------------------------------------------------------------------------------------------------
constexpr int TABC = 128;                  // cond table: second 128 words of the weight buffer
constexpr int TABC_LW = 40, TABC_LB = 41;  // bottleneck Linear weight [16][32*T*17] / bias

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
template <int T, int NB, int MINW, bool LT = false>
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
    if (P.phase < -15) {    // tuning experiment: every workgroup starts at its own (hashed) offset of 0 .. 63 x |phase| x 16 cycles
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
        WM[threadIdx.x] = P.win_mask ? P.win_mask[b] : (int)P.fixed_mask;
        WM[4 + threadIdx.x] = b;
        if (threadIdx.x == 0) { WM[8] = 0; WM[9] = -1; }      // [8] "the trajectory that just ended diverged", [9] the chain slot that runs ALONE (-1: all) -- see the end of the sample loop
    }
    // biases the W-first layers add inside their mix store functors: from LDS there, not from global memory (a global
    // load in a functor that also stores to LDS is re-issued per call: one L2 round trip per output row)
    if (threadIdx.x >= 128 && threadIdx.x < 128 + NB * T) {
        // the prediction at frame t drives corrupt frame k = upd_of[t], which lives at frame pos_of[k] (the same frame except
        // for 'concat' with the condition at the END of the window, where the reference reads the prediction at the corrupt
        // frames' ORIGINAL indices, mocodad.py:829-838); with per-window frame sets (random_imp) every clear bit updates itself
        const int i = threadIdx.x - 128, n = i / T, t = i % T;
        const int fixed = P.win_mask ? P.win_mask[window_of(n)] : (int)P.fixed_mask;
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
    if constexpr (MINW >= 4) {
        // the priority time slice of this launch (see the top of the step loop), kept in LDS: a sixth of the launch's duration --
        // measured by the previous launch of the same grid (its workgroup 0's lifetime x the rounds of workgroups; every
        // workgroup reads the same words, so the co-resident ones agree), else the host's estimate from the shapes (calibrated
        // at 2.4 GHz: on a power-capped node it is off, and the gain depends on the slice length, profiles/r03s_prio_slices.txt)
        if (threadIdx.x == 0) {
            int sh = P.prio_shift;
            if (sh > 0 && P.tune) {
                const int sig = (int)gridDim.x * 1009 + P.S * 31 + P.ns;
                const int ticks = P.tune[1];
                if (P.tune[0] == sig && ticks > 0) {
                    const unsigned v = (unsigned)(((unsigned long long)(unsigned)ticks * (unsigned)(P.prio_rounds > 0 ? P.prio_rounds : 1)) / 6u) | 1u;
                    sh = 31 - __clz((int)v);
                    if ((v >> (sh > 0 ? sh - 1 : 0)) & 1u) ++sh;        // (>= 1.5 x 2^sh: round to the nearest power of two)
                    sh = sh < 10 ? 10 : (sh > 26 ? 26 : sh);
                }
            }
            UPD[15] = sh;
            UPD[14] = (int)(unsigned)__builtin_amdgcn_s_memrealtime();
        }
    }
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
    else if (threadIdx.x >= 80 && threadIdx.x < 112) BIA[threadIdx.x] = P.wbuf[tab_i(P.wbuf, 8 * F_STRIDE + F_BIAS) + threadIdx.x - 80];
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
#define MCD_STASH 0      // (round 5: nothing is hand-parked any more; bit 0 / 1: d2 / d1 of the 6-frame kernel, +1.7 % without)
#endif
#ifndef MCD_T6_LOWO
#define MCD_T6_LOWO 0       // (round 6: off again -- with layer 8 W-first, the single-read mixes and the swapped-operand fragments in, the
#endif                      //  plain forms are +1.2 .. 1.4 % at 6 frames, profiles/r06j_switch_sweep_ab.txt; round 5 had measured +1.1 % WITH it)
#ifndef MCD_RELAUNDER_UP
#define MCD_RELAUNDER_UP 1      // 0: off, 1: the register-capped kernels that spilled (see the step loop), 7: every kernel (A/B)
#endif
    // LOWO: the stage forms of the kernels with registers to spare -- pinned X reads in the mixes (FORCE), the pipelined GEMM's
    // read-ahead, interleaved resampler units.  The 6-frame kernel joined them in round 5 (114 of its 128 registers once the
    // up path derives its addresses again, see the step loop): +1.1 %; 3 frames -5 %, 5 frames +0.3 % (profiles/r05u_lowocc4_ab.txt)
    constexpr bool LOWO = MINW <= MCD_LOWOCC || (MCD_T6_LOWO && T == 6 && NB == 1 && MINW >= 4);
    constexpr bool STASH2 = !LT && ((MINW >= 4 && ((T == 6 && (MCD_STASH & 1)) || (T == 3 && (MCD_STASH & 4)))) || (MINW == 3 && (MCD_STASH & 32)));
    constexpr bool STASH1 = !LT && ((MINW >= 4 && ((T == 6 && (MCD_STASH & 2)) || (T == 3 && (MCD_STASH & 8)))) || (MINW == 3 && (MCD_STASH & 16)));
    float stash1_mem[STASH1 ? RS1::PER * RS1::SK : 1];
    float stash2_mem[STASH2 ? RS2::PER * RS2::SK : 1];
    typedef float __attribute__((address_space(5))) priv_float;         // (explicit private address space: scratch_*, not flat_*)

    const int i_first = P.mode == 1 ? P.step_single : P.ns - 1;
    const int i_last = P.mode == 1 ? P.step_single : 1;
    // embeddings of the first pass; those of pass i-1 are computed during the last layer of pass i
    auto silu_row = [&](int step, int t_id) {      // SEN[n][k] = SiLU(pe(step)[k] + cond[window of chain n][k])
        if (t_id >= 0 && t_id < NB * EDIM) {
            const int solo = NB > 1 ? WM[9] : -1;       // (a slot that sits out a solo run: no condition, x = 0, no noise -- a benign chain)
            const float e = P.step_table[step * (4 + EDIM) + 4 + (t_id % EDIM)] + ((solo >= 0 && t_id / EDIM != solo) ? 0.f : CE[t_id]);
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
            const int solo = NB > 1 ? WM[9] : -1;
            float xv[C0];
#pragma unroll
            for (int c = 0; c < C0; ++c) {
                if (solo >= 0 && n != solo) {
                    xv[c] = 0.f;
                } else if (mode == 1) {
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
#ifndef MCD_WEARLY
#define MCD_WEARLY 1
#endif
    constexpr bool WEARLY = T <= 4 && MCD_WEARLY;
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
            const int sh = UPD[15];
            if (sh > 0) {
                unsigned hwid;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                const unsigned slice = (unsigned)(__builtin_amdgcn_s_memrealtime() >> sh);
                prof.pp.slice = (int)(((hwid >> 16) ^ slice) & 1u);
                if constexpr (MCD_PHPRIO == 0) {
                    if (prof.pp.slice) __builtin_amdgcn_s_setprio(1);
                    else __builtin_amdgcn_s_setprio(0);
                }
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
                const int solo = NB > 1 ? WM[9] : -1;
                if (!((fixed >> t) & 1) && (solo < 0 || n == solo)) {
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
#ifndef MCD_EARLY2_ALL
#define MCD_EARLY2_ALL 0       // tuning: EARLY2 in every kernel
#endif
        constexpr bool EARLY2 = (LOWO && !MCD_NO_EARLY2) || T == 6 || MCD_EARLY2_ALL;
        LMix<1, T, NB> mc1;
        // layer 0 reads the chain state XT[col][4] in place (x in channels 0,1): its lanes' channels 2..15 are then other
        // columns' coordinates -- finite, and multiplied by the zero-padded K rows of the layer's weights -- so no 16-channel
        // copy of x has to be zeroed and rewritten every pass
        LAfr<1> A1; LAfr<2> A2; LAfr<3> A3; LAfr<4> A4; LAfr<5> A5; LAfr<7> A7; LAfr<8> A8; LAfr<9> A9;
        auto wearly = [&](auto& A, auto lc) { if constexpr (WEARLY) load_lafr<decltype(lc)::value>(A, wb, wave, lane); };
#define MCD_LC(l) std::integral_constant<int, l>{}
        layer_std<0, T, NB, LOWO, 4>(wb, mc0, XT, RG + PL::L0_z, RG + PL::L0_out, EMB, wave, lane, prof,
                            [&] { mix_early(mc1, 1); }, [&] { wearly(A1, MCD_LC(1)); }, WEARLY ? &A0 : nullptr);     // sp1a (2 -> 16)
        STAGE(2);
        lt_dump(0, RG + PL::L0_out, 20, 16, 17);
        lt_inject(1, RG + PL::L1_in, 20, 16, 17);
        // ---- down path
        LMix<2, T, NB> mc2;
        // SiLU(pe + cond) for the NEXT pass's embeddings (consumed in this pass's last layer): two global loads and an
        // exp -- by the idle waves too, not on wave 0's path at the top of the pass
        silu_row(sidx > 0 ? sidx - 1 : 0, tid - NZ_T0);
        layer_std<1, T, NB, LOWO>(wb, mc1, RG + PL::L1_in, RG + PL::L1_z, RG + PL::L1_out, EMB, wave, lane, prof,
                            [&] { mix_early(mc2, 2); }, [&] { wearly(A2, MCD_LC(2)); }, WEARLY ? &A1 : nullptr);     // sd1.0
        STAGE(3);
        lt_dump(1, RG + PL::L1_out, 36, 32, 17);
        lt_inject(2, RG + PL::L2_in, 36, 32, 17);
        RsCoef<32, 17, 12, T, NB, true> rc1;
        LMix<3, T, NB> mc3;
        layer_std<2, T, NB, LOWO>(wb, mc2, RG + PL::L2_in, RG + PL::L2_z, RG + PL::L2_out, EMB, wave, lane, prof,
                            [&] { rs_early(rc1, 0); }, [&] { if constexpr (EARLY2) mix_early(mc3, 3); }, WEARLY ? &A2 : nullptr);      // sd1.1 -> d1
        STAGE(4);
        lt_dump(2, RG + PL::L2_out, 36, 32, 17);
        lt_inject(11, RG + PL::L2_out, 36, 32, 17);
        if constexpr (!EARLY2) mix_early(mc3, 3);
        // (SQ12: down1 / up2 take their units in the order of layer 8's W-first mix, so that up2 follows that mix without a barrier)
        constexpr bool SQ12 = MCD_L8_WFIRST && !LT && MixCfg<32, 12, T, NB>::SAMEQ;
        resample_stage<32, 17, 12, T, NB, true, false, (LOWO && MCD_RS_ILP), SQ12>(RG + PL::L2_out, 36, RG + PL::DN1_out, 36, rc1, skip1, wave, lane);  // down1 (captures d1)
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
        layer_std<3, T, NB, LOWO>(wb, mc3, RG + PL::L3_in, RG + PL::L3_z, RG + PL::L3_out, EMB, wave, lane, prof,
                            [&] { mix_early(mc4, 4); }, [&] { wearly(A4, MCD_LC(4)); }, WEARLY ? &A3 : nullptr);     // sd2.0
        STAGE(6);
        lt_dump(3, RG + PL::L3_out, 68, 64, 12);
        lt_inject(4, RG + PL::L4_in, 68, 64, 12);
        RsCoef<64, 12, 10, T, NB, true> rc2;
        LMix<5, T, NB> mc5;
        layer_std<4, T, NB, LOWO>(wb, mc4, RG + PL::L4_in, RG + PL::L4_z, RG + PL::L4_out, EMB, wave, lane, prof,
                            [&] { rs_early(rc2, 1); }, [&] { if constexpr (EARLY2) mix_early(mc5, 5); }, WEARLY ? &A4 : nullptr);      // sd2.1 -> d2
        STAGE(7);
        lt_dump(4, RG + PL::L4_out, 68, 64, 12);
        lt_inject(12, RG + PL::L4_out, 68, 64, 12);
        if constexpr (!EARLY2) mix_early(mc5, 5);
        resample_stage<64, 12, 10, T, NB, true, false, (LOWO && MCD_RS_ILP)>(RG + PL::L4_out, 68, RG + PL::DN2_out, 68, rc2, skip2, wave, lane);  // down2 (captures d2)
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
            layer_std<5, T, NB, LOWO>(wb, mc5, RG + PL::L5_in, RG + PL::L5_z, RG + PL::L5_out, EMB, wave, lane, prof, nohook,
                                [&] {
                                    load_afrags<8, 8>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr, 0);
                                    if constexpr (EARLY2) mc6.load(wb + lw.tq, wb + lw.am, wave, lane);
                                }, WEARLY ? &A5 : nullptr);  // sd3.0
            STAGE(9);
            lt_dump(5, RG + PL::L5_out, 132, 128, 10);
            lt_inject(6, RG + PL::L6_in, 132, 128, 10);
            float* Pb = RG + PL::L6_p;
            prof.pp.thr();
            if constexpr (!EARLY2) mc6.load(wb + lw.tq, wb + lw.am, wave, lane);
            auto epi6 = [&](auto ti, int col, int c0, f32x4 acc, int col0, int) {
                constexpr int STEP = Tiling<8, NT>::NG * 16 * 132;
                if (col < COLS) *reinterpret_cast<float4*>(Pb + __mul24(col0, 132) + c0 + decltype(ti)::value * STEP) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            };
            gemm_tiles<8, NT, 8, 0, false, LOWO>(afr, RG + PL::L6_in, 132, RG + PL::L6_in, 132, wave, lane, epi6, 0);
#pragma unroll
            for (int mi = 1; mi < Tiling<8, NT>::MW; ++mi) {
                load_afrags<8, 8>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr, mi);
                gemm_tiles<8, NT, 8, 0, false, LOWO>(afr, RG + PL::L6_in, 132, RG + PL::L6_in, 132, wave, lane, epi6, mi);
            }
            if constexpr (EARLY2) { rs_early(rc3, 2); mix_early(mc7, 7); }      // up3's fragments, layer 7's mix coefficients
            bsync();
            prof.pp.lat();
            STAGE(10);
            const float slope6 = lw.slope;
            const float pinf6 = prelu_bound(slope6);
            mix_stage<64, 10, T, NB, LOWO, true>(Pb, 132, mc6, wb + lw.tq, wb + lw.am, wave, lane,
                                     [&](int n, int q, int w, ChIdx c) {   // P_r of joint w, the lane's 4 channels: one ds_read_b128
                                         // (address: the unit's part on the scalar unit + one v_mad for the lane's, see mix_stage;
                                         // joints >= 10 read the next frame's rows -- inside the 64-column region -- and are never stored)
                                         const float* pp = (Pb + (n * (T * 10) * 132 + q * (10 * 132) + 64 + c.cb16)) + (__mul24(w, 132) + c.j);
                                         const float4 r = lds_load4(lds_addr(pp));
                                         return f32x4{r.x, r.y, r.z, r.w};
                                     },
                                     [&](int n, int q, int w, ChIdx c, f32x4 v) {
                                         const float4 b4 = lds_load4(lds_addr(BIA + c.cb16) + 4u * (unsigned)c.j);
                                         const float4 e4 = lds_load4(lds_addr(EMB + n * EMB_STRIDE + emb_off(6) + c.cb16) + 4u * (unsigned)c.j);
                                         float* pp = (Pb + (n * (T * 10) * 132 + q * (10 * 132) + 64 + c.cb16)) + (__mul24(w, 132) + c.j);
                                         const f32x2 t0 = f32x2{v[0], v[1]} + f32x2{b4.x, b4.y}, t1 = f32x2{v[2], v[3]} + f32x2{b4.z, b4.w};
                                         const f32x2 m0 = t0 * slope6, m1 = t1 * slope6;
                                         const f32x2 r0 = f32x2{__builtin_amdgcn_fmed3f(t0[0], m0[0], pinf6), __builtin_amdgcn_fmed3f(t0[1], m0[1], pinf6)} + f32x2{e4.x, e4.y};
                                         const f32x2 r1 = f32x2{__builtin_amdgcn_fmed3f(t1[0], m1[0], pinf6), __builtin_amdgcn_fmed3f(t1[1], m1[1], pinf6)} + f32x2{e4.z, e4.w};
                                         lds_store4(lds_addr(pp), r0[0], r0[1], r1[0], r1[1]);
                                     });
        }
        if constexpr (STASH2) {
            const priv_float* sp = (const priv_float*)stash2_mem;
            asm volatile("" : "+v"(sp));
#pragma unroll
            for (int i = 0; i < RS2::PER * RS2::SK; ++i) skip2[i] = sp[i];
        }
        if constexpr (MCD_RELAUNDER_UP == 7 || (MCD_RELAUNDER_UP && (MINW == 3 || (MINW >= 4 && T >= 5)))) {
            // The register-capped kernels (twelve waves: 168 registers; two workgroups per CU: 128): per-lane LDS addresses of the
            // down path that the up path's stages compute again -- the B-operand address of the 17-joint GEMMs (layers 2 and 9), a
            // mix's store address (layers 1 and 8) -- were kept live across the 128-channel layers, i.e. SPILLED there (all 11
            // spilled registers of the 12-frame kernel, 20 at 10 / 11 frames, 11 at 6), and every reload is a vector-memory load:
            // returned in order, it drains the coefficient prefetches in flight (s_waitcnt vmcnt(0) in the middle of a stage).
            // Opaque from here on: the up path derives its addresses again, ~40 integer instructions per pass, and nothing is
            // spilled any more -- 12 / 10 / 6 / 5 frames +0.6 / +3.6 / +0.8 / +0.7 % (profiles/r05s_relaunder_up_ab.txt).  Not for the
            // kernels that had no spills inside the step loop (3 frames: -0.4 %; 7 / 8 frames, uncapped: 0).
            tid = tid0;
            asm volatile("" : "+v"(tid));
            lane = tid & 63;
            wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
        resample_stage<64, 10, 12, T, NB, false, true, (LOWO && MCD_RS_ILP)>(RG + PL::L6_p + 64, 132, RG + PL::UP3_out, 68, rc3, skip2, wave, lane);  // up3 (+ d2)
        wearly(A7, MCD_LC(7));
        // (up3 and layer 7's mix are both wave-aligned at 3 frames x 2 chains, and this barrier was removed in round 6 for +0.3 % --
        // and put back: layer 7's z region [s64b, 2 s64b) overlaps layer 6's P region, which slower waves are still READING in up3;
        // the golden tests passed, the determinism tests (same call twice, two streams) caught it.  The LDS plan has no room for a
        // z that avoids both; profiles/r06i_barrier_value_ab.txt.)
        bsync();
        STAGE(12);
        lt_dump(13, RG + PL::UP3_out, 68, 64, 12);
        lt_inject(7, RG + PL::L7_in, 68, 64, 12);
        LMix<8, T, NB> mc8;
        RsCoef<32, 12, 17, T, NB, false> rc4;
        LMix<9, T, NB> mc9;
        auto stash1_back = [&] {
            if constexpr (STASH1) {
                const priv_float* sp = (const priv_float*)stash1_mem;
                asm volatile("" : "+v"(sp));
#pragma unroll
                for (int i = 0; i < RS1::PER * RS1::SK; ++i) skip1[i] = sp[i];
            }
        };
        if constexpr (MCD_L8_WFIRST) {
            // ---- su4.0, then su4.1 (64 -> 32) W-first: P = [W_t ; W_r] X (64 rows), then out = PReLU(mix(P_t) + P_r + b) + e on the 32
            //      output channels, in place of P_r; up2 reads it from there (row stride 68)
            constexpr int NT = PL::P12 / 16;
            constexpr int COLS = NB * T * 12;
            const LayerW lw = layer_w(wb, 8);
            float4 afr8[4];
            MixCoef<32, 12, T, NB> mc8w;
            layer_std<7, T, NB, LOWO>(wb, mc7, RG + PL::L7_in, RG + PL::L7_z, RG + PL::L7_out, EMB, wave, lane, prof, nohook,
                                [&] {
                                    load_afrags<4, 4>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr8, 0);
                                    if constexpr (EARLY2) mc8w.load(wb + lw.tq, wb + lw.am, wave, lane);
                                }, WEARLY ? &A7 : nullptr);     // su4.0
            STAGE(13);
            lt_dump(7, RG + PL::L7_out, 68, 64, 12);
            lt_inject(8, RG + PL::L8_in, 68, 64, 12);
            float* Pb = RG + PL::L8_p;
            prof.pp.thr();
            if constexpr (!EARLY2) mc8w.load(wb + lw.tq, wb + lw.am, wave, lane);
            auto epi8 = [&](auto ti, int col, int c0, f32x4 acc, int col0, int) {
                constexpr int STEP = Tiling<4, NT>::NG * 16 * 68;
                if (col < COLS) *reinterpret_cast<float4*>(Pb + __mul24(col0, 68) + c0 + decltype(ti)::value * STEP) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            };
            gemm_tiles<4, NT, 4, 0, false, LOWO>(afr8, RG + PL::L8_in, 68, RG + PL::L8_in, 68, wave, lane, epi8, 0);
#pragma unroll
            for (int mi = 1; mi < Tiling<4, NT>::MW; ++mi) {
                load_afrags<4, 4>(reinterpret_cast<const float4*>(wb + lw.wp), wave, lane, afr8, mi);
                gemm_tiles<4, NT, 4, 0, false, LOWO>(afr8, RG + PL::L8_in, 68, RG + PL::L8_in, 68, wave, lane, epi8, mi);
            }
            if constexpr (EARLY2) { rs_early(rc4, 3); mix_early(mc9, 9); }      // up2's fragments, layer 9's mix coefficients
            stash1_back();
            bsync();
            prof.pp.lat();
            prof.mark(32 + 3 * 8 + 1);                                          // (the tool's "gemm" column of layer 8)
            const float slope8 = lw.slope;
            const float pinf8 = prelu_bound(slope8);
            mix_stage<32, 12, T, NB, LOWO, true>(Pb, 68, mc8w, wb + lw.tq, wb + lw.am, wave, lane,
                                     [&](int n, int q, int w, ChIdx c) {   // P_r of joint w, the lane's 4 channels (w >= 12: the next frame's rows, inside the region, never stored)
                                         const float* pp = (Pb + (n * (T * 12) * 68 + q * (12 * 68) + 32 + c.cb16)) + (__mul24(w, 68) + c.j);
                                         const float4 r = lds_load4(lds_addr(pp));
                                         return f32x4{r.x, r.y, r.z, r.w};
                                     },
                                     [&](int n, int q, int w, ChIdx c, f32x4 v) {
                                         const float4 b4 = lds_load4(lds_addr(BIA + 80 + c.cb16) + 4u * (unsigned)c.j);
                                         const float4 e4 = lds_load4(lds_addr(EMB + n * EMB_STRIDE + emb_off(8) + c.cb16) + 4u * (unsigned)c.j);
                                         float* pp = (Pb + (n * (T * 12) * 68 + q * (12 * 68) + 32 + c.cb16)) + (__mul24(w, 68) + c.j);
                                         const f32x2 t0 = f32x2{v[0], v[1]} + f32x2{b4.x, b4.y}, t1 = f32x2{v[2], v[3]} + f32x2{b4.z, b4.w};
                                         const f32x2 m0 = t0 * slope8, m1 = t1 * slope8;
                                         const f32x2 r0 = f32x2{__builtin_amdgcn_fmed3f(t0[0], m0[0], pinf8), __builtin_amdgcn_fmed3f(t0[1], m0[1], pinf8)} + f32x2{e4.x, e4.y};
                                         const f32x2 r1 = f32x2{__builtin_amdgcn_fmed3f(t1[0], m1[0], pinf8), __builtin_amdgcn_fmed3f(t1[1], m1[1], pinf8)} + f32x2{e4.z, e4.w};
                                         lds_store4(lds_addr(pp), r0[0], r0[1], r1[0], r1[1]);
                                     });
            if constexpr (!EARLY2) rs_early(rc4, 3);
            if constexpr (!SQ12) bsync();      // (SQ12: up2's unit (frame, block) is the one this wave's mix unit just wrote)
            prof.mark(32 + 3 * 8);                                              // (the tool's "mix" column: mix + epilogue)
            STAGE(14);
            lt_dump(8, RG + PL::L8_p + 32, 68, 32, 12);
            lt_inject(14, RG + PL::L8_p + 32, 68, 32, 12);
            if constexpr (!EARLY2) mix_early(mc9, 9);
            resample_stage<32, 12, 17, T, NB, false, true, (LOWO && MCD_RS_ILP), SQ12>(RG + PL::L8_p + 32, 68, RG + PL::UP2_out, 36, rc4, skip1, wave, lane);  // up2 (+ d1)
        } else {
        layer_std<7, T, NB, LOWO>(wb, mc7, RG + PL::L7_in, RG + PL::L7_z, RG + PL::L7_out, EMB, wave, lane, prof,
                            [&] { mix_early(mc8, 8); }, [&] { wearly(A8, MCD_LC(8)); }, WEARLY ? &A7 : nullptr);     // su4.0
        STAGE(13);
        lt_dump(7, RG + PL::L7_out, 68, 64, 12);
        lt_inject(8, RG + PL::L8_in, 68, 64, 12);
        layer_std<8, T, NB, LOWO>(wb, mc8, RG + PL::L8_in, RG + PL::L8_z, RG + PL::L8_out, EMB, wave, lane, prof,
                            [&] { rs_early(rc4, 3); },
                            [&] {
                                if constexpr (EARLY2) mix_early(mc9, 9);
                                stash1_back();
                            }, WEARLY ? &A8 : nullptr);                                            // su4.1
        STAGE(14);
        lt_dump(8, RG + PL::L8_out, 36, 32, 12);
        lt_inject(14, RG + PL::L8_out, 36, 32, 12);
        if constexpr (!EARLY2) mix_early(mc9, 9);
        resample_stage<32, 12, 17, T, NB, false, true, (LOWO && MCD_RS_ILP)>(RG + PL::L8_out, 36, RG + PL::UP2_out, 36, rc4, skip1, wave, lane);  // up2 (+ d1)
        }
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
            layer_std<9, T, NB, LOWO>(wb, mc9, RG + PL::L9_in, RG + PL::L9_z, RG + PL::L9_out, EMB, wave, lane, prof,
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
            mix_stage<16, 17, T, NB, LOWO, true>(Pb, 20, mc10, wb + lw.tq, wb + lw.am, wave, lane,
                                     ZeroInit{},
                                     [&](int n, int t, int w, ChIdx c, auto val) {
                                         if constexpr (std::is_same_v<decltype(val), f32x4>) {     // joint w, channels c.j .. c.j + 3: the two coordinates are lane group 0's
                                             if (c.j == 0) *reinterpret_cast<float2*>(ZO + ((n * T + t) * 17 + w) * C0) = make_float2(val[0], val[1]);
                                         } else {                                                  // joint 16, channel c.j
                                             if (c.j < C0) ZO[((n * T + t) * 17 + w) * C0 + c.j] = val;
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
            const int solo = NB > 1 ? WM[9] : -1;
            const bool valid = win0 + n < Bq && (solo < 0 || n == solo);
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
            const int solo = NB > 1 ? WM[9] : -1;
            if (solo < 0 || n == solo) {
                if (win0 + n < Bq && loss_out) loss_out[(size_t)(win0 + n) * Sq + s] = l;
                if (s < 64) LOSSB[n * 64 + s] = l;
                if (!(fabsf(l) <= 3.0e38f)) WM[8] = 1;       // NaN / Inf: this chain diverged
            }
        }
    }
    bsync();        // RED (the work region) and XT are rewritten by the next sample
    // A diverged chain (NaN / Inf activations: broken weights, or a NaN in the caller's noise tensor) leaves non-finite values in
    // the work region.  Pad rows meet zero coefficients all over the pass -- the joints behind a frame's last one in a mix's or a
    // resampler's padded k-step are the NEXT frame's (the next chain's) rows, stale pad columns are another layer's real ones --
    // and NaN x 0 = NaN: the diverged chain takes (i) the other chains of this pass and (ii) every later trajectory of this
    // workgroup with it, where the reference's chains are independent (mocodad.py:155-180).  So, behind a pass set whose loss is
    // not finite: (ii) the region is cleared again, and (i) with several chains per workgroup the sample is run again once per
    // chain slot, ALONE -- the other slots sit out as benign chains (x = 0, no noise, no condition) and write nothing -- so that a
    // chain that is finite on its own gets its own loss back (bit-identical: a chain's arithmetic never depends on its
    // neighbours' finite values).  The common path pays two LDS reads per sample.  Not covered: non-finite INPUT poses, whose
    // window takes its workgroup partners along through the in-kernel condition encoder (include/mocodad_hip.h: inputs are finite).
    // (Found by tests/test_samples50_gpu.py::test_a_diverged_chain_stays_alone.)
    {
        const int div = WM[8], solo = NB > 1 ? WM[9] : -1;
        if (div) {
            for (int u = tid_s; u < PL::R + PL::XT; u += NTHREADS) smem[u] = 0.f;
            for (int u = tid_s; u < PL::ZN; u += NTHREADS) { ZN[u] = 0.f; ZO[u] = 0.f; }
        }
        if (NB > 1 && (div || solo >= 0)) {
            // the joint run diverged: slot 0 alone next; a solo run: the next slot, or on to the next sample
            const int nxt = solo < 0 ? 0 : (solo + 1 < NB ? solo + 1 : -1);
            if (nxt >= 0) s -= P.split;                  // (the same sample again)
            bsync();                                     // every thread has read WM[8] / WM[9]
            if (tid_s == 0) { WM[8] = 0; WM[9] = nxt; }
            bsync();
        } else if (div) {
            bsync();
            if (tid_s == 0) WM[8] = 0;
        }
    }
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
    if constexpr (MINW >= 4) {
        if (P.mode == 0 && P.tune && blockIdx.x == 0 && te == 0) {      // this launch's measurement for the next one's time slice
            P.tune[1] = (int)((unsigned)__builtin_amdgcn_s_memrealtime() - (unsigned)UPD[14]);
            P.tune[0] = (int)gridDim.x * 1009 + P.S * 31 + P.ns;
        }
    }
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


}  // namespace mcd
