// mcd_api.hip — host side of libmocodad_hip.so: the C ABI of include/mocodad_hip.h, the weight packer (BatchNorm folding, MFMA
// fragment order), dispatch to the kernel instantiations of mcd_inst.hip (declared `extern template` in mcd_launch.hpp), and the
// kernels that are not templates:
//   cond_encode_kernel          STSE.encode for any channel list / 21 .. 31 condition frames   models/stsae/stsae.py:59-92
//   cond_unet_generic_kernel    'E_unet' condition encoder at any frame count (cross-check)    models/stsae/stsae_unet.py:62-146
//   score_generic_kernel        plain-FMA runtime-shape trajectory kernel: the CROSS-CHECK of the MFMA kernels (MCD_OPT_GENERIC_UNET)
//   aggregate_kernel            MoCoDAD._aggregation_strategy                                  models/mocodad.py:454-520
//   scatter_max / frame_scatter / frame_scores kernels   post_processing                       models/mocodad.py:362-425
// The device code shared by the trajectory kernels (stage functions, LDS plan) is mcd_device.hpp; the kernels themselves are
// mcd_score_kernel.hpp (1 .. 12 U-Net frames) and mcd_tiled_kernel.hpp (13 .. 32).  See DESIGN.md section 2.

#include "mcd_launch.hpp"

#if MCD_NWAVES != 8 && !defined(MCD_FAST_T)      // (developer builds pass one flag set to every file)
#error "mcd_api.hip is built with the default wave count: per-unit wave counts belong to mcd_inst.hip (MCD_UNIT_FLAGS_<n>)"
#endif
namespace { constexpr int API_THREADS = 512; }      // block size of this file's own kernels
#pragma GCC poison NWAVES NTHREADS                  // (translation-unit constants of the kernel units: see mcd_launch.hpp)

using namespace mcd;

namespace {

// ------------------------------------------------------------------------------------------------
// condition encoder (runtime channel list; 0.3 % of the work): one workgroup per window, VALU only.
// ------------------------------------------------------------------------------------------------

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
// aggregation over the S samples (mocodad.py:454-520); one 64-lane wave per window, ANY S (the reference's shipped
// n_generated_samples is 50, config/*/mocodad_test.yaml; its _aggregation_strategy has no cap)
// ------------------------------------------------------------------------------------------------
struct AggrParams {
    const float* loss_all; const float* pose_all; const float* data; float* loss_agg; float* pose_agg;
    int B, S, C, Tx, V, seg_len, strategy, loss_fn, in_lds;
    float q;
    int corrupt_idx[MCD_MAX_FRAMES];
};
constexpr int AGG_LDS_MAX = 8192;       // sample values staged in LDS (32 KB); a longer sample axis is read in place

// The S values of one window (its per-sample losses, or one pose element across the samples): staged in LDS, or -- beyond
// AGG_LDS_MAX samples -- read where they lie (stride = floats between consecutive samples).
struct SampleVals {
    const float* p; long long stride;
    __device__ __forceinline__ float operator()(int k) const { return p[(long long)k * stride]; }
};
__device__ __forceinline__ SampleVals stage_samples(const float* src, long long stride, int S, float* lds, bool in_lds, int lane) {
    if (!in_lds) return SampleVals{src, stride};
    __syncthreads();                                     // the previous round's readers are done with `lds`
    for (int k = lane; k < S; k += 64) lds[k] = src[(long long)k * stride];
    __syncthreads();
    return SampleVals{lds, 1};
}
// Order statistics by rank counting: sample i's rank = #{k : x_k < x_i or (x_k == x_i and k < i)} is a permutation of
// 0 .. S-1 whatever the ties; lane l ranks the samples l, l + 64, ...; the samples of rank r0 / r1 land in slot[0] / slot[1].
// (No sort, no per-thread array: O(S^2 / 64) broadcast reads per lane.)  All lanes return the same pair.
// A NaN among the samples (a diverged chain) breaks the permutation -- every NaN ranks 0 and the rank asked for may have no
// writer -- and torch.median / torch.quantile return NaN then (mocodad.py:489-492,513-516): so does this, for both values.
__device__ __forceinline__ void wave_rank_select(const SampleVals& X, int S, int lane, int r0, int r1, float* slot, float& v0, float& v1) {
    bool nan = false;
    for (int i = lane; i < S; i += 64) {
        const float x = X(i);
        nan |= x != x;
        int r = 0;
        for (int k = 0; k < S; ++k) {
            const float y = X(k);
            r += (y < x || (y == x && k < i)) ? 1 : 0;
        }
        if (r == r0) slot[0] = x;
        if (r == r1) slot[1] = x;
    }
    __syncthreads();
    const bool any_nan = __ballot(nan) != 0ull;        // (one 64-lane wave per workgroup: aggregate_kernel's launch bound)
    v0 = any_nan ? __builtin_nanf("") : slot[0];
    v1 = any_nan ? __builtin_nanf("") : slot[1];
    __syncthreads();
}
// torch.median: the lower middle value; torch.quantile: linear interpolation, torch.lerp's two-sided form
__device__ __forceinline__ float wave_order_stat(const SampleVals& X, int S, int lane, int strategy, float q, float* slot) {
    float a, c;
    if (strategy == MCD_AGGR_MEDIAN) {
        wave_rank_select(X, S, lane, (S - 1) / 2, (S - 1) / 2, slot, a, c);
        return a;
    }
    const float pos = fminf(fmaxf(q, 0.f), 1.f) * (float)(S - 1);      // (q is validated on the host; the clamp is a backstop)
    const int lo = (int)floorf(pos);
    const int hi = lo + 1 < S ? lo + 1 : S - 1;
    const float wgt = pos - (float)lo;
    wave_rank_select(X, S, lane, lo, hi, slot, a, c);
    return wgt < 0.5f ? a + wgt * (c - a) : c - (c - a) * (1.f - wgt);
}

// Sums and best / worst run in sample order on wave-uniform values: bit-identical to the sequential loops of
// aggregate_losses in the fused kernel.
__global__ __launch_bounds__(64) void aggregate_kernel(const AggrParams P) {
    extern __shared__ float agg_lds[];
    float* slot = agg_lds;               // [2] selected order statistics
    float* vals = agg_lds + 2;           // [S] staged sample values (in_lds)
    const int lane = threadIdx.x, S = P.S, per = P.C * P.Tx * P.V;
    const bool in_lds = P.in_lds != 0;
    for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
        if (P.strategy <= MCD_AGGR_QUANTILE && P.strategy != MCD_AGGR_MEAN_POSE && P.strategy != MCD_AGGR_MEDIAN_POSE) {
            const SampleVals X = stage_samples(P.loss_all + (size_t)b * S, 1, S, vals, in_lds, lane);
            if (P.strategy == MCD_AGGR_BEST || P.strategy == MCD_AGGR_WORST) {
                const bool best = P.strategy == MCD_AGGR_BEST;
                float cur = best ? 1e10f : -1.f;       // mocodad.py:504-512: strict comparisons from 1e10 / -1 (the FIRST of equal samples stays)
                int sel = -1;
                for (int k = 0; k < S; ++k) {
                    const float y = X(k);
                    if (best ? (y < cur) : (y > cur)) { cur = y; sel = k; }
                }
                if (lane == 0) P.loss_agg[b] = cur;
                if (P.pose_agg)
                    for (int e = lane; e < per; e += 64)
                        P.pose_agg[(size_t)b * per + e] = sel >= 0 ? P.pose_all[((size_t)b * S + sel) * per + e] : 0.f;
            } else if (P.strategy == MCD_AGGR_MEAN) {
                float sum = 0.f;
                for (int k = 0; k < S; ++k) sum += X(k);
                if (lane == 0) P.loss_agg[b] = sum / (float)S;
            } else {
                const float r = wave_order_stat(X, S, lane, P.strategy, P.q, slot);
                if (lane == 0) P.loss_agg[b] = r;
            }
        } else {  // mean_pose / median_pose: per element over the S generated poses, then the loss of that pose (mocodad.py:493-503)
            float acc = 0.f;      // (every lane carries the same running sum: the per-element values are wave-uniform)
            for (int e = 0; e < per; ++e) {
                const SampleVals X = stage_samples(P.pose_all + (size_t)b * S * per + e, per, S, vals, in_lds, lane);
                float val;
                if (P.strategy == MCD_AGGR_MEAN_POSE) {
                    float sum = 0.f;
                    for (int k = 0; k < S; ++k) sum += X(k);
                    val = sum / (float)S;
                } else {
                    val = wave_order_stat(X, S, lane, MCD_AGGR_MEDIAN, 0.f, slot);
                }
                if (P.pose_agg && lane == 0) P.pose_agg[(size_t)b * per + e] = val;
                const int c = e / (P.Tx * P.V), tx = (e / P.V) % P.Tx, v = e % P.V;
                const float gt = P.data[(((size_t)b * P.C + c) * P.seg_len + P.corrupt_idx[tx]) * P.V + v];
                acc += loss_elem(val, gt, P.loss_fn);
            }
            if (lane == 0) P.loss_agg[b] = acc / (float)per;
        }
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

// 'E_unet' condition encoder at any frame count (the U-Net's down path without embeddings + to_time_dim), same scratch scheme
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
    // (+ MIX_QPAD zero rows: the ragged frame groups of 5 / 7 / 11 frames compute up to one output frame beyond the last)
    tqf = B.alloc((size_t)(TP + MIX_QPAD) * NR * 64);
    af = B.alloc((size_t)(TP + MIX_QPAD) * MT * KS * 64);
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

namespace {

int launch_score(const mcd_weights* w, int T, ScoreParams& P, hipStream_t st, bool* fused = nullptr) {
    P.force_split = w->opt[MCD_OPT_SPLIT];
    P.phase = w->opt[MCD_OPT_PHASE];
#if defined(MCD_FAST_T)     // developer builds: one instantiation only (mcd_launch.hpp)
    if (T != MCD_FAST_T) return fail(MCD_EUNSUPPORTED, "fast build");
    return launch_score_t<MCD_FAST_T, MCD_FAST_NB, MCD_FAST_MINW>(P, st, fused);
#else
#ifdef MCD_TUNING_VARIANTS  // alternative workgroup shapes (MCD_OPT_VARIANT), developer builds only
    const int variant = w->opt[MCD_OPT_VARIANT];
    if (T == 3 && variant == 1) return launch_score_t<3, 4, 2>(P, st, fused);   // 4 chains / WG, 1 WG per CU
    if (T == 3 && variant == 3) return launch_score_t<3, 1, 4>(P, st, fused);   // 1 chain / WG (tuning variant)
    if (T == 3 && variant == 2) return launch_score_t<3, 2, 2>(P, st, fused);   // the default shape without the register cap
    if (T == 6 && variant == 1) return launch_score_t<6, 2, 2>(P, st, fused);   // 2 chains / WG, 1 WG per CU (no register cap)
#else
    if (w->opt[MCD_OPT_VARIANT] != 0) return fail(MCD_EUNSUPPORTED, "MCD_OPT_VARIANT needs a -DMCD_TUNING_VARIANTS build");
#endif
    switch (T) {
        case 3: return launch_score_t<3, 2, 4>(P, st, fused);                 // 2 chains / WG, 2 WGs per CU (<= 128 VGPRs)
        case 6: return launch_score_t<6, 1, 4>(P, st, fused);                 // 1 chain / WG, 2 WGs per CU
        case 12: return launch_score_t<12, 1, 3>(P, st, fused);               // 1 chain / WG of 12 waves, 1 WG per CU (168 registers; mcd_instances.hpp)
        case 4: return launch_score_t<4, 1, 4>(P, st, fused);                 // e.g. seg_len 8 split in halves
        case 5: return launch_score_t<5, 1, 4>(P, st, fused);                 // e.g. seg_len 10 split in halves (1 chain / WG, 2 WGs per CU: +4.7 % over <5,2,2>, profiles/r04r_t5_shape_ab.txt)
        case 8: return launch_score_t<8, 1, 2>(P, st, fused);                 // e.g. seg_len 8 concat / seg_len 12 with 4 condition frames
        case 10: return launch_score_t<10, 1, 3>(P, st, fused);               // e.g. seg_len 20 split in halves / seg_len 10 concat
        case 7: return launch_score_t<7, 1, 2>(P, st, fused);                 // odd frame counts: one output frame per mix unit
        case 9: return launch_score_t<9, 1, 3>(P, st, fused);                 // (12 waves, like 12 frames)
        case 11: return launch_score_t<11, 1, 3>(P, st, fused);
        case 1: return launch_score_t<1, 4, 4>(P, st, fused);                 // (4 chains / WG, 2 WGs per CU)
        case 2: return launch_score_t<2, 2, 4>(P, st, fused);                 // e.g. seg_len 4 split in halves (2 chains / WG, 2 WGs per CU: every mix is one round of units; +31 % over <2,3,4>, profiles/r04aa_t2_nb_ab.txt)
        default: return fail(MCD_EUNSUPPORTED, "U-Net frame count " + std::to_string(T) + " not instantiated (supported: 1 .. 12)");
    }
#endif
}

}  // namespace

namespace {
// T_c -> NB of the MFMA condition encoders (the chains-per-workgroup of the trajectory kernels' LDS plans)
#ifdef MCD_FAST_T
#define MCD_COND_CASE(fn, unit, T, NB) case T: if (T == MCD_FAST_T && NB == MCD_FAST_NB) return fn<MCD_FAST_T, MCD_FAST_NB>(w, data, fi, seg_len, emb, B, st); break;
#else
#define MCD_COND_CASE(fn, unit, T, NB) case T: return fn<T, NB>(w, data, fi, seg_len, emb, B, st);
#endif
int launch_cond_fast(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, hipStream_t st) {
    switch (w->cond.Tc) {
#define MCD_CASE(unit, T, NB) MCD_COND_CASE(launch_cond_fast_t, unit, T, NB)
        MCD_COND_FAST_INSTANCES(MCD_CASE)
#undef MCD_CASE
        default: break;
    }
    return fail(MCD_EUNSUPPORTED, "cond_fast: frame count not instantiated");
}
// frame counts the MFMA 'E_unet' encoder is instantiated for (the trajectory kernel's LDS plans)
bool cond_unet_has_kernel(int Tc) {
#ifdef MCD_FAST_T
    return Tc == MCD_FAST_T;
#else
    return Tc >= 1 && Tc <= 12;
#endif
}
int launch_cond_unet(const mcd_weights* w, const DataView& data, const FrameIdx& fi, int seg_len, float* emb, int B, hipStream_t st) {
    switch (w->cond.Tc) {
#define MCD_CASE(unit, T, NB) MCD_COND_CASE(launch_cond_unet_t, unit, T, NB)
        MCD_COND_UNET_INSTANCES(MCD_CASE)
#undef MCD_CASE
        default: break;
    }
    return fail(MCD_EUNSUPPORTED, "E_unet condition encoder: frame count not instantiated");
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
int tiled_wgs(const mcd_weights* w, int64_t chains, int TP) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, w->device);     // the handle's device, whatever the caller's current one
    if (cus < 1) cus = 256;
    const int64_t units = (chains + tl_nb(TP) - 1) / tl_nb(TP);
    cus *= tl_wgs_per_cu(TP);
    return (int)(units < cus ? units : cus);         // one workgroup per CU (110 - 135 KB of LDS), persistent over the chains
}
int64_t tiled_scratch_bytes(const mcd_weights* w, int64_t chains, int TP) { return (int64_t)tiled_wgs(w, chains, TP) * tl_slab_floats(TP * tl_nb(TP)) * 4; }
int launch_score_tiled(const mcd_weights* w, const ScoreParams& P, const FrameMaps& M, float* scratch, hipStream_t st, bool layer_test = false) {
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != w->device) return fail(MCD_EINVAL, "the current device is not the handle's device (the workspace slabs are sized for it)");
    const int wgs = tiled_wgs(w, P.n_chains, w->tiled_tp);
#ifdef MCD_FAST_T
#ifdef MCD_FAST_TILED
    if (w->tiled_tp == MCD_FAST_TILED) return launch_score_tiled_t<MCD_FAST_TILED, tl_nb(MCD_FAST_TILED)>(w, P, M, scratch, wgs, st);
#endif
    (void)wgs; (void)layer_test;
    return fail(MCD_EUNSUPPORTED, "fast build");
#else
    switch (layer_test ? -w->tiled_tp : w->tiled_tp) {
#define MCD_CASE(unit, TP, NB, LT) case (LT ? -TP : TP): return launch_score_tiled_t<TP, NB, LT>(w, P, M, scratch, wgs, st);
        MCD_TILED_INSTANCES(MCD_CASE)
#undef MCD_CASE
        default: return fail(MCD_EUNSUPPORTED, "tiled kernel: frame count");
    }
#endif
}
// plain condition encoder (any channel list; 21 .. 31 condition frames of the shipped one).  scratch: cond_plain_scratch_bytes()
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
    if (w->tiled_cond_tp && !w->opt[MCD_OPT_COND_GENERIC]) {      // 13 .. 32 condition frames: the slab-tiled MFMA stages, one window per "chain"
        ScoreParams P;
        memset(&P, 0, sizeof(P));
        P.wbuf = w->dbuf; P.dv = data; P.seg_len = seg_len; P.B = B; P.S = 1; P.n_chains = B; P.ns = 2; P.eps_out = emb;
        FrameMaps M;
        memset(&M, 0, sizeof(M));
        for (int t = 0; t < Tc; ++t) M.src_frame[t] = fi.idx[t];
        const int wgs = tiled_wgs(w, B, w->tiled_cond_tp);
        switch (w->tiled_cond_tp) {
#ifdef MCD_FAST_T
#ifdef MCD_FAST_TILED_COND
            case MCD_FAST_TILED_COND: return launch_score_tiled_t<MCD_FAST_TILED_COND, tl_nb(MCD_FAST_TILED_COND), false, true>(w, P, M, scratch, wgs, st);
#endif
#else
#define MCD_CASE(unit, TP, NB) case TP: return launch_score_tiled_t<TP, NB, false, true>(w, P, M, scratch, wgs, st);
            MCD_TILED_COND_INSTANCES(MCD_CASE)
#undef MCD_CASE
#endif
            default: break;      // (developer builds without this instantiation: the runtime-shape kernel below)
        }
    }
    const int wgs = B < GEN_MAX_WGS ? B : GEN_MAX_WGS;
    hipLaunchKernelGGL(cond_unet_generic_kernel, dim3(wgs), dim3(GEN_THREADS), 0, st, w->dbuf, w->gcond, data, fi, seg_len, Tc, B, emb, scratch);
    HIP_TRY(hipGetLastError());
    return MCD_OK;
}
}  // namespace

static unsigned long long* g_prof = nullptr;  // MCD_PROFILE builds: device buffer of 32 accumulators

// test aid (mcd_debug_poison_lds): every CU's LDS filled with signalling garbage (NaN bit patterns), so that a kernel reading
// shared memory it never wrote produces NaNs instead of depending on what the previous kernel happened to leave there
__global__ __launch_bounds__(API_THREADS) void poison_lds_kernel(unsigned* sink, int words) {
    extern __shared__ unsigned psm[];
    for (int u = threadIdx.x; u < words; u += API_THREADS) psm[u] = 0x7fc00000u | (unsigned)u;
    __syncthreads();
    if (threadIdx.x == 0 && sink) atomicOr(sink, psm[(blockIdx.x * 7919) % words] & 1u);      // (keeps the stores alive)
    __builtin_amdgcn_s_sleep(64);
}

// the library is built with -fvisibility=hidden: only the C ABI of include/mocodad_hip.h is exported
#pragma GCC visibility push(default)
extern "C" {

void mcd_debug_set_prof(void* p) { g_prof = reinterpret_cast<unsigned long long*>(p); }

int mcd_debug_poison_lds(void* stream) {
    constexpr size_t lds = 160 * 1024;
    LDS_LIMIT(poison_lds_kernel, lds);
    // one 160 KB workgroup per CU at a time; several waves of them so that every CU of every XCD takes at least one
    hipLaunchKernelGGL(poison_lds_kernel, dim3(4096), dim3(API_THREADS), lds, static_cast<hipStream_t>(stream), (unsigned*)nullptr, (int)(lds / 4));
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
    struct HostLayer { int tq, am, wp, bias; float slope; };
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
        //   W-first layers 6, 8 and 10: M = [W_t' ; W_r'] stacked (layer 10: rows 0,1 / 2,3 of one 16-row tile), K = cinp
        const bool wfirst = (l == 6 || l == 10 || (l == 8 && MCD_L8_WFIRST));
        const int cinp = D.cin;
        const int M = (l == 6 || (l == 8 && MCD_L8_WFIRST)) ? 2 * D.cout : mpad;
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
            if (l == 6 || (l == 8 && MCD_L8_WFIRST)) {    // this kernel runs layers 6 and 8 mix-first like the others: [W_t' | W_r'] fragments (the specialised kernels' are W-first)
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
    TiledNet TNc;                   // ... and its tables for score_tiled_kernel<.., COND> (13 .. 32 condition frames)
    memset(&TNc, 0, sizeof(TNc));
    int tiled_cond_tp = 0;
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
        if (Tc > 12) {      // the slab-tiled MFMA stages: mix tables for the padded frame count; GEMM fragments, biases, slopes as above
            tiled_cond_tp = Tc <= 16 ? 16 : Tc <= 24 ? 24 : 32;
            for (int l = 0; l < 7; ++l) {
                const std::string p = std::string("condition_encoder.") + unames[l];
                if (!pack_mix_mfma(tm, p, Tc, uv[l], B, TNc.tq[l], TNc.am[l], tiled_cond_tp)) return fail(MCD_EMISSING, tm.missing);
                const float* Tm = tm.get(p + ".gcn.T", (int64_t)uv[l] * Tc * Tc);
                if (!Tm) return fail(MCD_EMISSING, tm.missing);
                TNc.tqm[l] = pack_time_mfma(Tm, Tc, uv[l], tiled_cond_tp, tl_nb(tiled_cond_tp), B);
                TNc.wp[l] = utab[l * F_STRIDE + F_WP]; TNc.bias[l] = utab[l * F_STRIDE + F_BIAS];
                memcpy(&TNc.slope[l], &utab[l * F_STRIDE + F_SLOPE], sizeof(float));
            }
            for (int r = 0; r < 2; ++r) {
                Folded f;
                const std::string p = std::string("condition_encoder.") + urs[r];
                if (!fold_conv_bn(tm, p + ".block.0", p + ".block.1", urout[r], urin[r], f)) return fail(MCD_EMISSING, tm.missing);
                const int vin = urin[r], vout = urout[r], KS = (vin + 3) / 4, MTr = (vout + 15) / 16;
                const int wf = B.alloc((size_t)MTr * KS * 64 + 32);
                for (int mt = 0; mt < MTr; ++mt) for (int ks = 0; ks < KS; ++ks) for (int lane = 0; lane < 64; ++lane) {
                    const int vo = mt * 16 + (lane & 15), v = rs_vmap(false, vin, ks, lane >> 4);
                    B.buf[wf + (mt * KS + ks) * 64 + lane] = (vo < vout && v < vin) ? (float)f.w[(size_t)vo * vin + v] : 0.f;
                }
                for (int vo = 0; vo < vout; ++vo) B.buf[wf + MTr * KS * 64 + vo] = (float)f.b[vo];
                TNc.rsw[r] = wf;
            }
            TNc.we = utab[TABC_ULW]; TNc.be = utab[TABC_ULB];
        }
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
                    Cw.Tc >= 1 && Cw.Tc <= MCD_COND_FAST_MAX_T;
#ifdef MCD_FAST_T
        cond_fast = cond_fast && Cw.Tc == MCD_FAST_T;     // (developer builds hold one frame count; the rest takes the plain encoder)
#endif
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
    w->zero_row = zero_row; w->fast_unet = fast_unet; w->gen = G; w->gcond = GC; w->tiled = TN; w->tiled_tp = tiled_tp; w->tiled_cond = TNc; w->tiled_cond_tp = tiled_cond_tp;
    w->cfg = *cfg; w->device = device; w->n_floats = B.buf.size(); w->has_cond = has_cond; w->cond_fast = cond_fast; w->cond_unet = cond_unet;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&w->dbuf), B.buf.size() * sizeof(float));
    if (e != hipSuccess) { delete w; return fail(MCD_EDEVICE, std::string("hipMalloc: ") + hipGetErrorString(e)); }
    e = hipMemcpy(w->dbuf, B.buf.data(), B.buf.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(w->dbuf); delete w; return fail(MCD_EDEVICE, std::string("hipMemcpy: ") + hipGetErrorString(e)); }
    w->tune = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&w->tune), 4 * sizeof(int)) != hipSuccess || hipMemset(w->tune, 0, 4 * sizeof(int)) != hipSuccess) {
        if (w->tune) (void)hipFree(w->tune);
        (void)hipFree(w->dbuf); delete w;
        return fail(MCD_EDEVICE, "hipMalloc (tuning words)");
    }
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
    if (w->tune) (void)hipFree(w->tune);
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

// scratch of the single-pass entries: the slabs of the slab-tiled kernel (13 .. 32 U-Net frames) or of the runtime-shape kernel
int64_t mcd_pass_workspace_bytes(const mcd_weights_t* w, int32_t n_windows) {
    if (!w || n_windows <= 0) return 0;
    int64_t b = 0;
    if (!w->fast_unet || w->opt[MCD_OPT_GENERIC_UNET]) b = gen_scratch_bytes(n_windows, w->cfg.t_unet);
    if (!w->fast_unet && w->tiled_tp) { const int64_t t = tiled_scratch_bytes(w, n_windows, w->tiled_tp); if (t > b) b = t; }
    return b;
}

int mcd_unet_forward(const mcd_weights_t* w, const float* x, const float* cond, const float* step_table, int32_t t,
                     int32_t n_windows, float* eps_out, void* workspace, void* stream) {
    if (!w) return fail(MCD_EINVAL, "null argument");
    if (n_windows <= 0) return MCD_OK;
    if (!x || !step_table || !eps_out) return fail(MCD_EINVAL, "null argument");
    if (t < 0) return fail(MCD_EINVAL, "t must be >= 0 (step_table needs at least t + 1 rows)");
    ScoreParams P;
    memset(&P, 0, sizeof(P));
    P.wbuf = w->dbuf; P.x_in = x; P.cond_emb = cond; P.step_table = step_table; P.eps_out = eps_out;
    P.B = n_windows; P.S = 1; P.ns = t + 1; P.seg_len = w->cfg.t_unet; P.n_corrupt = w->cfg.t_unet; P.fixed_mask = 0;
    P.mode = 1; P.step_single = t; P.n_chains = n_windows;
    hipStream_t st = (hipStream_t)stream;
    if (w->fast_unet && !w->opt[MCD_OPT_GENERIC_UNET]) return launch_score(w, w->cfg.t_unet, P, st);
    // 13 .. 32 frames: the slab-tiled kernel in single-pass mode; MCD_OPT_GENERIC_UNET: the runtime-shape kernel
    if (!workspace) return fail(MCD_EINVAL, "workspace required (mcd_pass_workspace_bytes)");
    FrameMaps M;
    memset(&M, 0, sizeof(M));
    if (w->tiled_tp && !w->opt[MCD_OPT_GENERIC_UNET]) return launch_score_tiled(w, P, M, reinterpret_cast<float*>(workspace), st);
    return launch_score_generic(w, P, M, reinterpret_cast<float*>(workspace), st);
}

int mcd_layer_forward(const mcd_weights_t* w, int32_t stage, const float* x, const float* skip, const float* emb, int32_t n_windows,
                      float* out, void* workspace, void* stream) {
    if (!w) return fail(MCD_EINVAL, "null argument");
    if (stage < 0 || stage > 14) return fail(MCD_EINVAL, "stage must be 0..10 (ST-GCN layers) or 11..14 (down1, down2, up3, up2)");
    if (n_windows <= 0) return MCD_OK;
    if (!x || !out || !emb) return fail(MCD_EINVAL, "null argument");
    ScoreParams P;
    memset(&P, 0, sizeof(P));
    P.wbuf = w->dbuf; P.cond_emb = emb; P.step_table = w->dbuf + w->zero_row;   // pe = 0: the layers see SiLU(emb)
    P.B = n_windows; P.S = 1; P.ns = 1; P.seg_len = w->cfg.t_unet; P.n_corrupt = w->cfg.t_unet;
    P.mode = 1; P.step_single = 0; P.n_chains = n_windows;
    P.lt_stage = stage; P.lt_in = x; P.lt_out = out; P.lt_skip = skip;
    P.x_in = stage == 0 ? x : nullptr;     // layer 0 reads the chain state itself; the other stages start from x = 0
#ifdef MCD_FAST_T
    return fail(MCD_EUNSUPPORTED, "fast build");
#else
    hipStream_t st = (hipStream_t)stream;
    if (w->tiled_tp) {      // 13 .. 32 frames: the joint resamplers are fused into layers 3, 5, 7, 9 (no stages of their own)
        if (stage > 10) return fail(MCD_EUNSUPPORTED, "13 .. 32 U-Net frames: the joint resamplers are part of stages 3, 5, 7, 9");
        if (skip && stage != 7 && stage != 9) return fail(MCD_EINVAL, "skip tensor: stages 7 (d2) and 9 (d1) only");
        if (!workspace) return fail(MCD_EINVAL, "workspace required (mcd_pass_workspace_bytes)");
        FrameMaps M;
        memset(&M, 0, sizeof(M));
        return launch_score_tiled(w, P, M, reinterpret_cast<float*>(workspace), st, true);
    }
    if (skip) return fail(MCD_EINVAL, "skip tensor: only the fused stages of 13 .. 32 U-Net frames take one");
    switch (w->cfg.t_unet) {
        case 3: return launch_score_t<3, 2, 4, true>(P, st, nullptr);
        case 6: return launch_score_t<6, 1, 4, true>(P, st, nullptr);
        case 12: return launch_score_t<12, 1, 3, true>(P, st, nullptr);      // (the template arguments and unit flags of the production kernels)
        case 5: return launch_score_t<5, 1, 4, true>(P, st, nullptr);
        case 7: return launch_score_t<7, 1, 2, true>(P, st, nullptr);
        case 9: return launch_score_t<9, 1, 3, true>(P, st, nullptr);
        case 10: return launch_score_t<10, 1, 3, true>(P, st, nullptr);
        case 11: return launch_score_t<11, 1, 3, true>(P, st, nullptr);
        default: return fail(MCD_EUNSUPPORTED, "mcd_layer_forward is instantiated for 3, 5, 6, 7, 9, 10, 11, 12 and 13 .. 32 U-Net frames");
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
        const int64_t g3 = tiled_scratch_bytes(w, (int64_t)cfg->n_windows * cfg->n_samples, w->tiled_tp);
        if (g3 > gen) gen = g3;
    }
    if (w->cond_unet) { const int64_t g2 = gen_scratch_bytes(cfg->n_windows, w->cond.Tc); if (g2 > gen) gen = g2; }
    if (w->cond_unet && w->tiled_cond_tp) { const int64_t g5 = tiled_scratch_bytes(w, cfg->n_windows, w->tiled_cond_tp); if (g5 > gen) gen = g5; }
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

static int launch_aggregate(AggrParams& A, hipStream_t st) {
    A.in_lds = A.S <= AGG_LDS_MAX;
    const size_t lds = (size_t)(2 + (A.in_lds ? A.S : 0)) * sizeof(float);
    hipLaunchKernelGGL(aggregate_kernel, dim3(A.B < 65536 ? A.B : 65536), dim3(64), lds, st, A);      // one wave per window
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
        if (aggr == MCD_AGGR_QUANTILE && !(quantile >= 0.f && quantile <= 1.f))       // (also rejects NaN; torch.quantile raises)
            return fail(MCD_EINVAL, "quantile must be in [0, 1]");
    }
    if (S < 1 || cfg->noise_steps < 2) return fail(MCD_EINVAL, "need n_samples >= 1 and noise_steps >= 2");
    if ((long long)B * S > 0x7fffffffll) return fail(MCD_EINVAL, "n_windows x n_samples exceeds 2^31 - 1: score in smaller batches");
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
    P.tune = w->opt[MCD_OPT_PHASE] == -2 ? nullptr : w->tune;      // (phase -2: the host's estimate only -- A/B of the self-calibration)
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
    // data frames the kernels read (load_coord) must lie inside the window
    for (int k = 0; k < cfg->n_cond; ++k)
        if (cfg->cond_idx[k] < 0 || cfg->cond_idx[k] >= cfg->seg_len) return fail(MCD_EINVAL, "cond_idx outside [0, seg_len)");
    for (int k = 0; k < cfg->n_corrupt; ++k)
        if (cfg->corrupt_idx[k] < 0 || cfg->corrupt_idx[k] >= cfg->seg_len) return fail(MCD_EINVAL, "corrupt_idx outside [0, seg_len)");
    for (int k = 0; k < tf && !rnd; ++k) {
        const int t = strat == MCD_STRATEGY_INBETWEEN_IMP ? cfg->cond_idx[k] : k;
        if (t < 0 || t >= Tu || ((P.fixed_mask >> t) & 1)) return fail(MCD_EINVAL, "bad cond_idx");
        P.fixed_mask |= 1u << t;
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
    if (cfg->n_samples < 1) return fail(MCD_EINVAL, "need n_samples >= 1");
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
#pragma GCC visibility pop
