/*
 * mocodad_hip.h — C ABI of the MI355X-native MoCoDAD anomaly-scoring path (libmocodad_hip.so).
 *
 * The reference (aleflabo/MoCoDAD) is pure Python/PyTorch and has no FFI; the seam it offers is the
 * Python module surface of models/mocodad.py.  These entry points are what a maintainer binds with
 * ctypes from that surface; each cites the reference code it replaces.
 *
 * Conventions
 *   - every function returns 0 on success or a negative MCD_E* code; mcd_last_error() returns a
 *     thread-local message.  Nothing throws across the ABI.
 *   - the CALLER owns every buffer (device pointers unless stated "host"); the library owns only
 *     mcd_weights_t.  All tensors are dense row-major float32 in the reference's own layouts.
 *   - compute entry points are asynchronous on the caller's hipStream_t (passed as void*), perform no
 *     allocation and no host synchronisation, and are re-entrant across streams and devices.  (One exception, a diagnostic
 *     entry: mcd_debug_set_prof sets one process-wide pointer.)
 *   - there is NO CPU fallback: without a gfx950 device these calls fail with MCD_EDEVICE.
 */
#ifndef MOCODAD_HIP_H
#define MOCODAD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCD_ABI_VERSION 4

enum {
    MCD_OK = 0,
    MCD_EINVAL = -1,      /* bad argument / unsupported shape */
    MCD_EMISSING = -2,    /* a state_dict tensor is missing or has the wrong size */
    MCD_EDEVICE = -3,     /* HIP error (no device, launch failure, ...) */
    MCD_EUNSUPPORTED = -4 /* configuration the kernels do not cover (message says which) */
};

/* conditioning strategy of models/mocodad.py:24-29,100-126 (canonical names) */
enum { MCD_STRATEGY_INJECT = 0, MCD_STRATEGY_CONCAT = 1, MCD_STRATEGY_NO_CONDITION = 2,
       MCD_STRATEGY_INBETWEEN_IMP = 3, /* condition frames stay at their own positions among the U-Net frames */
       MCD_STRATEGY_RANDOM_IMP = 4     /* same, with a different random set of n_cond condition frames per window
                                          (mcd_window_view_t.cond_mask) */ };
/* loss_fn of models/mocodad.py:24,66 (reduction='none', mean over C*Tx*V at :484) */
enum { MCD_LOSS_SMOOTH_L1 = 0, MCD_LOSS_L1 = 1, MCD_LOSS_MSE = 2 };
/* aggregation strategy of models/mocodad.py:454-520 */
enum {
    MCD_AGGR_ALL = 0, MCD_AGGR_BEST = 1, MCD_AGGR_WORST = 2, MCD_AGGR_MEAN = 3, MCD_AGGR_MEDIAN = 4,
    MCD_AGGR_MEAN_POSE = 5, MCD_AGGR_MEDIAN_POSE = 6, MCD_AGGR_QUANTILE = 7
};

#define MCD_MAX_FRAMES 32
#define MCD_MAX_COND_LAYERS 8
/* cond_layers value selecting the 'E_unet' condition encoder (STSE_Unet: the U-Net's down path + to_time_dim,
 * models/mocodad.py:110-114, stsae_unet.py:8-251); cond_channels is ignored */
#define MCD_COND_UNET (-1)

/* One named fp32 tensor of the Lightning checkpoint's state_dict (HOST memory).  Names are the
 * reference's own keys: "model.st_gcnnsd1.0.tcn.0.weight", "condition_encoder.btlnk.bias", ... */
typedef struct {
    const char* name;
    const float* data;
    int64_t numel;
} mcd_tensor_t;

/* Architecture, as MoCoDAD.build_model derives it (models/mocodad.py:90-126, stsae_unet.py:254-357). */
typedef struct {
    int32_t num_coords;   /* C: 2 */
    int32_t n_joints;     /* V: 17 (the U-Net hard-wires 17/12/10, stsae_unet.py:11) */
    int32_t t_unet;       /* frames the U-Net runs on: n_frames_corrupt (inject) or seg_len (the other strategies) */
    int32_t t_cond;       /* condition frames seen by the condition encoder (0 when there is none) */
    int32_t emb_dim;      /* embedding_dim == latent_dim: 16 */
    int32_t strategy;     /* MCD_STRATEGY_* */
    int32_t cond_layers;  /* ST-GCN layers of the condition encoder: len(channels)+1, or MCD_COND_UNET */
    int32_t cond_channels[MCD_MAX_COND_LAYERS]; /* their output channels: channels + [h_dim] */
} mcd_model_cfg_t;

/* One scoring call = MoCoDAD.forward on one batch (models/mocodad.py:129-184). */
typedef struct {
    int32_t n_windows;    /* B */
    int32_t n_samples;    /* n_generated_samples S */
    int32_t noise_steps;  /* ns: ns-1 denoiser passes per sample (mocodad.py:163) */
    int32_t seg_len;      /* T of the data tensor (B,C,T,V) */
    int32_t n_cond;       /* len(cond_idx)   (0 for no_condition) */
    int32_t n_corrupt;    /* len(corrupt_idx) */
    int32_t cond_idx[MCD_MAX_FRAMES];     /* frame indices selected at mocodad.py:743-748 */
    int32_t corrupt_idx[MCD_MAX_FRAMES];
    int32_t loss_fn;      /* MCD_LOSS_* */
} mcd_score_cfg_t;

typedef struct mcd_weights mcd_weights_t;

/* Replaces: LightningModule.load_state_dict + model.eval() (eval_MoCoDAD.py:36-38).
 * Folds every eval-mode BatchNorm2d into the preceding 1x1 conv (stsgcn.py:57-80,181-182), repacks the
 * channel-mixing matrices into MFMA fragment order and uploads them to `device`.
 * The handle is read-only for the scoring calls (any number of streams / host threads may share it) with ONE exception: two
 * device words in which workgroup 0 of a trajectory launch leaves (launch signature, its own lifetime in 100 MHz ticks) for
 * the next launch of the same grid to size its wave-priority time slice from.  Launches that overlap on different streams
 * race on these words (plain stores, no ordering); a torn or stale pair only selects the host's estimate of the slice (the
 * signature does not match) or a slice measured by the other launch -- scheduling, never results: scores are bit-identical
 * whatever the words hold (tests/test_hip_parity.py::test_overlapping_launches_on_two_streams). */
int mcd_pack_weights(const mcd_tensor_t* tensors, int32_t n_tensors, const mcd_model_cfg_t* cfg,
                     int32_t device, mcd_weights_t** out);
void mcd_free_weights(mcd_weights_t* w);

/* Replaces: MoCoDAD._encode_condition -> STSAE/STSE.encode (mocodad.py:546-560, stsae.py:59-92).
 * cond_data (B,C,t_cond,V) -> emb_out (B,emb_dim).  The AE decoder (dead work at eval) is not run.
 * (A test / diagnostic entry without a workspace argument: encoders that need scratch memory -- 'E_unet' above 12 condition
 * frames, the shipped encoder at 25 .. 31 (three 32-channel activation buffers no longer fit LDS) -- return MCD_EUNSUPPORTED here and run inside mcd_score / mcd_score_fused.) */
int mcd_cond_encode(const mcd_weights_t* w, const float* cond_data, int32_t n_windows, float* emb_out,
                    void* stream);

/* Scratch of the two single-pass entries below for n_windows windows: 0 for 1 .. 12 U-Net frames (everything lives in LDS), the
 * activation slabs of the slab-tiled kernel for 13 .. 32 frames (or of the runtime-shape kernel under MCD_OPT_GENERIC_UNET). */
int64_t mcd_pass_workspace_bytes(const mcd_weights_t* w, int32_t n_windows);

/* Replaces: STSAE_Unet.forward (stsae_unet.py:406-438) for one timestep shared by the batch.
 * x (B,C,t_unet,V), step_table row `t` (see mcd_score; t >= 0, the table must hold at least t + 1 rows -- a negative t is
 * MCD_EINVAL), cond (B,emb_dim) or NULL -> eps_out (B,C,t_unet,V).
 * Runs the production kernel of the frame count in single-pass mode: score_kernel<T_u,...> (1 .. 12 frames) or
 * score_tiled_kernel (13 .. 32; workspace = mcd_pass_workspace_bytes, else may be NULL). */
int mcd_unet_forward(const mcd_weights_t* w, const float* x, const float* cond, const float* step_table,
                     int32_t t, int32_t n_windows, float* eps_out, void* workspace, void* stream);

/* TEST ENTRY.  One stage of the U-Net alone, run by the production stage functions inside the trajectory kernel:
 * stage 0..10 = ST_GCNN_layer.forward of the 11 layers in execution order (stsgcn.py:94-116; st_gcnnsp1a.0, sd1.0, sd1.1,
 * sd2.0, sd2.1, sd3.0, sd3.1, su4.0, su4.1, su3.0, su3.1), 11..14 = CNN_layer over the joint axis as called at
 * stsae_unet.py:205,213,381,391 (down1, down2, up3, up2; without the skip add).
 * x (B,Cin,t_unet,Vin), emb (B,emb_dim) = the layer's `t` argument (the layer adds Linear(SiLU(emb)); required),
 * out (B,Cout,t_unet,Vout).  Instantiated for 3, 5, 6, 7, 9, 10, 11, 12 U-Net frames (score_kernel's LDS plan; each with the
 * template arguments and compile flags of the production kernel of that frame count -- twelve waves at 9 .. 12) and for 13 .. 32
 * (score_tiled_kernel: the stage's input is put where the previous layer's epilogue leaves it -- slab and LDS hand-over
 * regions -- and its output read from where its own epilogue puts it; workspace = mcd_pass_workspace_bytes).  In the slab-tiled
 * kernel the joint resamplers are not stages of their own: stages 3, 5, 7, 9 are (down1 | down2 | up3 | up2) + the layer, x is
 * the RESAMPLER's input (B,Cin,t_unet,17|12|10|12) and `skip` (stages 7, 9; or NULL) the U-Net skip tensor d2 (B,64,t_unet,12) /
 * d1 (B,32,t_unet,17) added behind the resampler (stsae_unet.py:381-383,391-393); stages 11..14 return MCD_EUNSUPPORTED there.
 * `skip` must be NULL otherwise. */
int mcd_layer_forward(const mcd_weights_t* w, int32_t stage, const float* x, const float* skip, const float* emb,
                      int32_t n_windows, float* out, void* workspace, void* stream);

/* The noise tensor the perf mode (noise == NULL) of mcd_score draws in-kernel, in mcd_score's `noise` layout
 * (S, max(ns-1,1), B, C=2, Tx, V=17): mcd_score(noise = this tensor) reproduces mcd_score(noise = NULL, seed,
 * first_window_id) bit for bit.  Replaces nothing in the reference (which draws torch.randn_like, mocodad.py:162,176);
 * it exists so that the in-kernel generator's distribution can be tested and the perf mode can be replayed by an oracle. */
int mcd_philox_noise(uint64_t seed, int64_t first_window_id, int32_t n_windows, int32_t n_samples, int32_t noise_steps,
                     int32_t n_corrupt, float* noise_out, void* stream);

/* Bytes of caller-provided device scratch a scoring call may need: condition embeddings when the condition encoder runs
 * as its own launch, (B,S) losses when an aggregation cannot be fused, scratch slabs of the runtime-shape kernels.
 * ALWAYS allocate it: workspace == NULL is accepted only by calls that end up as ONE launch (mcd_plan_split() == 1 with the
 * shipped condition encoder on a specialised frame count and a loss-based aggregation); every other form -- one trajectory
 * per workgroup for small or oddly sized batches, mcd_score without aggregation, another encoder architecture, more than 12
 * U-Net frames (the slab-tiled kernel's activations live in the workspace) -- fails with MCD_EINVAL ("workspace required")
 * without it. */
int64_t mcd_score_workspace_bytes(const mcd_weights_t* w, const mcd_score_cfg_t* cfg);

/* How the library would cut this scoring call into workgroups (a pure function of the shapes, the device and MCD_OPT_SPLIT):
 * 1 = a workgroup runs every sample of its windows -- condition encoder, trajectories and aggregation in ONE kernel launch,
 * the default for batches that fill the device (1024 windows x 5 samples on 256 CUs); n_samples = one trajectory per
 * workgroup with the encoder and the aggregation as their own launches (better fill for small batches); 0 = the call does
 * not run on score_kernel<T_u, ...> at all: 13 .. 32 U-Net frames (the slab-tiled kernel, persistent workgroups of one chain at a
 * time, aggregation as its own launch) or MCD_OPT_GENERIC_UNET.  Negative: MCD_E*. */
int32_t mcd_plan_split(const mcd_weights_t* w, const mcd_score_cfg_t* cfg);

/* Per-handle options (no environment variables; the only process-wide state is mcd_debug_set_prof's pointer).  Set them before the calls they affect, from the
 * thread that owns the handle; none is needed for normal use.
 *   MCD_OPT_VARIANT       alternative workgroup shapes of the trajectory kernel (tuning experiments only).
 *   MCD_OPT_COND_GENERIC  1: run the condition encoder through the runtime-channel-list kernel even when the shipped
 *                         architecture's MFMA kernel applies (used by the tests to cover both).
 *   MCD_OPT_GENERIC_UNET  1: run the trajectory through the plain-FMA runtime-shape kernel instead of the MFMA kernels (every
 *                         frame count 1 .. 32 has one; the tests use this to cross-check the two implementations).
 *   MCD_OPT_SPLIT         0 (default): the library chooses how many workgroups share a window's samples (1 = a workgroup
 *                         runs all samples of its windows: condition encoder and aggregation fused into the ONE launch;
 *                         n_samples = one trajectory per workgroup, better fill for odd batch sizes); n > 0 forces it
 *                         (tests). */
enum { MCD_OPT_VARIANT = 0, MCD_OPT_COND_GENERIC = 1, MCD_OPT_GENERIC_UNET = 2, MCD_OPT_SPLIT = 3,
       MCD_OPT_PHASE = 4, /* tuning experiment: start the second half of the grid `value` x 1024 clock cycles late */
       MCD_OPT_COUNT = 5 };
int mcd_set_option(mcd_weights_t* w, int32_t option, int32_t value);

/* Replaces: the hot loop of MoCoDAD.forward (mocodad.py:155-180) + the per-sample loss of :484.
 *   data        (B,C,T,V) windows
 *   noise       NULL -> in-kernel Philox4x32-10 keyed by (seed, first_window_id+b, s, step, element or joint-pair group);
 *               else (S, max(ns-1,1), B, C, Tx, V): slot 0 = x_T, slot k = z added at step i = ns-k
 *               (what torch.randn_like returns at mocodad.py:162,176 in call order)
 *   step_table  (ns, 4+emb_dim): row i = [1/sqrt(alpha_i), (1-alpha_i)/sqrt(1-alpha_hat_i), sqrt(beta_i), 0,
 *               pos_encoding(i)[0..emb_dim)]  (mocodad.py:172-178, stsae_unet.py:173-179)
 *   loss_out    (B,S)  per-sample window loss
 *   pose_out    NULL or (B,S,C,Tx,V) generated x_0
 * Chains are independent, as in the reference: a chain that diverges (NaN / Inf activations; a NaN in `noise`) returns a
 * non-finite loss and changes no other chain's result -- the persistent kernels clear their state behind it and re-run the
 * chains that shared its passes on their own.  PRECONDITION: `data` is finite.  (A window with non-finite poses scores NaN, as
 * in the reference, but can take the other windows of its workgroup -- up to 4 neighbours in the batch -- with it.) */
int mcd_score(const mcd_weights_t* w, const mcd_score_cfg_t* cfg, const float* data, const float* noise,
              uint64_t seed, int64_t first_window_id, const float* step_table, void* workspace,
              float* loss_out, float* pose_out, void* stream);

/* Window view: lets the kernels read windows straight out of per-person trajectory buffers
 * and apply the dataset's test-time affine transform while loading, instead of the host materialising
 * seg_len x num_transform copies (reference: sliding windows utils/preprocessing.py:14-86, transforms
 * utils/dataset_utils.py:255-310 applied per item in utils/dataset.py:67-76).
 * Element (b, c, t, v) of window b is data[base[b] + c*stride_c + t*stride_t + v], then, when trans != NULL,
 * [x', y'] = A[trans[b]] @ [x, y, 1] with A = affine + 6*trans[b] = rows [a00 a01 a02 a10 a11 a12].
 * The view also carries the per-window condition-frame sets of the random_imp strategy. */
typedef struct {
    const int64_t* base;      /* device (B,) element offsets; NULL = dense (B,C,T,V) tensor */
    int64_t stride_c;         /* elements between the two coordinates of one joint */
    int64_t stride_t;         /* elements between consecutive frames */
    const int32_t* trans;     /* device (B,) transform index per window, or NULL */
    const float* affine;      /* device (n_transform, 6), required when trans != NULL */
    const int32_t* cond_mask; /* MCD_STRATEGY_RANDOM_IMP only: device (B,), bit t set = frame t of the window is a condition
                                 frame (exactly n_cond bits; mocodad.py:719-724 keeps both frame subsets in ascending
                                 order, so the mask is all the kernel needs); NULL otherwise */
} mcd_window_view_t;

/* mcd_score with a window view (view == NULL is exactly mcd_score). */
int mcd_score_view(const mcd_weights_t* w, const mcd_score_cfg_t* cfg, const float* data, const mcd_window_view_t* view,
                   const float* noise, uint64_t seed, int64_t first_window_id, const float* step_table, void* workspace,
                   float* loss_out, float* pose_out, void* stream);

/* mcd_score_view + the loss-based aggregation over the samples (mocodad.py:489-492,504-516: MCD_AGGR_BEST / WORST / MEAN /
 * MEDIAN / QUANTILE) in ONE call -- and, whenever the workgroups own whole windows, in ONE kernel launch: the condition
 * encoder runs in the workgroup that owns the window, the per-sample losses stay in its LDS, loss_agg (B,) is all that
 * is written.  loss_all (B,S) and pose_out are optional extra outputs (NULL = not wanted).  The *_pose strategies need the
 * generated poses of all samples: mcd_score + mcd_aggregate.  Any n_samples >= 1 (the reference's shipped test configs use
 * 50, config/<dataset>/mocodad_test.yaml): a workgroup keeps up to 64 per-sample losses of a window in LDS; with more samples the
 * per-sample losses go through the workspace and the aggregation runs as its own launch (as for mcd_plan_split() != 1). */
int mcd_score_fused(const mcd_weights_t* w, const mcd_score_cfg_t* cfg, const float* data, const mcd_window_view_t* view,
                    const float* noise, uint64_t seed, int64_t first_window_id, const float* step_table, void* workspace,
                    int32_t aggregation, float quantile, float* loss_agg, float* loss_all, float* pose_out, void* stream);

/* Replaces: MoCoDAD._aggregation_strategy (mocodad.py:454-520) on the (B,S) losses / (B,S,C,Tx,V) poses.
 * data/cfg give the ground-truth corrupt frames for the *_pose strategies.  loss_agg (B,), pose_agg
 * NULL or (B,C,Tx,V).  MCD_AGGR_ALL is the identity and is not handled here.  Any n_samples >= 1, like the reference: one
 * wave per window, order statistics (median = torch's lower middle value, quantile = torch's linear interpolation) by rank
 * counting over the samples, best / worst with the reference's strict comparisons (the first of equal samples is kept). */
int mcd_aggregate(const mcd_score_cfg_t* cfg, int32_t num_coords, int32_t n_joints, int32_t strategy, float quantile,
                  const float* loss_all, const float* pose_all, const float* data, float* loss_agg,
                  float* pose_agg, void* stream);

/* Frame-score assembly that follows the path (mocodad.py:386-401 + eval_utils.py:27-34): scatter-max of
 * window scores to their frames.  scores (N,), frames (N,seg_len) 1-based int32, row (N,) int32 = output
 * row (one per (transform, clip, person)), out (n_rows, n_frames) pre-zeroed by the callee. */
int mcd_scatter_max(const float* scores, const int32_t* frames, const int32_t* row, int64_t n, int32_t seg_len,
                    int32_t n_rows, int32_t n_frames, float* out, void* stream);

/* Frame-score assembly after the path, whole: replaces the (transform, clip, person) loops of
 * MoCoDAD.post_processing (mocodad.py:362-425) with compute_var_matrix + np.nanmax (eval_utils.py:27-34, mocodad.py:392-393),
 * pad_scores (eval_utils.py:133-149), the person aggregation mean + (max - min of log1p) (mocodad.py:401-403), the HR frame
 * masks (:405-413), score_process = shift + scipy gaussian_filter1d(sigma = filter_kernel_size; truncate 4, 'reflect')
 * (eval_utils.py:100-106) and the mean over the transforms (:422); float64 like the NumPy code.  What is left for the host is
 * roc_auc_score.  All table pointers are DEVICE memory built once per dataset by the caller. */
typedef struct {
    int32_t n_clips;              /* ground-truth files, in sorted file-name order */
    int32_t num_transform;
    int32_t n_persons;            /* dense person-id range: ids 0 .. n_persons-1 (max id + 1) */
    int32_t max_frames;           /* row stride: max over the clips of len(gt) */
    int32_t pad_size;             /* anomaly_score_pad_size, -1 = no padding */
    int32_t frames_shift;         /* anomaly_score_frames_shift (>= 1) */
    int32_t gauss_radius;         /* int(4 * sigma + 0.5) */
    const int64_t* clip_keys;     /* (n_clips,) ascending (scene << 32 | clip) */
    const int32_t* clip_n_frames; /* (n_clips,) len(gt) */
    const int32_t* frame_dst;     /* (n_clips, max_frames) position of the frame in the clip's output after the HR masks, -1 = dropped */
    const int32_t* clip_out_len;  /* (n_clips,) frames kept */
    const int64_t* clip_out_off;  /* (n_clips,) offset of the clip's scores in `out` (sorted file-name order, concatenated) */
    const double* gauss_weights;  /* (2 * gauss_radius + 1,) scipy's normalised kernel */
} mcd_frame_cfg_t;
int64_t mcd_frame_scores_workspace_bytes(const mcd_frame_cfg_t* cfg);
/* scores (N,) f32, trans (N,) i64, meta (N,4) i64 = [scene, clip, person, first_frame], frames (N,seg_len) i32 1-based
 * (the arrays MoCoDAD.post_processing receives, on the device) -> out (sum clip_out_len,) f64 = `pds` of mocodad.py:422. */
int mcd_frame_scores(const mcd_frame_cfg_t* cfg, const float* scores, const int64_t* trans, const int64_t* meta,
                     const int32_t* frames, int64_t n_windows, int32_t seg_len, void* workspace, double* out, void* stream);

/* Profiling builds only (-DMCD_PROFILE, tools/stage_profile.py): device buffer of 96 uint64 per-stage cycle accumulators
 * written by workgroup 0 of the trajectory kernel; NULL (the default) disables it.  A no-op in the shipped build. */
void mcd_debug_set_prof(void* device_buffer);

/* Test aid: fills the LDS of every CU with NaN bit patterns (4096 workgroups of 160 KB on `stream`), so that a later kernel
 * that reads shared memory it never wrote yields NaNs instead of whatever the previous kernel left behind
 * (tests/test_hip_parity.py::test_no_uninitialised_reads). */
int mcd_debug_poison_lds(void* stream);

const char* mcd_last_error(void);
int32_t mcd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MOCODAD_HIP_H */
