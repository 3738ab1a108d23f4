"""Drop-in for the reference's models/mocodad.py::MoCoDAD on MI355X.

Same constructor (an argparse.Namespace with the YAML keys of config/*/mocodad_test.yaml), same
`forward / test_step / on_test_epoch_* / validation_* / post_processing / test_on_saved_tensors` surface and
the same state_dict key layout (a Lightning checkpoint's 'state_dict' loads verbatim) — but `forward` runs the
reverse-diffusion scoring loop in the hand-written HIP kernels behind the C ABI (mocodad_amd.engine).

Reference: /root/reference/models/mocodad.py (forward :129-184, _aggregation_strategy :454-520,
_set_conditioning_strategy :753-796, _select_frames :708-750, post_processing :337-430).

Scope: inference scoring with the 'inject', 'concat', 'no_condition' strategies, the
'AE' / 'E' / 'E_unet' condition encoders and the 'inbetween_imp' / 'random_imp' imputation strategies.  Training
(training_step / configure_optimizers) is outside the accelerated path and raises NotImplementedError.
"""
import argparse
import os
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..utils.diffusion_utils import Diffusion
from ..utils.eval_utils import post_process_scores
from ..utils.model_utils import processing_data

try:  # Lightning is optional: the reference needs it, this image does not have it
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # pragma: no cover - exercised in this image
    class _Base(nn.Module):
        """Minimal stand-in for pl.LightningModule (device tracking, log, save_hyperparameters)."""

        def __init__(self):
            super().__init__()
            self._device = torch.device("cpu")
            self.logged: Dict[str, float] = {}

        @property
        def device(self):
            return self._device

        def _apply(self, fn, *a, **k):
            out = super()._apply(fn, *a, **k)
            for p in self.parameters():
                self._device = p.device
                break
            return out

        def log(self, name, value, **kw):
            self.logged[name] = float(value)

        def save_hyperparameters(self, *a, **k):
            self.hparams = a[0] if a else None

        def on_test_epoch_start(self):
            pass

        def on_validation_epoch_start(self):
            pass


# ------------------------------------------------------------------ parameter containers
# These modules only HOLD parameters under the reference's names (stsgcn.py:47-91,134-141,178-184); the math
# is done by the HIP kernels from the packed state_dict, so they define no forward.
class _GraphMix(nn.Module):
    def __init__(self, time_dim: int, joints_dim: int):
        super().__init__()
        self.A = nn.Parameter(torch.empty(time_dim, joints_dim, joints_dim).uniform_(-1, 1) / joints_dim ** 0.5)
        self.T = nn.Parameter(torch.empty(joints_dim, time_dim, time_dim).uniform_(-1, 1) / time_dim ** 0.5)


class STGCNParams(nn.Module):
    def __init__(self, cin: int, cout: int, time_dim: int, joints_dim: int, dropout: float, emb_dim: Optional[int]):
        super().__init__()
        self.in_channels, self.out_channels = cin, cout
        self.gcn = _GraphMix(time_dim, joints_dim)
        self.tcn = nn.Sequential(nn.Conv2d(cin, cout, (1, 1)), nn.BatchNorm2d(cout), nn.Dropout(dropout, inplace=True))
        self.residual = nn.Sequential(nn.Conv2d(cin, cout, 1), nn.BatchNorm2d(cout)) if cin != cout else nn.Identity()
        self.prelu = nn.PReLU()
        if emb_dim is not None:
            self.emb_layer = nn.Sequential(nn.SiLU(), nn.Linear(emb_dim, cout))


class JointResampleParams(nn.Module):
    def __init__(self, vin: int, vout: int, dropout: float):
        super().__init__()
        self.block = nn.Sequential(nn.Conv2d(vin, vout, (1, 1)), nn.BatchNorm2d(vout), nn.Dropout(dropout, inplace=True))


def _stack(chans: List[Tuple[int, int]], T: int, V: int, dropout: float, emb: Optional[int]) -> nn.ModuleList:
    return nn.ModuleList(STGCNParams(a, b, T, V, dropout, emb) for a, b in chans)


class UNetParams(nn.Module):
    """Parameter tree of STSAE_Unet (stsae_unet.py:50-156,294-363): joints 17 -> 12 -> 10 -> 12 -> 17."""
    down_channels = [16, 32, 32, 64, 64, 128, 64]
    up_channels = [64, 32, 32, 2]

    def __init__(self, c_in: int, embedding_dim: int, n_frames: int, dropout: float):
        super().__init__()
        d, u, T, e = self.down_channels, self.up_channels, n_frames, embedding_dim
        self.st_gcnnsp1a = _stack([(c_in, d[0])], T, 17, dropout, e)
        self.st_gcnnsd1 = _stack([(d[0], d[1]), (d[1], d[2])], T, 17, dropout, e)
        self.st_gcnnsd2 = _stack([(d[2], d[3]), (d[3], d[4])], T, 12, dropout, e)
        self.st_gcnnsd3 = _stack([(d[4], d[5]), (d[5], d[6])], T, 10, dropout, e)
        self.down1 = JointResampleParams(17, 12, dropout)
        self.down2 = JointResampleParams(12, 10, dropout)
        self.st_gcnnsu4 = _stack([(d[6], u[0]), (u[0], u[1])], T, 12, dropout, e)
        self.st_gcnnsu3 = _stack([(u[1], u[2]), (u[2], u[3])], T, 17, dropout, e)
        self.up2 = JointResampleParams(12, 17, dropout)
        self.up3 = JointResampleParams(10, 12, dropout)


class CondUNetParams(nn.Module):
    """Parameter tree of STSE_Unet used as the 'E_unet' condition encoder (stsae_unet.py:8-146; mocodad.py:110-114:
    embedding_dim=None, set_out_layer=True): the U-Net's down path ending in 6 channels + to_time_dim."""
    down_channels = [16, 32, 32, 64, 64, 128, 6]

    def __init__(self, c_in: int, latent_dim: int, n_frames: int, dropout: float):
        super().__init__()
        d, T = self.down_channels, n_frames
        self.st_gcnnsp1a = _stack([(c_in, d[0])], T, 17, dropout, None)
        self.st_gcnnsd1 = _stack([(d[0], d[1]), (d[1], d[2])], T, 17, dropout, None)
        self.st_gcnnsd2 = _stack([(d[2], d[3]), (d[3], d[4])], T, 12, dropout, None)
        self.st_gcnnsd3 = _stack([(d[4], d[5]), (d[5], d[6])], T, 10, dropout, None)
        self.down1 = JointResampleParams(17, 12, dropout)
        self.down2 = JointResampleParams(12, 10, dropout)
        self.to_time_dim = nn.Linear(d[6] * T * 10, latent_dim)


class _LayerList(nn.Module):
    def __init__(self, chans, T, V, dropout):
        super().__init__()
        self.model_layers = _stack(chans, T, V, dropout, None)


class CondEncoderParams(nn.Module):
    """Parameter tree of STSE / STSAE (stsae.py:43-55,137-146; components.py:41-66,122-146)."""

    def __init__(self, c_in, h_dim, latent_dim, n_frames, n_joints, layer_channels, dropout, with_decoder: bool):
        super().__init__()
        enc = list(layer_channels) + [h_dim]
        self.channels = enc
        self.encoder = _LayerList(list(zip([c_in] + enc[:-1], enc)), n_frames, n_joints, dropout)
        self.btlnk = nn.Linear(h_dim * n_frames * n_joints, latent_dim)
        if with_decoder:  # parameters only: the decoder is dead work at inference (mocodad.py:157)
            dec = list(layer_channels)[::-1] + [c_in]
            self.decoder = _LayerList(list(zip([h_dim] + dec[:-1], dec)), n_frames, n_joints, dropout)
            self.rev_btlnk = nn.Linear(latent_dim, h_dim * n_frames * n_joints)


# ------------------------------------------------------------------ the module
class MoCoDAD(_Base):
    losses = {"l1": "l1", "smooth_l1": "smooth_l1", "mse": "mse"}
    conditioning_strategies = {"cat": "concat", "concat": "concat", "add2layers": "inject", "inject": "inject",
                               "inbetween_imp": "inbetween_imp", "interleave": "inbetween_imp",
                               "random_indices": "random_imp", "random_imp": "random_imp",
                               "no_condition": "no_condition", "none": "no_condition"}

    def __init__(self, args: argparse.Namespace) -> None:
        super().__init__()
        self.save_hyperparameters(args)
        g = lambda k, d=None: getattr(args, k, d)
        self.n_frames = args.seg_len
        self.num_coords = args.num_coords
        self.n_joints = 14 if g("headless", False) else 18 if g("kp18_format", False) else 17
        self.embedding_dim = args.embedding_dim
        self.dropout = args.dropout
        self.conditioning_strategy = self.conditioning_strategies[args.conditioning_strategy]
        self.conditioning_indices = args.conditioning_indices
        self.n_frames_condition, self.n_frames_corrupt, self.input_n_frames = self._set_conditioning_strategy()
        self.conditioning_architecture = args.conditioning_architecture if self.conditioning_strategy == "inject" else None
        self.cond_h_dim, self.cond_latent_dim = g("h_dim"), g("latent_dim")
        self.cond_channels, self.cond_dropout = g("channels"), args.dropout
        self.learning_rate = g("opt_lr")
        self.loss_name = self.losses[g("loss_fn", "smooth_l1")]
        self.rec_weight = g("rec_weight")
        self.noise_steps = args.noise_steps
        self.aggregation_strategy = g("aggregation_strategy", "best")
        self.n_generated_samples = g("n_generated_samples", 1)
        self.model_return_value = g("model_return_value", "loss")
        self.gt_path, self.split, self.use_hr = g("gt_path"), g("split", "test"), g("use_hr", False)
        self.ckpt_dir, self.save_tensors = g("ckpt_dir"), g("save_tensors", False)
        self.num_transforms = g("num_transform", 1)
        self.anomaly_score_pad_size = g("pad_size", -1)
        self.anomaly_score_filter_kernel_size = g("filter_kernel_size", 1)
        self.anomaly_score_frames_shift = g("frames_shift", 0)
        self.dataset_name = g("dataset_choice")
        self.seed = int(g("seed", 0) or 0)
        self._set_diffusion_variables()
        self.build_model()
        self._scorer = None
        self._scorer_key = None
        self._calls = 0
        self.shard = None  # optional mocodad_amd.parallel.WindowShard set by the multi-GPU driver
        self.hip_options: Dict[str, int] = {}   # per-handle library switches (engine.HipScorer.set_option), e.g. {"split": 1}

    # -------------------------------------------------------------- construction
    def build_model(self) -> None:
        if self.num_coords != 2 or self.n_joints != 17:
            raise NotImplementedError("the HIP path (like the reference U-Net) supports num_coords=2 and 17 joints")
        enc = None
        if self.conditioning_strategy == "inject":
            arch = self.conditioning_architecture
            if arch not in ("AE", "E", "E_unet"):
                raise NotImplementedError(f"Conditioning architecture {arch} not implemented.")
            if arch == "E_unet":
                enc = CondUNetParams(self.num_coords, self.cond_latent_dim, self.n_frames_condition, self.cond_dropout)
            else:
                enc = CondEncoderParams(self.num_coords, self.cond_h_dim, self.cond_latent_dim, self.n_frames_condition,
                                        self.n_joints, self.cond_channels, self.cond_dropout, with_decoder=(arch == "AE"))
        self.condition_encoder = enc
        self.model = UNetParams(self.num_coords, self.embedding_dim, self.input_n_frames, self.dropout)
        self.eval()

    def _set_conditioning_strategy(self) -> Tuple[int, int, int]:
        s, T, ci = self.conditioning_strategy, self.n_frames, self.conditioning_indices
        inp = T
        if s == "no_condition":
            nc = 0
        elif s == "random_imp":
            assert isinstance(ci, int), "Random imputation requires an integer number of frames to condition on, not a list of indices"
            nc = ci
        elif s == "inbetween_imp":
            nc = T // ci if isinstance(ci, int) else len(ci)
        elif s in ("concat", "inject"):
            if isinstance(ci, int):
                nc = T // ci
            else:
                assert ci == list(range(min(ci), max(ci) + 1)), "Conditioning indices must be a list of consecutive integers"
                assert min(ci) == 0 or max(ci) == T - 1, "Conditioning indices must start from 0 or end at the last frame"
                nc = len(ci)
            inp = T - nc if s == "inject" else T
        else:
            raise NotImplementedError(f"Conditioning strategy {s} not implemented")
        return nc, T - nc, inp

    def _frame_split(self) -> Tuple[List[int], List[int]]:
        """(cond_idx, corrupt_idx) exactly as _select_frames builds them (mocodad.py:726-748)."""
        T, ci = self.n_frames, self.conditioning_indices
        if self.conditioning_strategy == "no_condition":
            return [], list(range(T))
        if self.conditioning_strategy == "random_imp":   # only the counts matter: the sets are drawn per window in forward
            return list(range(ci)), list(range(ci, T))
        if isinstance(ci, int):
            if self.conditioning_strategy == "inbetween_imp":
                cond = list(range(0, T, ci))
                return cond, [i for i in range(T) if i not in cond]
            n = T // ci
            return list(range(n)), list(range(n, T))
        return list(ci), [i for i in range(T) if i not in ci]

    def _set_diffusion_variables(self) -> None:
        self.noise_scheduler = Diffusion(noise_steps=self.noise_steps, n_joints=self.n_joints, device="cpu", time=self.n_frames)
        self._beta_ = self.noise_scheduler.schedule_noise()
        self._alpha_ = 1.0 - self._beta_
        self._alpha_hat_ = torch.cumprod(self._alpha_, dim=0)

    @property
    def _beta(self):
        return self._beta_.to(self.device)

    @property
    def _alpha(self):
        return self._alpha_.to(self.device)

    @property
    def _alpha_hat(self):
        return self._alpha_hat_.to(self.device)

    # -------------------------------------------------------------- HIP engine
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._scorer = None  # weights changed -> repack lazily
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def scorer(self):
        """Packed-weights handle on the module's current device (built lazily, rebuilt after load_state_dict)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("MoCoDAD (mocodad_amd) scores on an MI355X only: move the module to a cuda device "
                               "(there is no CPU fallback)")
        key = str(dev)
        if self._scorer is None or self._scorer_key != key:
            from ..engine import HipScorer
            ci, xi = self._frame_split()
            unet_enc = isinstance(self.condition_encoder, CondUNetParams)
            chans = list(self.condition_encoder.channels) if self.condition_encoder is not None and not unet_enc else []
            self._scorer = HipScorer(self.state_dict(), strategy=self.conditioning_strategy, seg_len=self.n_frames,
                                     cond_idx=ci, corrupt_idx=xi, cond_channels=chans, cond_unet=unet_enc,
                                     num_coords=self.num_coords, n_joints=self.n_joints, emb_dim=self.embedding_dim, device=dev,
                                     options=self.hip_options)
            self._scorer_key = key
        return self._scorer

    # -------------------------------------------------------------- forward
    def forward(self, input_data: List[torch.Tensor], aggr_strategy: str = None, return_: str = None, *,
                noise: Optional[torch.Tensor] = None, window_offset: Optional[int] = None,
                cond_mask: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """[data (B,C,T,V), transformation_idx, metadata, actual_frames] -> [loss and/or pose] + the inputs.

        noise (extension, keyword only): (S, max(ns-1,1), B, C, Tx, V) tensor replacing the in-kernel Philox
        stream, slot 0 = x_T, slot k = z of step ns-k — what torch.randn_like returns in the reference in call
        order; used for parity tests.  window_offset: global index of the first window (keys the noise stream).
        cond_mask ('random_imp' only): (B,) bitmasks of the condition frames; default = drawn like the reference does
        (one torch.randperm per window on the default CPU generator, mocodad.py:719-724)."""
        tensor_data, meta_out = self._unpack_data(input_data)
        aggr = self.aggregation_strategy if aggr_strategy is None else aggr_strategy
        ret = return_ if return_ is not None else self.model_return_value
        if ret is None:
            raise ValueError("Either return_ or self.model_return_value must be set")
        S, ns = self.n_generated_samples, self.noise_steps
        sc = self.scorer()
        pose_aggr = aggr in ("all", "random", "mean_pose", "median_pose")
        want_pose = ret in ("pose", "all") or pose_aggr
        if window_offset is None:
            window_offset = self._calls
        self._calls += tensor_data.shape[0]
        if hasattr(tensor_data, "as_view") and pose_aggr and aggr in ("mean_pose", "median_pose"):
            tensor_data = tensor_data.materialize()      # the *_pose strategies compare against the windows themselves
        if self.conditioning_strategy == "random_imp":
            if aggr in ("mean_pose", "median_pose"):
                raise NotImplementedError("the *_pose aggregations are not available with 'random_imp'")
            if cond_mask is None:
                cond_mask = self.draw_random_imp_mask(tensor_data.shape[0])
        kw = dict(n_samples=S, noise_steps=ns, noise=noise, seed=self.seed, first_window_id=window_offset, loss_fn=self.loss_name,
                  cond_mask=cond_mask)
        fusable = aggr in ("best", "worst", "mean", "median") or "quantile" in aggr
        if fusable and not want_pose:
            # loss-only output: trajectories, condition encoder and the aggregation over the samples in ONE launch
            loss, _, _ = sc.score_fused(tensor_data, aggregation=aggr, **kw)
            selected_x = None
        else:
            loss_all, poses_all = sc.score(tensor_data, want_poses=want_pose, **kw)
            selected_x, loss = self._aggregate(sc, tensor_data, loss_all, poses_all, aggr, want_pose)
        return self._pack_out_data(selected_x, loss, [tensor_data] + meta_out, return_=ret)

    def draw_random_imp_mask(self, n_windows: int) -> torch.Tensor:
        """'random_imp' frame sets exactly as _select_frames draws them (mocodad.py:719-724, 535): one randperm per
        window; frame t conditions iff perm[t] < conditioning_indices.  -> (B,) int32 bitmasks."""
        T, k = self.n_frames, int(self.conditioning_indices)
        idx = torch.tensor([torch.randperm(T).tolist() for _ in range(n_windows)])
        return ((idx < k).int() << torch.arange(T, dtype=torch.int32)).sum(1).to(torch.int32)

    def _aggregate(self, sc, data, loss_all, poses_all, aggr: str, want_pose: bool):
        if aggr == "all":
            return poses_all, loss_all
        if aggr == "random":  # the reference returns a bare tensor here (mocodad.py:480-481); return (pose, its loss)
            s = int(np.random.randint(loss_all.shape[1]))
            return poses_all[:, s], loss_all[:, s]
        known = ("best", "worst", "mean", "median", "mean_pose", "median_pose")
        if aggr not in known and "quantile" not in aggr:
            raise ValueError(f"Unknown aggregation strategy {aggr}")
        return sc.aggregate(data, loss_all, poses_all, aggr, noise_steps=self.noise_steps, loss_fn=self.loss_name,
                            want_pose=want_pose)

    def _pack_out_data(self, selected_x, loss_of_selected_x, additional_out, return_: str):
        if return_ is None:
            if self.model_return_value is None:
                raise ValueError("Either return_ or self.model_return_value must be set")
            return_ = self.model_return_value
        if return_ == "pose":
            out = [selected_x]
        elif return_ == "loss":
            out = [loss_of_selected_x]
        elif return_ == "all":
            out = [loss_of_selected_x, selected_x]
        else:
            raise ValueError(f"Unknown return mode {return_}")
        return out + additional_out

    def _unpack_data(self, x):
        # x[0] is the (B,C,T,V) tensor of the reference, or a WindowBatch view into trajectories (data/windows.py)
        return x[0].to(self.device), [x[1], x[2], x[3]]

    # -------------------------------------------------------------- test / validation loops
    def test_step(self, batch, batch_idx: int) -> None:
        self._test_output_list.append(self.forward(batch))

    def on_test_epoch_start(self) -> None:
        super().on_test_epoch_start()
        self._test_output_list = []
        self._calls = 0

    def on_test_epoch_end(self) -> float:
        return self._epoch_end("_test_output_list")

    def validation_step(self, batch, batch_idx: int) -> None:
        self._validation_output_list.append(self.forward(batch))

    def on_validation_epoch_start(self) -> None:
        super().on_validation_epoch_start()
        self._validation_output_list = []
        self._calls = 0

    def on_validation_epoch_end(self) -> float:
        return self._epoch_end("_validation_output_list")

    def _epoch_end(self, attr: str) -> float:
        outs = getattr(self, attr)
        delattr(self, attr)
        # The test_step loop only ENQUEUES device work: when it returns the GPU still has most of the epoch's batches in front of
        # it.  The host-side first-use work of the post-processing -- reading the ground-truth masks, building the frame tables --
        # is done NOW, under that queue, before anything below waits for the scores (it used to be a constant ~0.1 s behind them).
        import time as _time
        _t = [_time.perf_counter()]
        _tick = lambda: _t.append(_time.perf_counter())
        if self.device.type == "cuda" and self.anomaly_score_frames_shift >= 1 and (self.shard is None or self.shard.rank == 0):
            try:
                self._frame_assembler()
            except (OSError, ValueError, KeyError):
                pass        # (no / unreadable ground truth: post_processing below reports it where it always did)
        _tick()
        if self.shard is not None:
            # multi-GPU: every rank scored its contiguous window shard (possibly an empty one); ONE all-gather reassembles
            # the per-window scores, then rank 0 alone runs the post-processing and the AUC (the other ranks return nan)
            if self.aggregation_strategy == "all" or self.model_return_value != "loss":
                raise ValueError("sharded evaluation exchanges one score per window: use model_return_value='loss' and an "
                                 "aggregation strategy other than 'all'")
            local = torch.cat([o[0].reshape(-1) for o in outs]) if outs else torch.empty(0, dtype=torch.float32, device=self.device)
            out, trans, meta, frames = self.shard.gather(local, None, None, None, device=self.device)
            gt_data = None      # windows stay on their ranks (post_processing does not use them)
            if self.shard.rank != 0:
                return float("nan")
        else:
            if not outs:
                raise ValueError("no batches were scored")
            out, gt_data, trans, meta, frames = processing_data(outs)
        _tick()
        self.last_scores = np.asarray(out)     # the (gathered) per-window scores of this epoch, in dataset order
        if self.save_tensors:
            self._save_tensors({"prediction": out, "gt_data": gt_data, "trans": trans, "metadata": meta, "frames": frames},
                               split_name=self.split, aggr_strategy=self.aggregation_strategy, n_gen=self.n_generated_samples)
        auc = self.post_processing(out, gt_data, trans, meta, frames)
        _tick()
        if os.environ.get("MCD_EPOCH_TIMING"):      # (diagnostic: where an epoch end's time goes)
            import sys
            print("epoch end: frame tables %.1f ms | collate / gather %.1f ms | post-processing + AUC %.1f ms" % tuple(
                1e3 * (b - a) for a, b in zip(_t[:-1], _t[1:])), file=sys.stderr)
        self.log("AUC", auc)
        return auc

    # -------------------------------------------------------------- after the path: scores -> AUC
    def _gt_and_masks(self):
        from ..utils.eval_utils import get_avenue_mask, get_hr_ubnormal_mask
        names = sorted(f for f in os.listdir(self.gt_path) if f.endswith(".npy"))
        gts = {(int(f.split("_")[0]), int(f.split("_")[1].split(".")[0])): np.load(os.path.join(self.gt_path, f)) for f in names}
        masks = {}
        if self.use_hr and self.dataset_name == "UBnormal":
            masks = dict(get_hr_ubnormal_mask(self.split))
        if self.dataset_name == "HR-Avenue":
            masks.update({("clip", k): v for k, v in get_avenue_mask().items()})
        return gts, masks

    def _frame_assembler(self):
        """Device frame-score assembly for the current dataset settings (tables built once, rebuilt when a setting changes)."""
        key = (self.gt_path, self.dataset_name, self.use_hr, self.split, self.num_transforms, self.anomaly_score_pad_size,
               self.anomaly_score_filter_kernel_size, self.anomaly_score_frames_shift, str(self.device))
        if getattr(self, "_asm_key", None) != key:
            from ..engine import FrameScoreAssembler
            gts, masks = self._gt_and_masks()
            self._asm = FrameScoreAssembler(gts, masks, num_transform=self.num_transforms, pad_size=self.anomaly_score_pad_size,
                                            filter_kernel_size=self.anomaly_score_filter_kernel_size,
                                            frames_shift=self.anomaly_score_frames_shift, device=self.device)
            self._asm_key = key
        return self._asm

    def post_processing(self, out, gt_data, trans, meta, frames) -> float:
        """Window scores -> AUC (mocodad.py:337-430).  On a GPU the whole frame-score assembly runs in mcd_frame_scores and
        the host keeps roc_auc_score; without one (saved tensors evaluated offline) the NumPy path below does it.
        out / trans / meta / frames: NumPy arrays (the reference's signature) or tensors; gt_data is unused, as in the reference."""
        from sklearn.metrics import roc_auc_score
        if self.device.type == "cuda" and self.anomaly_score_frames_shift >= 1:
            import time as _time
            _a = _time.perf_counter()
            asm = self._frame_assembler()
            pds = asm(out, trans, meta, frames)
            _b = _time.perf_counter()
            if pds is not None:
                auc = float(roc_auc_score(asm.gt, pds))
                if os.environ.get("MCD_EPOCH_TIMING"):
                    import sys
                    print("post-processing: frame scores %.1f ms | roc_auc_score over %d frames %.1f ms" % (1e3 * (_b - _a), len(pds), 1e3 * (_time.perf_counter() - _b)), file=sys.stderr)
                return auc
        gts, masks = self._gt_and_masks()
        _np = lambda a: a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
        pds, gt = post_process_scores(_np(out), _np(trans), _np(meta), _np(frames), gts,
                                      num_transform=self.num_transforms, pad_size=self.anomaly_score_pad_size,
                                      filter_kernel_size=self.anomaly_score_filter_kernel_size,
                                      frames_shift=self.anomaly_score_frames_shift, masks=masks,
                                      scatter_max=self._scorer.scatter_max if self._scorer is not None else None)
        return float(roc_auc_score(gt, pds))

    def test_on_saved_tensors(self, split_name: str) -> float:
        t = self._load_tensors(split_name, self.aggregation_strategy, self.n_generated_samples)
        auc = self.post_processing(t["prediction"], t["gt_data"], t["trans"], t["metadata"], t["frames"])
        print(f"AUC score: {auc:.6f}")
        return auc

    def _tensor_dir(self, split_name, aggr_strategy, n_gen):
        return os.path.join(self.ckpt_dir, "saved_tensors_{}_{}_{}".format(split_name, aggr_strategy, n_gen))

    def _save_tensors(self, tensors, split_name, aggr_strategy, n_gen) -> None:
        path = self._tensor_dir(split_name, aggr_strategy, n_gen)
        os.makedirs(path, exist_ok=True)
        for name, t in tensors.items():
            torch.save(t, os.path.join(path, name + ".pt"))

    def _load_tensors(self, split_name, aggr_strategy, n_gen):
        path = self._tensor_dir(split_name, aggr_strategy, n_gen)
        return {f.split(".")[0]: torch.load(os.path.join(path, f), weights_only=False) for f in os.listdir(path)}

    # -------------------------------------------------------------- out of the accelerated scope
    def training_step(self, batch, batch_idx):
        raise NotImplementedError("mocodad_amd accelerates inference scoring only; train with the reference implementation")

    def configure_optimizers(self):
        raise NotImplementedError("mocodad_amd accelerates inference scoring only; train with the reference implementation")
