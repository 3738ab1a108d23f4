"""Host-side driver of the HIP scoring path: owns the packed weights handle and the small per-config
device tables, and turns torch CUDA tensors into raw pointers for the C ABI (include/mocodad_hip.h).
PyTorch is used for device memory and streams only; every FLOP of the path runs in libmocodad_hip.so."""
import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .utils.diffusion_utils import step_table


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.to(device=device, dtype=torch.float32).contiguous()


def _quantile_of(strategy: str) -> float:
    """'quantile:q' -> q, rejected outside [0, 1] like torch.quantile does (mocodad.py:513-516)."""
    try:
        q = float(strategy.split(":")[-1])
    except ValueError:
        raise ValueError(f"bad quantile in aggregation strategy {strategy!r}") from None
    if not 0.0 <= q <= 1.0:      # (NaN fails too)
        raise ValueError(f"quantile() q must be in the range [0, 1], got {q} ({strategy!r})")
    return q


class HipScorer:
    """One packed model on one GPU.

    state_dict: the reference's Lightning-checkpoint keys ('model.*', 'condition_encoder.*') -> tensors.
    strategy: canonical conditioning strategy ('inject' | 'concat' | 'no_condition' | 'inbetween_imp').
    cond_channels: output channels of the 'AE' / 'E' condition encoder's layers; cond_unet: 'E_unet' encoder instead.
    """

    def __init__(self, state_dict: Dict[str, torch.Tensor], *, strategy: str, seg_len: int, cond_idx: Sequence[int],
                 corrupt_idx: Sequence[int], cond_channels: Sequence[int] = (), cond_unet: bool = False,
                 num_coords: int = 2, n_joints: int = 17, emb_dim: int = 16, device=None,
                 options: Optional[Dict[str, int]] = None):
        self.L = _lib.lib()
        if not torch.cuda.is_available():
            raise RuntimeError("mocodad_amd needs an MI355X (gfx950) GPU: the scoring path has no CPU fallback")
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.strategy = strategy
        self.seg_len = int(seg_len)
        self.cond_idx = [int(i) for i in cond_idx]
        self.corrupt_idx = [int(i) for i in corrupt_idx]
        self.num_coords, self.n_joints, self.emb_dim = num_coords, n_joints, emb_dim
        self.t_cond = len(self.cond_idx) if strategy == "inject" else 0
        self.t_unet = len(self.corrupt_idx) + (len(self.cond_idx) if strategy in ("concat", "inbetween_imp", "random_imp") else 0)
        self._tables: Dict[int, torch.Tensor] = {}
        self._ws: Dict[int, torch.Tensor] = {}   # condition-embedding workspace, one per stream (launches on different
        #                                          streams may overlap, each needs its own)

        cfg = _lib.ModelCfg()
        cfg.num_coords, cfg.n_joints, cfg.t_unet, cfg.t_cond = num_coords, n_joints, self.t_unet, self.t_cond
        cfg.emb_dim, cfg.strategy = emb_dim, _lib.STRATEGY[strategy]
        if strategy == "inject" and cond_unet:      # 'E_unet' condition encoder (the U-Net's down path)
            cfg.cond_layers = _lib.COND_UNET
        else:
            cfg.cond_layers = len(cond_channels) if strategy == "inject" else 0
            for i, c in enumerate(cond_channels):
                cfg.cond_channels[i] = int(c)
        keep = []  # keep host copies alive during the call
        arr = (_lib.Tensor * len(state_dict))()
        n = 0
        for k, v in state_dict.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            h = v.detach().to("cpu", torch.float32).contiguous()
            keep.append(h)
            arr[n].name = k.encode()
            arr[n].data = h.data_ptr()
            arr[n].numel = h.numel()
            n += 1
        handle = C.c_void_p()
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        with torch.cuda.device(self.device):
            _lib.check(self.L.mcd_pack_weights(arr, n, C.byref(cfg), idx, C.byref(handle)))
        self._h = handle
        for name, value in (options or {}).items():
            self.set_option(name, value)

    def set_option(self, name: str, value: int) -> None:
        """Per-handle switch of the library (include/mocodad_hip.h MCD_OPT_*): 'variant', 'cond_generic',
        'generic_unet', 'split', 'phase'."""
        if name not in _lib.OPT:
            raise ValueError(f"unknown option {name!r} (known: {sorted(_lib.OPT)})")
        _lib.check(self.L.mcd_set_option(self._h, _lib.OPT[name], int(value)))

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                self.L.mcd_free_weights(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------ helpers
    def table(self, noise_steps: int) -> torch.Tensor:
        t = self._tables.get(noise_steps)
        if t is None:
            t = step_table(noise_steps, self.emb_dim).to(self.device)
            self._tables[noise_steps] = t
        return t

    def _score_cfg(self, B: int, S: int, ns: int, loss_fn: str) -> "_lib.ScoreCfg":
        c = _lib.ScoreCfg()
        c.n_windows, c.n_samples, c.noise_steps, c.seg_len = B, S, ns, self.seg_len
        c.n_cond, c.n_corrupt = len(self.cond_idx), len(self.corrupt_idx)
        for i, v in enumerate(self.cond_idx):
            c.cond_idx[i] = v
        for i, v in enumerate(self.corrupt_idx):
            c.corrupt_idx[i] = v
        c.loss_fn = _lib.LOSS[loss_fn]
        return c

    def plan_split(self, n_windows: int, n_samples: int, noise_steps: int) -> int:
        """How a scoring call of this size is cut into workgroups (mcd_plan_split): 1 = ONE launch (a workgroup runs all samples
        of its windows, condition encoder and aggregation inside), n_samples = one trajectory per workgroup + the encoder and
        the aggregation as their own launches, 0 = not on score_kernel (13 .. 32 U-Net frames: the slab-tiled kernel; or 'generic_unet')."""
        cfg = self._score_cfg(int(n_windows), int(n_samples), int(noise_steps), "smooth_l1")
        with torch.cuda.device(self.device):
            r = int(self.L.mcd_plan_split(self._h, C.byref(cfg)))
        if r < 0:
            _lib.check(r)
        return r

    # ------------------------------------------------------------------ entry points
    def _check_shape(self, what: str, t: torch.Tensor, tail: Tuple[int, ...]) -> None:
        if t.dim() != len(tail) + 1 or tuple(t.shape[1:]) != tuple(tail):
            raise ValueError(f"{what} must have shape (B, {', '.join(map(str, tail))}), got {tuple(t.shape)}")

    def cond_encode(self, cond_data: torch.Tensor) -> torch.Tensor:
        if self.strategy != "inject":
            raise ValueError("this model has no condition encoder")
        self._check_shape("cond_data", cond_data, (self.num_coords, self.t_cond, self.n_joints))
        x = _f32c(cond_data, self.device)
        out = torch.empty(x.shape[0], self.emb_dim, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mcd_cond_encode(self._h, _ptr(x), x.shape[0], _ptr(out), _stream()))
        return out

    def unet_forward(self, x: torch.Tensor, t: int, cond: Optional[torch.Tensor], noise_steps: Optional[int] = None) -> torch.Tensor:
        self._check_shape("x", x, (self.num_coords, self.t_unet, self.n_joints))
        if cond is not None:
            self._check_shape("cond", cond, (self.emb_dim,))
            if cond.shape[0] != x.shape[0]:
                raise ValueError(f"cond has {cond.shape[0]} rows, x has {x.shape[0]} windows")
        x = _f32c(x, self.device)
        cond = None if cond is None else _f32c(cond, self.device)
        tab = self.table(noise_steps if noise_steps is not None else max(int(t) + 1, 2))
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            ws = self._pass_workspace(x.shape[0])
            _lib.check(self.L.mcd_unet_forward(self._h, _ptr(x), _ptr(cond), _ptr(tab), int(t), x.shape[0], _ptr(out), _ptr(ws), _stream()))
        return out

    def _pass_workspace(self, n_windows: int) -> Optional[torch.Tensor]:
        """Scratch of the single-pass entries (mcd_pass_workspace_bytes): the slab-tiled kernel's activation slabs; None for
        1 .. 12 U-Net frames.  Call inside `torch.cuda.device(self.device)`."""
        nbytes = int(self.L.mcd_pass_workspace_bytes(self._h, int(n_windows)))
        return torch.empty(nbytes, device=self.device, dtype=torch.uint8) if nbytes > 0 else None

    def score(self, data, *, n_samples: int, noise_steps: int, noise: Optional[torch.Tensor] = None,
              seed: int = 0, first_window_id: int = 0, loss_fn: str = "smooth_l1", want_poses: bool = False,
              cond_mask: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """data (B,C,T,V) tensor, or a mocodad_amd.data.windows.WindowBatch (windows read in place from trajectory
        buffers, test-time transform applied on load) -> (loss (B,S), poses (B,S,C,Tx,V) | None).
        cond_mask (random_imp only): (B,) int32, bit t set = frame t of the window conditions.
        Asynchronous on the current stream."""
        _, loss, poses = self._score(data, n_samples, noise_steps, noise, seed, first_window_id, loss_fn, want_poses, cond_mask,
                                     aggregation=None, want_all=True, out=None)
        return loss, poses

    def score_fused(self, data, *, n_samples: int, noise_steps: int, aggregation: str = "best", noise: Optional[torch.Tensor] = None,
                    seed: int = 0, first_window_id: int = 0, loss_fn: str = "smooth_l1", want_all: bool = False, want_poses: bool = False,
                    cond_mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
        """`score` + the loss-based aggregation over the samples ('best' | 'worst' | 'mean' | 'median' | 'quantile:q') in one
        call -- one kernel launch whenever the workgroups own whole windows (mcd_score_fused).
        -> (aggregated loss (B,), loss (B,S) | None, poses | None).  out: optional preallocated (B,) result."""
        return self._score(data, n_samples, noise_steps, noise, seed, first_window_id, loss_fn, want_poses, cond_mask,
                           aggregation=aggregation, want_all=want_all, out=out)

    def _score(self, data, n_samples, noise_steps, noise, seed, first_window_id, loss_fn, want_poses, cond_mask, *, aggregation,
               want_all, out):
        view = None
        keep = None
        if (cond_mask is not None) != (self.strategy == "random_imp"):
            raise ValueError("cond_mask is required by, and only valid for, the random_imp strategy")
        if cond_mask is not None:
            cond_mask = cond_mask.to(self.device, torch.int32).contiguous()
        if hasattr(data, "as_view"):
            wb = data.to(self.device)
            keep = wb
            view = _lib.WindowView(base=wb.base.data_ptr(), stride_c=wb.stride_c, stride_t=wb.stride_t,
                                   trans=wb.trans.data_ptr() if wb.trans is not None else None,
                                   affine=wb.affine.data_ptr() if wb.affine is not None else None,
                                   cond_mask=cond_mask.data_ptr() if cond_mask is not None else None)
            B = int(wb.base.shape[0])
            data = wb.buffer
            if wb.seg_len != self.seg_len:
                raise ValueError(f"window view has seg_len {wb.seg_len}, model expects {self.seg_len}")
        else:
            self._check_shape("data", data, (self.num_coords, self.seg_len, self.n_joints))
            data = _f32c(data, self.device)
            B = data.shape[0]
            if cond_mask is not None:     # dense windows + per-window condition sets
                view = _lib.WindowView(base=None, stride_c=0, stride_t=0, trans=None, affine=None, cond_mask=cond_mask.data_ptr())
        if cond_mask is not None and cond_mask.numel() != B:
            raise ValueError(f"cond_mask must have {B} entries")
        S = int(n_samples)
        Tx = len(self.corrupt_idx)
        cfg = self._score_cfg(B, S, int(noise_steps), loss_fn)
        loss = torch.empty(B, S, device=self.device, dtype=torch.float32) if want_all else None
        poses = torch.empty(B, S, self.num_coords, Tx, self.n_joints, device=self.device, dtype=torch.float32) if want_poses else None
        if noise is not None:
            noise = _f32c(noise, self.device)
            exp = (S, max(noise_steps - 1, 1), B, self.num_coords, Tx, self.n_joints)
            if tuple(noise.shape) != exp:
                raise ValueError(f"noise must have shape {exp}, got {tuple(noise.shape)}")
        agg, q, name = None, 0.0, None
        if aggregation is not None:
            name = aggregation
            if "quantile" in aggregation:
                q, name = _quantile_of(aggregation), "quantile"
            if name not in ("best", "worst", "mean", "median", "quantile"):
                raise ValueError(f"score_fused aggregates losses (best, worst, mean, median, quantile:q), not {aggregation!r}")
            if out is None:
                agg = torch.empty(B, device=self.device, dtype=torch.float32)
            elif out.shape != (B,) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
                raise ValueError("out must be a contiguous float32 (B,) tensor on the scorer's device")
            else:
                agg = out
        need = int(self.L.mcd_score_workspace_bytes(self._h, C.byref(cfg)))
        with torch.cuda.device(self.device):
            sid = torch.cuda.current_stream().cuda_stream
            ws = self._ws.get(sid)
            if ws is None or ws.numel() < need:
                ws = self._ws[sid] = torch.empty(max(need, 256), device=self.device, dtype=torch.uint8)
            common = (self._h, C.byref(cfg), _ptr(data), C.byref(view) if view is not None else None, _ptr(noise),
                      C.c_uint64(seed & (2**64 - 1)), C.c_int64(first_window_id), _ptr(self.table(noise_steps)), _ptr(ws))
            if aggregation is None:
                _lib.check(self.L.mcd_score_view(*common, _ptr(loss), _ptr(poses), _stream()))
            else:
                _lib.check(self.L.mcd_score_fused(*common, _lib.AGGR[name], C.c_float(q), _ptr(agg), _ptr(loss), _ptr(poses), _stream()))
        del keep
        return agg, loss, poses

    # stage ids of mcd_layer_forward: (Cin, Vin, Cout, Vout)
    _STAGES = {0: (2, 17, 16, 17), 1: (16, 17, 32, 17), 2: (32, 17, 32, 17), 3: (32, 12, 64, 12), 4: (64, 12, 64, 12),
               5: (64, 10, 128, 10), 6: (128, 10, 64, 10), 7: (64, 12, 64, 12), 8: (64, 12, 32, 12), 9: (32, 17, 32, 17),
               10: (32, 17, 2, 17), 11: (32, 17, 32, 12), 12: (64, 12, 64, 10), 13: (64, 10, 64, 12), 14: (32, 12, 32, 17)}

    # the fused (joint resampler + layer) stages of the slab-tiled kernel (13 .. 32 U-Net frames): the stage's input is the
    # RESAMPLER's input (Cin, Vin); the skip tensor of stages 7 / 9 has the layer's own (Cin, V)
    _FUSED_IN = {3: (32, 17), 5: (64, 12), 7: (64, 10), 9: (32, 12)}

    def layer_forward(self, stage: int, x: torch.Tensor, emb: torch.Tensor, skip: Optional[torch.Tensor] = None) -> torch.Tensor:
        """TEST ENTRY: one U-Net stage alone (0..10 ST-GCN layers, 11..14 down1/down2/up3/up2), x (B,Cin,T,Vin),
        emb (B,emb_dim) -> (B,Cout,T,Vout).  13 .. 32 U-Net frames (slab-tiled kernel): stages 3, 5, 7, 9 are joint resampler +
        layer (x = the resampler's input; `skip` = d2 / d1 for stages 7 / 9), stages 11..14 do not exist on their own."""
        cin, vin, cout, vout = self._STAGES[int(stage)]
        tiled = self.t_unet > 12
        if tiled and int(stage) in self._FUSED_IN:
            lcin, lvin = cin, vin
            cin, vin = self._FUSED_IN[int(stage)]
            if skip is not None:
                self._check_shape("skip", skip, (lcin, self.t_unet, lvin))
                skip = _f32c(skip, self.device)
        elif skip is not None:
            raise ValueError("skip: only the fused stages 7 and 9 of the slab-tiled kernel (13 .. 32 U-Net frames) take one")
        self._check_shape("x", x, (cin, self.t_unet, vin))
        self._check_shape("emb", emb, (self.emb_dim,))
        x, emb = _f32c(x, self.device), _f32c(emb, self.device)
        out = torch.empty(x.shape[0], cout, self.t_unet, vout, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            ws = self._pass_workspace(x.shape[0])
            _lib.check(self.L.mcd_layer_forward(self._h, int(stage), _ptr(x), _ptr(skip), _ptr(emb), x.shape[0], _ptr(out), _ptr(ws), _stream()))
        return out

    def philox_noise(self, n_windows: int, *, n_samples: int, noise_steps: int, seed: int = 0, first_window_id: int = 0) -> torch.Tensor:
        """The noise tensor (S, max(ns-1,1), B, C, Tx, V) the perf mode of `score` draws in-kernel for these keys."""
        S, K, Tx = int(n_samples), max(int(noise_steps) - 1, 1), len(self.corrupt_idx)
        out = torch.empty(S, K, int(n_windows), self.num_coords, Tx, self.n_joints, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mcd_philox_noise(C.c_uint64(seed & (2**64 - 1)), C.c_int64(first_window_id), int(n_windows), S,
                                               int(noise_steps), Tx, _ptr(out), _stream()))
        return out

    def aggregate(self, data: torch.Tensor, loss_all: torch.Tensor, poses_all: Optional[torch.Tensor], strategy: str,
                  *, noise_steps: int, loss_fn: str = "smooth_l1", want_pose: bool = True,
                  out: Optional[torch.Tensor] = None) -> Tuple[Optional[torch.Tensor], torch.Tensor]:
        """_aggregation_strategy of the reference on device -> (selected pose | None, loss (B,)).
        out: optional preallocated contiguous fp32 (B,) device tensor receiving the loss."""
        B, S = loss_all.shape
        # the C ABI takes dense (B,S) / (B,S,C,Tx,V) tensors: views with other strides (e.g. a transposed (S,B) stack) are copied
        loss_all = _f32c(loss_all, self.device)
        if poses_all is not None:
            poses_all = _f32c(poses_all, self.device)
        q = 0.0
        name = strategy
        if "quantile" in strategy:
            q = _quantile_of(strategy)
            name = "quantile"
        if name not in _lib.AGGR or name == "all":
            raise ValueError(f"Unknown aggregation strategy {strategy}")
        cfg = self._score_cfg(B, S, int(noise_steps), loss_fn)
        needs_data = name in ("mean_pose", "median_pose")
        if torch.is_tensor(data):
            data = _f32c(data, self.device)
        elif needs_data:
            raise ValueError("the *_pose aggregation strategies need the materialised (B,C,T,V) windows")
        else:
            data = None
        if out is None:
            out = torch.empty(B, device=self.device, dtype=torch.float32)
        elif out.shape != (B,) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != self.device:
            raise ValueError("out must be a contiguous float32 (B,) tensor on the scorer's device")
        gives_pose = name in ("best", "worst", "mean_pose", "median_pose")
        pose = None
        if gives_pose and want_pose and poses_all is not None:
            pose = torch.empty(poses_all.shape[0], *poses_all.shape[2:], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mcd_aggregate(C.byref(cfg), self.num_coords, self.n_joints, _lib.AGGR[name], C.c_float(q),
                                            _ptr(loss_all), _ptr(poses_all), _ptr(data), _ptr(out), _ptr(pose), _stream()))
        return pose, out

    def scatter_max(self, scores: torch.Tensor, frames: torch.Tensor, row: torch.Tensor, n_rows: int, n_frames: int) -> torch.Tensor:
        scores = _f32c(scores, self.device)
        frames = frames.to(self.device, torch.int32).contiguous()
        row = row.to(self.device, torch.int32).contiguous()
        out = torch.empty(n_rows, n_frames, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            _lib.check(self.L.mcd_scatter_max(_ptr(scores), _ptr(frames), _ptr(row), scores.numel(), frames.shape[1],
                                              n_rows, n_frames, _ptr(out), _stream()))
        return out


class FrameScoreAssembler:
    """Window scores -> per-frame anomaly scores on the device (mcd_frame_scores): the whole of the reference's
    post_processing loops (mocodad.py:362-425) except roc_auc_score.  Built once per dataset from the ground-truth masks."""

    MAX_WORKSPACE = 8 << 30

    def __init__(self, gts, masks, *, num_transform: int, pad_size: int, filter_kernel_size: float, frames_shift: int, device):
        from .utils.eval_utils import frame_tables, gaussian_kernel1d
        self.L = _lib.lib()
        self.device = torch.device(device)
        t = frame_tables(gts, masks)
        self.gt, self.total, self.F = t["gt"], int(t["total"]), int(t["max_frames"])
        w = gaussian_kernel1d(filter_kernel_size)
        self._dev = {k: torch.from_numpy(np.ascontiguousarray(t[k])).to(self.device)
                     for k in ("clip_keys", "clip_n_frames", "frame_dst", "clip_out_len", "clip_out_off")}
        self._dev["gauss"] = torch.from_numpy(w).to(self.device)
        self.n_clips, self.num_transform = len(t["clip_keys"]), int(num_transform)
        self.pad_size, self.frames_shift, self.radius = int(pad_size), int(frames_shift), (len(w) - 1) // 2
        self._ws = None

    def _cfg(self, n_persons: int) -> "_lib.FrameCfg":
        d = self._dev
        return _lib.FrameCfg(n_clips=self.n_clips, num_transform=self.num_transform, n_persons=n_persons, max_frames=self.F,
                             pad_size=self.pad_size, frames_shift=self.frames_shift, gauss_radius=self.radius,
                             clip_keys=d["clip_keys"].data_ptr(), clip_n_frames=d["clip_n_frames"].data_ptr(),
                             frame_dst=d["frame_dst"].data_ptr(), clip_out_len=d["clip_out_len"].data_ptr(),
                             clip_out_off=d["clip_out_off"].data_ptr(), gauss_weights=d["gauss"].data_ptr())

    def __call__(self, scores, trans, meta, frames) -> Optional[np.ndarray]:
        """scores (N,), trans (N,), meta (N,4), frames (N,seg_len): arrays or tensors, host or device -> pds (total,) float64
        (None if the dense (transform, clip, person) table would not fit the workspace cap: the caller falls back to the host)."""
        dev = self.device
        # Host arrays (what processing_data hands over) are checked on the host: the same checks as torch device ops cost a lazy
        # code-object load per op on first use (isfinite, >=, &, all, max: ~80 ms of a cold epoch end, tools/postproc_time.py)
        ok = npers = None
        if not torch.is_tensor(scores) and not torch.is_tensor(meta):
            sc_h, me_h = np.asarray(scores), np.asarray(meta)
            if sc_h.size:
                ok = bool((np.isfinite(sc_h) & (sc_h >= 0)).all())
                npers = int(me_h[:, 2].max()) + 1 if me_h.ndim == 2 and me_h.shape[1] == 4 else None
        as_t = lambda a, dt: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, dt).contiguous()
        scores, trans, meta, frames = as_t(scores, torch.float32), as_t(trans, torch.int64), as_t(meta, torch.int64), as_t(frames, torch.int32)
        n = int(scores.numel())
        if meta.shape != (n, 4) or trans.numel() != n or frames.shape[0] != n:
            raise ValueError("scores / trans / meta / frames disagree on the number of windows")
        # the scatter-max orders non-negative floats by their bit patterns: a NaN / negative / infinite window score (a
        # diverged model) must surface as an error here, not as a plausible-looking AUC (the reference's NumPy path lets the
        # NaN reach roc_auc_score, which raises)
        if ok is None:
            ok = (not n) or bool((torch.isfinite(scores) & (scores >= 0)).all().item())
        if not ok:
            raise ValueError("window scores must be finite and non-negative (got NaN / inf / negative values: diverged model?)")
        n_persons = (npers if npers is not None else int(meta[:, 2].max().item()) + 1) if n else 1
        cfg = self._cfg(max(n_persons, 1))
        need = int(self.L.mcd_frame_scores_workspace_bytes(C.byref(cfg)))
        if need > self.MAX_WORKSPACE:
            return None
        with torch.cuda.device(dev):
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, device=dev, dtype=torch.uint8)
            out = torch.empty(self.total, device=dev, dtype=torch.float64)
            _lib.check(self.L.mcd_frame_scores(C.byref(cfg), _ptr(scores), _ptr(trans), _ptr(meta), _ptr(frames), n,
                                               int(frames.shape[1]) if n else 1, _ptr(self._ws), _ptr(out), _stream()))
        pds = out.cpu().numpy()
        if np.isnan(pds).any():
            raise ValueError("a (transform, clip) block has no pose windows (need at least one array to stack)")
        return pds
