"""Windows as views into per-person trajectories.

The reference materialises every stride-1 window (x seg_len) and every test-time transform (x num_transform) on the
host before the model sees anything (utils/preprocessing.py:14-86, utils/dataset.py:67-76).  Here a dataset is one flat
buffer of trajectories [(frame, coord, joint), ...] plus, per window, an element offset and a transform index; the HIP
kernels read a window in place and apply the affine transform while loading (include/mocodad_hip.h: mcd_window_view_t).
Only the trajectories cross PCIe: 1/(seg_len * num_transform) of the bytes of the materialised windows."""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from ..utils.transforms import affine_table

N_JOINTS, N_COORDS = 17, 2


@dataclass
class WindowBatch:
    """A batch of window views.  buffer: flat fp32 trajectories, frame-major (F_total, C, V)."""
    buffer: torch.Tensor
    base: torch.Tensor                 # (B,) int64 element offset of each window's first frame
    trans: Optional[torch.Tensor]      # (B,) int32 transform index, or None
    affine: Optional[torch.Tensor]     # (num_transform, 6) fp32
    seg_len: int
    stride_c: int = N_JOINTS
    stride_t: int = N_COORDS * N_JOINTS

    def as_view(self):
        return self

    @property
    def shape(self):
        return (int(self.base.shape[0]), N_COORDS, self.seg_len, N_JOINTS)

    def to(self, device):
        if self.buffer.device == torch.device(device) and self.base.device == torch.device(device):
            return self
        mv = lambda t: None if t is None else t.to(device, non_blocking=True)
        return WindowBatch(mv(self.buffer), mv(self.base), mv(self.trans), mv(self.affine), self.seg_len, self.stride_c, self.stride_t)

    def materialize(self) -> torch.Tensor:
        """The (B,C,T,V) tensor the reference's DataLoader would have produced (host-side check / fallback)."""
        dev = self.base.device
        t = torch.arange(self.seg_len, device=dev)[None, None, :, None] * self.stride_t
        c = torch.arange(N_COORDS, device=dev)[None, :, None, None] * self.stride_c
        v = torch.arange(N_JOINTS, device=dev)[None, None, None, :]
        w = self.buffer.to(dev)[self.base[:, None, None, None] + t + c + v]
        if self.trans is not None:
            a = self.affine.to(dev)[self.trans.long()].reshape(-1, 2, 3)
            x, y = w[:, 0], w[:, 1]
            w = torch.stack([(a[:, 0, 0, None, None] * x + a[:, 0, 1, None, None] * y) + a[:, 0, 2, None, None],
                             (a[:, 1, 0, None, None] * x + a[:, 1, 1, None, None] * y) + a[:, 1, 2, None, None]], 1)
        return w


class TrajectoryWindows:
    """All stride-`seg_stride` windows x transforms of a set of person trajectories, in the reference's dataset order
    (transform-major: index = trans * n_samples + sample, utils/dataset.py:67-71)."""

    def __init__(self, trajectories: Dict[Tuple[int, int, int], Tuple[int, np.ndarray]], seg_len: int,
                 num_transform: int = 1, seg_stride: int = 1):
        self.seg_len, self.num_transform = seg_len, max(1, num_transform)
        bufs, base, meta, frames = [], [], [], []
        off = 0
        for (scene, clip, person), (first_frame, traj) in sorted(trajectories.items()):
            traj = np.ascontiguousarray(traj, dtype=np.float32)            # (F, C, V)
            assert traj.shape[1:] == (N_COORDS, N_JOINTS), traj.shape
            F = traj.shape[0]
            for s in range(0, F - seg_len + 1, seg_stride):
                base.append(off + s * N_COORDS * N_JOINTS)
                meta.append((scene, clip, person, first_frame + s))
                frames.append(np.arange(first_frame + s, first_frame + s + seg_len))
            bufs.append(traj.reshape(-1))
            off += traj.size
        self.buffer = torch.from_numpy(np.concatenate(bufs)) if bufs else torch.zeros(0)
        n = len(base)
        self.n_samples = n
        nt = self.num_transform
        self.base = torch.tensor(base, dtype=torch.int64).repeat(nt)
        self.trans = torch.arange(nt, dtype=torch.int32).repeat_interleave(n)
        self.meta = torch.tensor(meta, dtype=torch.int64).reshape(-1, 4).repeat(nt, 1)
        self.frames = torch.from_numpy(np.stack(frames).astype(np.int32)).repeat(nt, 1) if n else torch.zeros(0, seg_len, dtype=torch.int32)
        self.affine = affine_table(nt)

    def __len__(self):
        return int(self.base.shape[0])

    def to(self, device):
        """Upload the trajectories once (the only bulk H2D copy of an evaluation)."""
        self.buffer = self.buffer.to(device)
        self.affine = self.affine.to(device)
        return self

    def batch(self, lo: int, hi: int) -> List:
        """[WindowBatch, transformation_idx, metadata, actual_frames] — the 4-list MoCoDAD.forward / test_step take."""
        wb = WindowBatch(self.buffer, self.base[lo:hi], self.trans[lo:hi], self.affine, self.seg_len)
        return [wb, self.trans[lo:hi].long(), self.meta[lo:hi], self.frames[lo:hi]]

    def batches(self, batch_size: int, lo: int = 0, hi: Optional[int] = None):
        hi = len(self) if hi is None else hi
        return [self.batch(i, min(i + batch_size, hi)) for i in range(lo, hi, batch_size)]

    def materialize(self) -> torch.Tensor:
        return WindowBatch(self.buffer, self.base, self.trans, self.affine, self.seg_len).materialize()
