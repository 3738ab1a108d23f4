"""Synthetic pose-window datasets in the tensor layout the reference's data pipeline produces
(utils/dataset.py:255-275: data (N,2,seg_len,17) f32, trans (N,) i64, meta (N,4) = [scene, clip, person,
first_frame] i64, frames (N,seg_len) 1-based i32) plus per-clip ground-truth frame masks written as
`{scene:02d}_{clip:04d}.npy` (README.md:41-85).  Real datasets are Google-Drive downloads (no network here), so
benchmarks and end-to-end tests run on these."""
import os
from typing import Dict, List, Tuple

import numpy as np
import torch


def make_dataset(n_clips: int = 4, frames_per_clip: int = 120, persons_per_clip: int = 3, seg_len: int = 6,
                 num_transform: int = 5, seed: int = 999, anomaly_gain: float = 3.0):
    """Smooth random-walk trajectories per (clip, person), robust-scaled-like (~N(0,1), clipped to +-5), cut into
    stride-1 windows, replicated for `num_transform` test-time transforms (mirrored / scaled copies).
    Anomalous intervals get jerkier motion so the AUC is not degenerate."""
    rng = np.random.default_rng(seed)
    data, trans, meta, frames = [], [], [], []
    gts: Dict[Tuple[int, int], np.ndarray] = {}
    for clip in range(1, n_clips + 1):
        scene = 1
        gt = np.zeros(frames_per_clip, dtype=np.int64)
        a = int(rng.integers(frames_per_clip // 4, frames_per_clip // 2))
        gt[a:a + frames_per_clip // 5] = 1
        gts[(scene, clip)] = gt
        for person in range(1, persons_per_clip + 1):
            f0 = int(rng.integers(1, 10))
            f1 = frames_per_clip - int(rng.integers(0, 10))
            n = f1 - f0 + 1
            step = rng.standard_normal((n, 2, 17)) * 0.08
            step[gt[f0 - 1:f1] == 1] *= anomaly_gain
            traj = np.cumsum(step, 0) + rng.standard_normal((1, 2, 17))
            traj = np.clip((traj - np.median(traj)) / (np.percentile(traj, 90) - np.percentile(traj, 10) + 1e-6), -5, 5)
            for tr in range(num_transform):
                t = traj.copy()
                if tr % 2 == 1:
                    t[:, 0] = -t[:, 0]
                t = t * (1.0 + 0.1 * tr)
                for s in range(0, n - seg_len + 1):
                    data.append(t[s:s + seg_len].transpose(1, 0, 2))
                    trans.append(tr)
                    meta.append((scene, clip, person, f0 + s))
                    frames.append(np.arange(f0 + s, f0 + s + seg_len))
    return (torch.from_numpy(np.stack(data).astype(np.float32)), torch.tensor(trans, dtype=torch.int64),
            torch.tensor(meta, dtype=torch.int64), torch.from_numpy(np.stack(frames).astype(np.int32)), gts)


def make_trajectories(n_clips: int = 4, frames_per_clip: int = 120, persons_per_clip: int = 3, seed: int = 999,
                      anomaly_gain: float = 3.0):
    """{(scene, clip, person): (first_frame, traj (F,2,17) f32)} + gt masks, same generative model as make_dataset."""
    rng = np.random.default_rng(seed)
    trajs, gts = {}, {}
    for clip in range(1, n_clips + 1):
        scene = 1
        gt = np.zeros(frames_per_clip, dtype=np.int64)
        a = int(rng.integers(frames_per_clip // 4, frames_per_clip // 2))
        gt[a:a + frames_per_clip // 5] = 1
        gts[(scene, clip)] = gt
        for person in range(1, persons_per_clip + 1):
            f0 = int(rng.integers(1, 10))
            f1 = frames_per_clip - int(rng.integers(0, 10))
            n = f1 - f0 + 1
            step = rng.standard_normal((n, 2, 17)) * 0.08
            step[gt[f0 - 1:f1] == 1] *= anomaly_gain
            traj = np.cumsum(step, 0) + rng.standard_normal((1, 2, 17))
            traj = np.clip((traj - np.median(traj)) / (np.percentile(traj, 90) - np.percentile(traj, 10) + 1e-6), -5, 5)
            trajs[(scene, clip, person)] = (f0, traj.astype(np.float32))
    return trajs, gts


def write_gt(gt_dir: str, gts: Dict[Tuple[int, int], np.ndarray]) -> None:
    os.makedirs(gt_dir, exist_ok=True)
    for (scene, clip), g in gts.items():
        np.save(os.path.join(gt_dir, f"{scene:02d}_{clip:04d}.npy"), g)


def batches(tensors, batch_size: int, lo: int = 0, hi: int = None) -> List[List[torch.Tensor]]:
    n = tensors[0].shape[0]
    hi = n if hi is None else hi
    return [[t[i:min(i + batch_size, hi)] for t in tensors[:4]] for i in range(lo, hi, batch_size)]
