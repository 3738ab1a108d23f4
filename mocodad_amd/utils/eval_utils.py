"""Window scores -> per-frame clip scores (the step right after the hot path).

Reference: models/mocodad.py:362-425 and utils/eval_utils.py (compute_var_matrix :27-34, score_process
:100-106, pad_scores :133-149, get_avenue_mask :152-166, get_hr_ubnormal_mask :169-185).  The reference walks
(transform, clip, person) with three nested Python loops and an O(N) boolean filter at each level; here all
windows are grouped once and scattered with a single scatter-max (on the GPU through the C ABI's
mcd_scatter_max when a scorer is available, np.maximum.at otherwise)."""
import os
from glob import glob
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
from scipy.ndimage import gaussian_filter1d

# HR-Avenue: frames kept (1) / dropped (0) per test clip, run-length encoded (value, count)
_AVENUE_RLE = {
    1: [(1, 75), (0, 46), (1, 269), (0, 47), (1, 427), (0, 47), (1, 20), (0, 70), (1, 438)],
    2: [(1, 272), (0, 48), (1, 403), (0, 41), (1, 447)],
    3: [(1, 293), (0, 48), (1, 582)],
    6: [(1, 561), (0, 64), (1, 189), (0, 193), (1, 276)],
    16: [(1, 728), (0, 12)],
}


def get_avenue_mask() -> Dict[int, List[int]]:
    return {k: [v for v, n in rle for _ in range(n)] for k, rle in _AVENUE_RLE.items()}


def get_hr_ubnormal_mask(split: str) -> Dict[Tuple[int, int], np.ndarray]:
    sub = "testing" if "test" in split else "validating"
    out = {}
    for path in glob(f"./data/UBnormal/hr_bool_masks/{sub}/test_frame_mask/*"):
        scene, clip = map(int, os.path.basename(path).split(".")[0].split("_"))
        out[(scene, clip)] = np.load(path)
    return out


def compute_var_matrix(pos: np.ndarray, frames_pos: np.ndarray, n_frames: int) -> np.ndarray:
    """(w,) window scores + (w, seg_len) 1-based frame ids -> (w, n_frames), zero where a window is absent."""
    mat = np.zeros((pos.shape[0], n_frames))
    if pos.shape[0]:
        rows = np.repeat(np.arange(pos.shape[0]), frames_pos.shape[1])
        mat[rows, frames_pos.reshape(-1) - 1] = np.repeat(pos, frames_pos.shape[1])
    return mat


def score_process(score: np.ndarray, shift: int, kernel_size: float) -> np.ndarray:
    shifted = np.zeros_like(score)
    shifted[shift:] = score[:-shift]
    return gaussian_filter1d(shifted, kernel_size)


def pad_scores(score: np.ndarray, gt: np.ndarray, pad_size: int) -> np.ndarray:
    """Zero `pad_size` frames around every interval (within the first len(gt)-1 frames) where the person is absent."""
    n = len(gt)
    absent = np.zeros(n + 1, dtype=bool)
    absent[1:n] = score[:n - 1] == 0          # 1-shifted so that diff() marks run starts/ends
    edges = np.flatnonzero(np.diff(absent.astype(np.int8)))
    zero_ranges = []
    for start, stop in zip(edges[0::2], edges[1::2]):   # run covers frames start .. stop-1
        end = stop - 1
        if start == 0 and end == n - 2:
            continue
        lo = start if start == 0 else max(start - pad_size, 0)
        hi = end if end == n - 2 else min(end + pad_size, n)
        zero_ranges.append((lo, hi))
    for lo, hi in zero_ranges:
        score[lo:hi] = 0
    return score


def frame_score_rows(out, trans, meta, frames, gts, num_transform, scatter_max: Optional[Callable] = None):
    """Scatter-max every window score onto its frames.  Returns (mat (n_rows, n_frames_max), row_keys (n_rows, 4))
    with one row per (transform, scene, clip, person) in lexicographic order."""
    clip_ok = np.zeros(len(out), dtype=bool)
    for (sc, cl) in gts:
        clip_ok |= (meta[:, 0] == sc) & (meta[:, 1] == cl)
    sel = clip_ok & (trans >= 0) & (trans < num_transform)
    keys = np.stack([trans[sel], meta[sel, 0], meta[sel, 1], meta[sel, 2]], axis=1).astype(np.int64)
    row_keys, row = np.unique(keys, axis=0, return_inverse=True)
    row = row.reshape(-1)
    n_frames = max(len(g) for g in gts.values())
    o, f = np.asarray(out)[sel], np.asarray(frames)[sel]
    if scatter_max is not None and len(o):
        import torch
        mat = scatter_max(torch.from_numpy(np.ascontiguousarray(o, dtype=np.float32)), torch.from_numpy(np.ascontiguousarray(f)),
                          torch.from_numpy(row.astype(np.int32)), len(row_keys), n_frames).cpu().numpy().astype(np.float64)
    else:
        mat = np.zeros((len(row_keys), n_frames))
        rr = np.repeat(row, f.shape[1])
        ff = f.reshape(-1) - 1
        ok = (ff >= 0) & (ff < n_frames)
        np.maximum.at(mat, (rr[ok], ff[ok]), np.repeat(o, f.shape[1])[ok])
    return mat, row_keys


def post_process_scores(out, trans, meta, frames, gts: Dict[Tuple[int, int], np.ndarray], *, num_transform: int,
                        pad_size: int, filter_kernel_size: float, frames_shift: int, masks: Optional[Dict] = None,
                        scatter_max: Optional[Callable] = None) -> Tuple[np.ndarray, np.ndarray]:
    """-> (per-frame anomaly scores averaged over transforms, ground truth), both concatenated over clips in
    sorted file-name order, as fed to roc_auc_score at mocodad.py:424-428."""
    masks = masks or {}
    clip_keys = sorted(gts.keys(), key=lambda k: f"{k[0]:02d}_{k[1]:04d}")
    mat, row_keys = frame_score_rows(np.asarray(out), np.asarray(trans), np.asarray(meta), np.asarray(frames), gts,
                                     num_transform, scatter_max)
    per_transform, gt_cat = [], None
    for tr in range(num_transform):
        scores, gcat = [], []
        for (sc, cl) in clip_keys:
            gt = gts[(sc, cl)]
            n = gt.shape[0]
            rsel = np.flatnonzero((row_keys[:, 0] == tr) & (row_keys[:, 1] == sc) & (row_keys[:, 2] == cl))
            if len(rsel) == 0:
                raise ValueError(f"no pose windows for transform {tr}, clip {sc:02d}_{cl:04d} (need at least one array to stack)")
            persons = mat[rsel, :n].copy()
            if pad_size != -1:
                for p in range(persons.shape[0]):
                    pad_scores(persons[p], gt, pad_size)
            lg = np.log1p(persons)
            cs = persons.mean(0) + (lg.max(0) - lg.min(0))
            g = gt
            if (sc, cl) in masks:
                keep = np.asarray(masks[(sc, cl)]).astype(bool)
                cs, g = cs[keep], g[keep]
            if ("clip", cl) in masks:
                keep = np.asarray(masks[("clip", cl)]) == 1
                cs, g = cs[keep], g[keep]
            scores.append(score_process(cs, frames_shift, filter_kernel_size))
            gcat.append(g)
        per_transform.append(np.concatenate(scores))
        if gt_cat is None:
            gt_cat = np.concatenate(gcat)
    return np.mean(np.stack(per_transform, 0), 0), gt_cat


# ------------------------------------------------------------------ device path (mcd_frame_scores)
def gaussian_kernel1d(sigma: float, truncate: float = 4.0) -> np.ndarray:
    """The weights scipy.ndimage.gaussian_filter1d(x, sigma) correlates with (order 0): radius int(truncate*sigma + 0.5)."""
    sd = float(sigma)
    radius = int(truncate * sd + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sd * sd) * x ** 2)
    return phi / phi.sum()


def frame_tables(gts: Dict[Tuple[int, int], np.ndarray], masks: Optional[Dict] = None):
    """Per-dataset tables of the device frame-score assembly: clips in ascending (scene, clip) key order (what the kernel
    binary-searches), their output offsets in sorted file-name order (the order the reference concatenates in), the
    position of every frame after the HR masks, and the concatenated ground truth."""
    masks = masks or {}
    by_name = sorted(gts.keys(), key=lambda k: f"{k[0]:02d}_{k[1]:04d}")
    slots = sorted(gts.keys(), key=lambda k: (int(k[0]) << 32) | (int(k[1]) & 0xffffffff))
    F = max(len(g) for g in gts.values())
    kept, gt_kept = {}, {}
    for (sc, cl) in by_name:
        g = np.asarray(gts[(sc, cl)])
        idx = np.arange(len(g))
        if (sc, cl) in masks:
            keep = np.asarray(masks[(sc, cl)]).astype(bool)
            idx, g = idx[keep], g[keep]
        if ("clip", cl) in masks:
            keep = np.asarray(masks[("clip", cl)]) == 1
            idx, g = idx[keep], g[keep]
        kept[(sc, cl)], gt_kept[(sc, cl)] = idx, g
    off, o = {}, 0
    for k in by_name:
        off[k] = o
        o += len(kept[k])
    dst = np.full((len(slots), F), -1, dtype=np.int32)
    for i, k in enumerate(slots):
        dst[i, kept[k]] = np.arange(len(kept[k]), dtype=np.int32)
    return {
        "clip_keys": np.array([(int(k[0]) << 32) | (int(k[1]) & 0xffffffff) for k in slots], dtype=np.int64),
        "clip_n_frames": np.array([len(gts[k]) for k in slots], dtype=np.int32),
        "frame_dst": dst,
        "clip_out_len": np.array([len(kept[k]) for k in slots], dtype=np.int32),
        "clip_out_off": np.array([off[k] for k in slots], dtype=np.int64),
        "total": o, "max_frames": F,
        "gt": np.concatenate([gt_kept[k] for k in by_name]) if by_name else np.zeros(0),
    }
