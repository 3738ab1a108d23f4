"""Host-side collation that follows the scoring path (reference: utils/model_utils.py:110-137)."""
from typing import Sequence, Tuple

import numpy as np
import torch


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def processing_data(data: Sequence[Sequence]) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """list of per-batch [output, tensor_data, transformation_idx, metadata, actual_frames] -> 5 arrays.

    Unlike the reference (one blocking .cpu() per tensor per batch), device tensors of one column are
    concatenated on the device first and cross PCIe once."""
    cols = list(zip(*[d[:5] for d in data]))
    out = []
    for ci, col in enumerate(cols):
        if ci == 1 and any(hasattr(t, "as_view") or t is None for t in col):
            out.append(None)     # windows were views into trajectories: nothing to collate (gt_data is unused downstream)
            continue
        if all(torch.is_tensor(t) and t.is_cuda for t in col):
            out.append(torch.cat(list(col), dim=0).cpu().numpy())
        else:
            out.append(np.concatenate([_np(t) for t in col], axis=0))
    return tuple(out)
