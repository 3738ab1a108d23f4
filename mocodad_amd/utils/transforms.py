"""Test-time pose transforms of the reference's dataset (utils/dataset_utils.py:255-310, applied per item in
utils/dataset.py:67-76) as a small affine table, so that the HIP kernels can apply them while loading a window
instead of the host materialising num_transform copies of the dataset."""
import math

import torch

# (sx, sy, tx, ty, rot_degrees, flip) of ae_trans_list, in order
AE_TRANSFORMS = [(1, 1, 0.0, 0.0, 0, False), (1, 1, 0.0, 0.0, 0, True), (1, 1, 0.0, 0.0, 90, False),
                 (1, 1, 0.0, 0.0, 90, True), (1, 1, 0.0, 0.0, 45, False)]


def affine_matrix(sx=1.0, sy=1.0, tx=0.0, ty=0.0, rot=0.0, flip=False) -> torch.Tensor:
    c, s = math.cos(math.radians(rot)), math.sin(math.radians(rot))
    flip_m = torch.diag(torch.tensor([-1.0 if flip else 1.0, 1.0, 1.0]))
    scale_m = torch.tensor([[sx, 0.0, tx], [0.0, sy, ty], [0.0, 0.0, 1.0]], dtype=torch.float32)
    rot_m = torch.tensor([[c, -s, 0.0], [s, c, 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    return flip_m @ (rot_m @ scale_m)


def affine_table(num_transform: int) -> torch.Tensor:
    """(num_transform, 6) fp32 rows [a00 a01 a02 a10 a11 a12]: x' = a00 x + a01 y + a02, y' = a10 x + a11 y + a12."""
    rows = [affine_matrix(*AE_TRANSFORMS[i])[:2].reshape(-1) for i in range(num_transform)]
    return torch.stack(rows).contiguous()
