"""Window sharding across the GPUs of one node (one process per GPU, torch.distributed; backend 'nccl' = RCCL over
xGMI on MI355X, 'gloo' in the CPU tests).

The (window, sample) chains are independent, the weights (0.58 MB) are replicated, and the in-kernel noise is keyed
by the GLOBAL window id, so scores do not depend on the number of GPUs.  The only exchange is ONE all-gather of the
fp32 window scores after the last batch (reference eval is single-device: eval_MoCoDAD.py:36)."""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of the window index range owned by `rank` (ceil-divided; last shards may be short/empty)."""
    per = -(-n // world) if world > 0 else n
    lo = min(rank * per, n)
    return lo, min(lo + per, n)


class WindowShard:
    def __init__(self, n_total: int, rank: Optional[int] = None, world: Optional[int] = None, group=None):
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.n_total = int(n_total)
        self.per = -(-self.n_total // self.world)
        self.lo, self.hi = shard_range(self.n_total, self.rank, self.world)

    def __len__(self):
        return self.hi - self.lo

    def all_gather_scores(self, local: torch.Tensor) -> torch.Tensor:
        """local (len(self),) fp32 on the collective's device -> (n_total,) on every rank.  Shards are padded to a
        common length so a single all_gather_into_tensor (one RCCL ncclAllGather) does the whole exchange."""
        pad = torch.zeros(self.per, dtype=torch.float32, device=local.device)
        pad[: len(self)] = local.to(torch.float32)
        full = torch.empty(self.per * self.world, dtype=torch.float32, device=local.device)
        if self.world == 1:
            full.copy_(pad)
        else:
            dist.all_gather_into_tensor(full, pad, group=self.group)
        return full[: self.n_total]

    def gather(self, out, trans, meta, frames, device=None):
        """Used by MoCoDAD._epoch_end: the local shard's scores (tensor or array, possibly empty) are exchanged;
        trans/meta/frames are index-deterministic host data that every rank can rebuild, so they are taken from
        `self.host_meta` (the full arrays) set by the driver."""
        dev = device if device is not None and (dist.get_backend(self.group) == "nccl") else "cpu"
        local = out if torch.is_tensor(out) else torch.as_tensor(np.asarray(out))
        # the length check is collective: a rank that raised alone would leave the others blocked in the all-gather
        bad = torch.tensor([int(local.numel() != len(self))], dtype=torch.int32, device=dev)
        if self.world > 1:
            dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=self.group)
        if int(bad.item()):
            raise ValueError(f"rank {self.rank} scored {local.numel()} windows, its shard has {len(self)} "
                             "(or another rank reported a mismatch: every rank fails together)")
        full = self.all_gather_scores(local.reshape(-1).to(device=dev, dtype=torch.float32)).cpu().numpy()
        hm = getattr(self, "host_meta", None)
        if hm is None:
            raise RuntimeError("WindowShard.host_meta = (trans, meta, frames) of the FULL dataset must be set by the driver")
        return full, hm[0], hm[1], hm[2]
