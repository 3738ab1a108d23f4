"""CPU: the multi-GPU path (window sharding + one all-gather of scores) with world_size 2 on gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mocodad_amd.parallel import WindowShard, shard_range


def test_shard_range_partitions_the_index_range():
    for n in (0, 1, 7, 8, 9, 1000, 1023):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) <= -(-n // world) if n else True


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = WindowShard(n_total)
        full_ref = torch.arange(n_total, dtype=torch.float32) * 0.5 + 1
        local = full_ref[shard.lo:shard.hi].clone()
        full = shard.all_gather_scores(local)
        ok = torch.equal(full, full_ref)
        shard.host_meta = (np.arange(n_total), np.zeros((n_total, 4)), np.zeros((n_total, 6)))
        out, tr, meta, fr = shard.gather(local.numpy(), None, None, None)
        ok = ok and np.array_equal(out, full_ref.numpy()) and len(tr) == n_total
        q.put((rank, bool(ok), shard.lo, shard.hi))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [10, 7, 1])
def test_all_gather_scores_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), res
    spans = sorted((r[2], r[3]) for r in res)
    assert spans[0][0] == 0 and spans[-1][1] == n_total
