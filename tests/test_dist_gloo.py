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


def _mismatch_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = WindowShard(10)
        shard.host_meta = (np.arange(10), np.zeros((10, 4)), np.zeros((10, 6)))
        n = len(shard) - (1 if rank == 1 else 0)          # rank 1 lost a window
        try:
            shard.gather(np.zeros(n, dtype=np.float32), None, None, None)
            q.put((rank, "no error"))
        except ValueError as e:
            q.put((rank, "ValueError"))
    finally:
        dist.destroy_process_group()


def test_shard_length_mismatch_fails_on_every_rank():
    """A rank whose score count does not match its shard must not leave the others blocked in the all-gather: the check is
    collective, every rank raises."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mismatch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == {0: "ValueError", 1: "ValueError"}, res


def _tiny_dataset():
    """One clip, one person, 9 windows: with 4 ranks the shards are 3 + 3 + 3 + 0 (ceil-divided), with 2 ranks 5 + 4."""
    from mocodad_amd.data import synthetic
    data, trans, meta, frames, gts = synthetic.make_dataset(n_clips=1, frames_per_clip=40, persons_per_clip=1, num_transform=1)
    return data[:9], trans[:9], meta[:9], frames[:9], gts


def _epoch_worker(rank, world, port, gt_dir, q):
    """MoCoDAD._epoch_end with a WindowShard: every rank contributes its (possibly EMPTY) shard of window scores, rank 0
    alone post-processes.  The scores are injected (no GPU here); the exchange and the host side are the real ones."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import golden_weights, make_args
        from mocodad_amd.models.mocodad import MoCoDAD
        data, trans, meta, frames, gts = _tiny_dataset()
        n = data.shape[0]
        _, cfg = golden_weights("inject")
        m = MoCoDAD(make_args(cfg, gt_path=gt_dir, num_transform=1, dataset_choice="HR-STC", pad_size=-1, filter_kernel_size=3,
                              frames_shift=2, save_tensors=False))
        scores = torch.linspace(0.1, 2.0, n)
        shard = WindowShard(n)
        shard.host_meta = (trans.numpy(), meta.numpy(), frames.numpy())
        m.shard = shard
        m.on_test_epoch_start()
        for lo in range(shard.lo, shard.hi, 2):
            hi = min(lo + 2, shard.hi)
            m._test_output_list.append([scores[lo:hi], data[lo:hi], trans[lo:hi], meta[lo:hi], frames[lo:hi]])
        auc = m.on_test_epoch_end()
        q.put((rank, float(auc), len(shard)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_epoch_end_rank0_auc(tmp_path, world):
    from mocodad_amd.data import synthetic
    from oracle import mocodad_oracle as O
    data, trans, meta, frames, gts = _tiny_dataset()
    synthetic.write_gt(str(tmp_path), gts)
    n = data.shape[0]
    ref, _, _ = O.post_processing(torch.linspace(0.1, 2.0, n).numpy(), trans.numpy(), meta.numpy(), frames.numpy(), gts,
                                  num_transform=1, pad_size=-1, filter_kernel_size=3, frames_shift=2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_epoch_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
    assert abs(res[0][1] - ref) < 1e-9, (res, ref)
    assert all(np.isnan(r[1]) for r in res[1:])
    # contiguous shards of ceil(n / world) windows, the tail ranks short or EMPTY (8 ranks = the node the path is built for)
    assert [r[2] for r in res] == [hi - lo for lo, hi in (shard_range(n, r, world) for r in range(world))]
    assert [r[2] for r in res] == {2: [5, 4], 4: [3, 3, 3, 0], 8: [2, 2, 2, 2, 1, 0, 0, 0]}[world]
