#!/usr/bin/env python3
"""eval_MoCoDAD.py — Trainer-free counterpart of the reference's eval_MoCoDAD.py (same `-c config.yaml` interface,
same YAML keys) running the MI355X path.

  python eval_MoCoDAD.py -c config.yaml                       # dataset + checkpoint from the YAML paths
  python eval_MoCoDAD.py -c config.yaml --synthetic 8         # 8 synthetic clips (no dataset files needed)
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 eval_MoCoDAD.py -c cfg.yaml --synthetic 64

With more than one process the window index range is sharded contiguously over the ranks (one GPU each) and the
per-window scores are reassembled by a single RCCL all-gather before the (rank-0) AUC computation."""
import argparse
import os
import sys
import tempfile
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL across processes (see bench.py); before HIP loads

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from mocodad_amd.data import synthetic  # noqa: E402
from mocodad_amd.models.mocodad import MoCoDAD  # noqa: E402
from mocodad_amd.parallel import WindowShard  # noqa: E402
from mocodad_amd.utils.argparser import load_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="MoCoDAD (MI355X)")
    ap.add_argument("-c", "--config", type=str, required=True)
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate on N synthetic clips instead of dataset files")
    ap.add_argument("--frames-per-clip", type=int, default=200)
    ap.add_argument("--device-windows", action="store_true",
                    help="upload per-person trajectories once and let the kernels window + transform them on load "
                         "(instead of materialising seg_len x num_transform copies on the host)")
    ap.add_argument("--random-init", action="store_true", help="score with seeded random-init weights when the checkpoint "
                    "is missing (otherwise a missing checkpoint is an error)")
    ap.add_argument("--dist-backend", default="nccl", help="'nccl' (= RCCL over xGMI) or 'gloo' (tests with several ranks on one GPU)")
    ap.add_argument("--dump-scores", type=str, default=None, help="rank 0 writes the gathered window scores + AUC to this .npz")
    cli = ap.parse_args()
    args = load_config(cli.config)
    if hasattr(args, "diffusion_on_latent"):
        raise NotImplementedError("the latent-diffusion variant (MoCoDADlatent) is outside the accelerated path")

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    # ranks on THIS node (torchrun exports LOCAL_WORLD_SIZE; a multi-node job has world > ndev with one GPU per rank)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if local_world > ndev and cli.dist_backend == "nccl":
        # RCCL refuses two ranks on one GPU (and would otherwise hang in its bootstrap)
        raise SystemExit(f"eval_MoCoDAD.py: {local_world} ranks on this node but {ndev} GPU(s) visible; the nccl backend needs one "
                         "rank per GPU (--dist-backend gloo is for tests that share a GPU)")
    local = local % ndev   # (several ranks per GPU only in the 1-GPU tests, with --dist-backend gloo)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    use_dist = world > 1 or "LOCAL_RANK" in os.environ     # under torch.distributed.run: RCCL path even with one rank
    if use_dist:
        import torch.distributed as dist
        if cli.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(cli.dist_backend)

    tw = None
    if cli.synthetic and cli.device_windows:
        from mocodad_amd.data.windows import TrajectoryWindows
        trajs, gts = synthetic.make_trajectories(n_clips=cli.synthetic, frames_per_clip=cli.frames_per_clip, seed=args.seed)
        tw = TrajectoryWindows(trajs, args.seg_len, args.num_transform).to(dev)
        data, trans, meta, frames = tw, tw.trans.long(), tw.meta, tw.frames
        gt_dir = tempfile.mkdtemp(prefix="mocodad_gt_")
        synthetic.write_gt(gt_dir, gts)
        args.gt_path = gt_dir
    elif cli.synthetic:
        data, trans, meta, frames, gts = synthetic.make_dataset(n_clips=cli.synthetic, frames_per_clip=cli.frames_per_clip,
                                                                seg_len=args.seg_len, num_transform=args.num_transform,
                                                                seed=args.seed)
        gt_dir = tempfile.mkdtemp(prefix="mocodad_gt_")
        synthetic.write_gt(gt_dir, gts)
        args.gt_path = gt_dir
    else:
        raise SystemExit("dataset files are read by the reference's own pipeline (utils/dataset.py); "
                         "pass --synthetic N here, or feed your DataLoader's batches to MoCoDAD.test_step")

    torch.manual_seed(int(getattr(args, "seed", 0)))      # without a checkpoint every rank must draw the same weights
    model = MoCoDAD(args).to(dev)
    if cli.synthetic:
        model.dataset_name = "synthetic"      # synthetic clips have their own lengths: no HR-Avenue / UBnormal frame masks
    ckpt = os.path.join(args.ckpt_dir, args.load_ckpt)
    if os.path.exists(ckpt):
        model.load_state_dict(torch.load(ckpt, map_location="cpu", weights_only=False)["state_dict"])
    elif not cli.random_init:
        raise SystemExit(f"checkpoint {ckpt} not found (pass --random-init to score with seeded random-init weights)")

    n = len(tw) if tw is not None else data.shape[0]
    shard = WindowShard(n, rank, world)
    if use_dist:
        shard.host_meta = (trans.numpy(), meta.numpy(), frames.numpy())
        model.shard = shard
    model.save_tensors = False
    import sklearn.metrics  # noqa: F401  (imported here, not inside the timed region: ~0.3 s on first use)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    model.on_test_epoch_start()
    with torch.no_grad():
        batches = tw.batches(args.batch_size, shard.lo, shard.hi) if tw is not None else \
            synthetic.batches((data, trans, meta, frames), args.batch_size, shard.lo, shard.hi)
        for i, batch in enumerate(batches):
            model._calls = shard.lo + i * args.batch_size     # global window id keys the noise stream
            model.test_step(batch, i)
    ev1.record()                             # (no synchronisation here: the epoch end's host work runs under the queued batches)
    auc = model.on_test_epoch_end()          # gather of the scores + frame-score assembly (device) + roc_auc_score (host)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t1 = t0 + ev0.elapsed_time(ev1) * 1e-3   # the scoring loop alone: device time from the first launch to the last batch's end
    if rank == 0:
        # end to end (test_step loop + gather + frame-score assembly + AUC) beside the scoring loop alone (the kernels)
        print(f"windows: {n}  gpus: {world}  time: {dt:.3f}s  ({n / dt:.0f} clips/s end to end; scoring loop {t1 - t0:.3f}s = {n / (t1 - t0):.0f} clips/s "
              f"+ epoch end {dt - (t1 - t0):.3f}s)  batch: {args.batch_size}  noise_steps: {args.noise_steps}  samples: {args.n_generated_samples}  AUC: {auc:.6f}")
        if cli.dump_scores:
            import numpy as np
            np.savez(cli.dump_scores, scores=model.last_scores, auc=np.float64(auc))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
