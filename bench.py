#!/usr/bin/env python3
"""bench.py — throughput of the MoCoDAD anomaly-scoring hot path on MI355X.

A "step" = one MoCoDAD.forward-equivalent call of the HIP path (condition encoder + S*(ns-1) U-Net
passes + DDPM updates + per-sample loss + 'best' aggregation) over one batch of synthetic pose windows
that is already resident in HBM.  Workload = BASELINE.json configs[1]: HR-Avenue-shaped windows
(B=1024 per step as in config/Avenue/mocodad_test.yaml, seg_len 6 = 3 condition + 3 denoised frames,
17 joints), noise_steps=10, 5 generated samples, inject conditioning, in-kernel Philox noise.

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (see the driver contract in the task description)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic FLOP per window (BASELINE.md §3 / SURVEY.md §8d): P * F_unet(T_u=3) + F_cond(T_c=3)
F_UNET_T3 = 4_290_352
F_COND_T3 = 545_904
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
# HBM bytes per launch of the default workload from the rocprofv3 PMC passes (see profiles/)
HBM_TRAFFIC_DEFAULT = 7.65e6   # (2 x 2706.0 + 80 + 2 x 1010.5 + 64 + 2 x 32.5 + 4) KB: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE
HBM_TRAFFIC_SOURCE = "profiles/r01t_pmc.txt"
PEAK_HBM_GBS = 8000.0


def load_weights():
    d = np.load(os.path.join(ROOT, "tests", "golden", "weights_inject.npz"))
    w = {k: d[k] for k in d.files}
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    return {k: torch.from_numpy(v) for k, v in w.items()}, cfg


def synth_windows(n, seg_len, seed):
    """HR-Avenue-shaped synthetic input: smooth per-joint random walks, robust-scaled-like, clipped to +-5."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(n, 2, 1, 17, generator=g)
    steps = torch.randn(n, 2, seg_len, 17, generator=g) * 0.15
    return (base + torch.cumsum(steps, dim=2)).clamp_(-5, 5).float().contiguous()


def usable_cores():
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = min(n, max(1, int(float(q[0]) / float(q[1]))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    return max(1, n)


def cpu_baseline(sd, ns, S, budget_s=15.0):
    """The oracle (a PyTorch-CPU op-for-op port of the reference path) timed on this host's cores, on a
    bounded sample: a 256-window probe sizes the timed sample to about `budget_s` seconds."""
    from oracle import mocodad_oracle as O
    threads = min(usable_cores(), 64)
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(123)

    def run(n, seed):
        data = synth_windows(n, 6, seed)
        noise = torch.randn(S, ns - 1, n, 2, 3, 17, generator=g)
        t0 = time.perf_counter()
        with torch.no_grad():
            O.score(sd, data, noise, noise_steps=ns, aggregation="best")
        return time.perf_counter() - t0

    run(64, 1)                      # warm-up (thread pool, allocator)
    probe = run(256, 2)
    n = int(min(16384, max(256, 256 * budget_s / max(probe, 1e-3))))
    n = max(256, n // 256 * 256)
    dt = run(n, 3)
    return {"value": round(n / dt, 2), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"one batch of {n} windows, ns={ns}, S={S}, oracle/mocodad_oracle.py (PyTorch CPU, {threads} threads), {dt:.1f}s"}


def opt_in_line(args):
    """Run this benchmark once more with --bf16x3 in a child process and return its value / kernel time (or None)."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--bf16x3", "--no-cpu-baseline", "--steps", str(args.steps),
                            "--warmup", str(args.warmup), "--batch", str(args.batch), "--noise-steps", str(args.noise_steps),
                            "--samples", str(args.samples)], capture_output=True, text=True, timeout=300)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        return {"value": d["value"], "unit": d["unit"], "kernel_ms_per_step": d["roofline"]["kernel_ms_per_step"],
                "note": "OPT-IN, not the headline: channel GEMMs as hi*hi + hi*lo + lo*hi on the bf16 matrix path, f32 accumulate; "
                        "scores within 2e-6 of the golden vectors (tests/test_bf16x3_gpu.py); DESIGN.md section 3"}
    except Exception as e:  # noqa: BLE001 -- never let the informational field break the benchmark line
        return {"value": None, "error": str(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="windows per step per GPU")
    ap.add_argument("--noise-steps", type=int, default=10)
    ap.add_argument("--samples", type=int, default=5)
    ap.add_argument("--streams", type=int, default=1, help="HIP streams consecutive batches alternate over (2: the ramp of "
                    "batch i+1 fills the tail of batch i, +2 %; per-launch durations then overlap, so the default keeps 1)")
    ap.add_argument("--bf16x3", action="store_true", help="OPT-IN, not the headline: channel GEMMs on the bf16 matrix path with both "
                    "operands split into bf16 pairs (hi*hi + hi*lo + lo*hi, fp32 accumulate; scores within ~1e-6 of the fp32 path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0, help="seconds of CPU work for the cpu_baseline sample")
    args = ap.parse_args()
    if args.bf16x3:
        os.environ["MCD_BF16X3"] = "1"     # read once by the library, before its first launch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")
    # under torch.distributed.run (LOCAL_RANK set) the RCCL path is used even with one rank, so a 1-process launch
    # exercises exactly what the N-GPU launches do
    use_dist = world > 1 or "LOCAL_RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=dev)

    from mocodad_amd.engine import HipScorer
    sd, cfg = load_weights()
    ns, S, B = args.noise_steps, args.samples, args.batch
    sc = HipScorer(sd, strategy="inject", seg_len=6, cond_idx=[0, 1, 2], corrupt_idx=[3, 4, 5],
                   cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device=dev)
    # weak scaling: every rank owns its own shard of B windows per step (global window ids keep the
    # Philox streams distinct and independent of the number of GPUs)
    data = synth_windows(B, 6, 1000 + rank).to(dev)
    # Window scores stay on their rank while the job runs; ONE all-gather after the last batch reassembles them before
    # the AUC (SURVEY.md 8e, and what eval_MoCoDAD.py does) -- it is inside the timed region.
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else []

    def run(n_steps, seed0, events=None):
        # same buffer / collective size in the warm-up and in the timed run (no size-dependent lazy set-up inside the latter)
        scores = torch.zeros(max(args.steps, n_steps), B, device=dev, dtype=torch.float32)
        main = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(main)
        for i in range(n_steps):
            with torch.cuda.stream(streams[i % len(streams)] if streams else main):
                if events is not None:
                    events[i][0].record()
                loss, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=seed0 + i, first_window_id=rank * B)
                if events is not None:
                    events[i][1].record()   # brackets the scoring launches (cond encoder + persistent kernel) on this stream
                sc.aggregate(data, loss, None, "best", noise_steps=ns, want_pose=False, out=scores[i])
        for st in streams:
            main.wait_stream(st)
        if use_dist:
            gathered = torch.empty(world * scores.numel(), device=dev, dtype=torch.float32)
            dist.all_gather_into_tensor(gathered, scores.view(-1))   # RCCL over xGMI
            return gathered
        return scores

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup, 0)
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    best = run(args.steps, 100, ev)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if streams:
        # launches on different streams overlap, so a launch's own start-to-end time counts its neighbour's work too;
        # the roofline then uses the timed region's average time per launch instead
        kern_ms = dt / args.steps * 1e3
    assert torch.isfinite(best).all()

    if rank == 0:
        total = world * B * args.steps
        P = S * (ns - 1)
        flop_per_window = P * F_UNET_T3 + F_COND_T3
        achieved = B * flop_per_window / (kern_ms * 1e-3) / 1e12
        out = {
            "metric": "pose-clips/sec (whole node) @ noise_steps=10, 5 samples",
            "value": round(total / dt, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16x3 split operands, f32 accumulate (opt-in)" if args.bf16x3 else "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: HR-Avenue-shaped windows (seg_len 6 = 3 cond + 3 denoised, 17 joints), "
                                   f"noise_steps={ns}, {S} generated samples, inject conditioning, 'best' aggregation",
                       "windows_per_step_per_gpu": B, "denoiser_passes_per_window": P, "weights": "seeded random init (tests/golden/weights_inject.npz)",
                       "noise": "in-kernel Philox4x32-10", "parallelism": f"windows sharded over {world} GPU(s), all-gather of scores",
                       "streams": max(args.streams, 1)},
            "roofline": {"bound": "mfma", "kernel": ("score_kernel<3,2,4,bf16x3>" if args.bf16x3 else "score_kernel<3,2,4>") + " (+ cond_fast_kernel<3,2>)", "achieved": round(achieved, 3),
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                         "flop_per_window": flop_per_window, "kernel_ms_per_step": round(kern_ms, 4),
                         "hbm_algorithmic_bytes_per_window": 820,
                         # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE);
                         # measured for the default workload only
                         "traffic": HBM_TRAFFIC_DEFAULT if (B, ns, S) == (1024, 10, 5) else None,
                         "traffic_source": HBM_TRAFFIC_SOURCE,
                         # the HBM roofline north_star asks to see beside it: measured bytes / launch time vs 8 TB/s
                         "hbm": ({"achieved_GBps": round(HBM_TRAFFIC_DEFAULT / (kern_ms * 1e-3) / 1e9, 2), "peak_GBps": 8000.0,
                                  "frac": round(HBM_TRAFFIC_DEFAULT / (kern_ms * 1e-3) / 8e12, 6)}
                                 if (B, ns, S) == (1024, 10, 5) else None)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, ns, S, args.cpu_budget)
        if world == 1 and not args.bf16x3 and not args.no_cpu_baseline and not use_dist:
            # informational only (never `value`): the opt-in split-bf16 GEMM path on the same box, in a child process
            # because the library reads its switch once per process
            out["opt_in_bf16x3"] = opt_in_line(args)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
