#!/usr/bin/env python3
"""bench.py — throughput of the MoCoDAD anomaly-scoring hot path on MI355X.

A "step" = one MoCoDAD.forward-equivalent call of the HIP path (condition encoder + S*(ns-1) U-Net
passes + DDPM updates + per-sample loss + 'best' aggregation; ONE kernel launch whenever the batch fills the device -- the
JSON line says which form ran, `roofline.launches_per_step`) over one batch of synthetic pose windows that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: HR-Avenue-shaped windows
(B=1024 per step as in config/Avenue/mocodad_test.yaml, seg_len 6 = 3 condition + 3 denoised frames,
17 joints), noise_steps=10, 5 generated samples, inject conditioning, in-kernel Philox noise.

  python bench.py [--gpus N --steps K --warmup W] [--config avenue|stc|ubnormal_concat|seq24] [--scaling weak|strong]

With --gpus N > 1 and no WORLD_SIZE in the environment the script launches its own N ranks (one per GPU,
torch.distributed.run on 127.0.0.1 with a free port); started under torch.distributed.run it uses the ranks it is given.
Rank 0 prints ONE JSON line (see the driver contract in the task description)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# dmabuf IPC (the host driver supports no legacy IPC handles): without it RCCL across processes fails with
# `hipIpcGetMemHandle: invalid argument`.  The boxes export it already; set before the HIP runtime loads, for any other launcher.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic FLOP per window (BASELINE.md §3 / SURVEY.md §8d): P * F_unet(T_u) + F_cond(T_c), from the per-layer MAC count of
# SURVEY.md §8 a8 / a10 (reproduces its table: F_unet(3) = 4 290 352, (6) = 8 799 400, (12) = 18 524 464, (24) = 40 802 464;
# F_cond(3) = 545 904, (12) = 2 484 720)
_UNET_LAYERS = [(2, 16, 17), (16, 32, 17), (32, 32, 17), (32, 64, 12), (64, 64, 12), (64, 128, 10), (128, 64, 10), (64, 64, 12),
                (64, 32, 12), (32, 32, 17), (32, 2, 17)]            # (C_in, C_out, V)
_RESAMPLERS = [(32, 17, 12), (64, 12, 10), (64, 10, 12), (32, 12, 17)]  # (C, V_in, V_out)


def f_unet(T, E=16):
    mac = sum(ci * V * T * T + ci * T * V * V + co * ci * T * V * (1 + (ci != co)) + co * E for ci, co, V in _UNET_LAYERS)
    mac += sum(C * T * vi * vo for C, vi, vo in _RESAMPLERS)
    return 2 * mac


def f_cond(T, chans=(32, 16, 32, 32), E=16):
    mac, ci = 0, 2
    for co in chans:
        mac += ci * 17 * T * T + ci * T * 17 * 17 + co * ci * T * 17 * (1 + (ci != co))
        ci = co
    return 2 * (mac + ci * T * 17 * E)


assert (f_unet(3), f_unet(12), f_cond(3)) == (4_290_352, 18_524_464, 545_904)
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: FP32 vector == FP32 MFMA peak
PEAK_HBM_GBS = 8000.0
# rocprofv3 PMC passes of this round's kernels committed under profiles/ (tools/profile_set.sh): per-launch counter averages
# of the profiled run; `roofline.traffic` and the matrix-pipe statistics of the bench line are READ from these files (they are
# measurements of the same command under the profiler, not of this run) when the workload matches the profiled one
PMC_SET = "r06zz"           # tools/profile_set.sh's run directory of the committed set: profiles/<PMC_SET>_<config>_pmc.txt
PMC_PROFILES = {"avenue": (1024, 10, 5), "stc": (2048, 10, 5), "ubnormal_concat": (1024, 10, 5), "seq24": (1024, 50, 8),
                "concat24": (1024, 10, 5), "concat32": (1024, 10, 5)}      # config -> (windows, noise_steps, samples) of the profiled command
CLOCK_GHZ = 2.4            # the clock the FP32 peak is quoted at (256 CUs x 4 SIMDs x 64 FLOP/cycle x 2.4 GHz = 157.3 TFLOP/s)


_SO_SHA = None


def loaded_library_sha256():
    global _SO_SHA
    if _SO_SHA is None:
        import hashlib
        from mocodad_amd import _lib
        _SO_SHA = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()
    return _SO_SHA


def pmc_profile(config, B, ns, S, flop_per_window, kern_ms, t_unet):
    """Per-launch HBM bytes (FETCH_SIZE x 2, the guide's gfx950 correction, + WRITE_SIZE, all kernels of a step) and the
    matrix-pipe statistics of the trajectory kernel from the committed PMC profile of this configuration."""
    if config not in PMC_PROFILES:
        return None
    pB, pns, pS = PMC_PROFILES[config]
    path = f"profiles/{PMC_SET}_{config}_pmc.txt"
    try:
        kernels, cur, man = {}, None, None
        for line in open(os.path.join(ROOT, path)):
            if line.startswith("# manifest:"):
                man = dict(kv.split("=", 1) for kv in line.split()[2:] if "=" in kv)
            if line.startswith("#") or not line.strip():
                continue
            if not line.startswith(" "):
                cur = kernels.setdefault(line.strip().replace("mcd::", ""), {})
            else:
                f = line.split()
                cur[f[0]] = float(f[1])
    except Exception:
        return None
    # the counters describe ONE library build: a profile whose manifest names another libmocodad_hip.so (or none) is refused
    # (`traffic` null), not silently attached to this run's kernels
    if not man or man.get("so_sha256") != loaded_library_sha256():
        return {"source": path, "refused": "the profile's manifest does not name the loaded libmocodad_hip.so "
                f"(profile: {(man or {}).get('so_sha256', 'no manifest')[:12]}, loaded: {loaded_library_sha256()[:12]}): regenerate with tools/profile_set.sh"}
    sk = next((v for k, v in kernels.items() if k.startswith("score_kernel") or k.startswith("score_tiled_kernel")), None)
    tiled = any(k.startswith("score_tiled_kernel") for k in kernels)
    if not sk:
        return None
    scale = (B * S * (ns - 1)) / float(pB * pS * (pns - 1))       # counters scale with the chain-passes of a launch
    out = {"source": path, "manifest": man, "profiled_workload": {"windows": pB, "noise_steps": pns, "samples": pS}}
    mfma = sk.get("SQ_INSTS_MFMA")
    if mfma:
        mfma *= scale
        # 16x16x4 fp32 MFMAs: 2048 FLOP and 32 cycles of one of the 1024 SIMDs each
        out["mfma_issued_per_launch"] = round(mfma)
        # the time mix (2 T^2 FLOP per (input channel, joint) of each of the 11 layers: sum of C_in V = 5396) runs on the
        # VALU (DPP FMAs), everything else on the matrix cores; what the issued MFMAs exceed that by is tile padding
        # (the slab-tiled kernel runs its time mix on the matrix cores too, layer 10's excepted: 32 x 17 = 544 of the 5396)
        vec_flop = S * (ns - 1) * 2 * t_unet * t_unet * (544 if tiled else 5396)
        out["useful_mfma_frac"] = round(min(1.0, B * (flop_per_window - vec_flop) / 2048.0 / mfma), 4)
        out["mfma_pipe_busy_frac"] = round(mfma * 32 / (1024 * CLOCK_GHZ * 1e9 * kern_ms * 1e-3), 4)
        if "SQ_INSTS_VALU" in sk:
            out["other_valu_per_mfma"] = round((sk["SQ_INSTS_VALU"] * scale - mfma) / mfma, 3)
        if "SQ_WAIT_ANY" in sk and sk.get("SQ_WAVE_CYCLES"):
            out["wave_wait_frac"] = round(sk["SQ_WAIT_ANY"] / sk["SQ_WAVE_CYCLES"], 4)     # a wave parked on any counter or barrier
    # barrier share of a pass from the committed in-kernel stage profile (an instrumented -DMCD_PROFILE build of the same kernel:
    # mean cycles a wave waits at the pass's workgroup barriers / cycles of the pass)
    try:
        import re
        m = re.search(r"mean wait per wave\s+(\d+)\s+\(pass total\s+(\d+)\)", open(os.path.join(ROOT, path.replace("_pmc.txt", "_stage_profile.txt"))).read())
        if m:
            out["barrier_wait_frac"] = round(int(m.group(1)) / int(m.group(2)), 4)
    except Exception:
        pass
    if (ns, S) == (pns, pS) and all("FETCH_SIZE" in v and "WRITE_SIZE" in v for v in kernels.values()):
        # same chain length: the bytes of a launch scale with its windows (inputs, scores, per-workgroup weight fetches, spills)
        out["hbm_bytes_per_step"] = sum(2 * v["FETCH_SIZE"] + v["WRITE_SIZE"] for v in kernels.values()) * 1024.0 * B / float(pB)
        if B != pB:
            out["hbm_bytes_scaled_from_windows"] = pB
    return out


# name -> (golden weights variant, windows per step, noise_steps, samples, description)
CONFIGS = {
    "avenue": ("inject", 1024, 10, 5, "BASELINE configs[1]: HR-Avenue-shaped windows (seg_len 6 = 3 cond + 3 denoised, 17 joints)"),
    "stc": ("inject", 2048, 10, 5, "BASELINE configs[2]: HR-ShanghaiTech-shaped windows (seg_len 6 = 3 cond + 3 denoised), batch 2048"),
    "ubnormal_concat": ("concat", 1024, 10, 5, "BASELINE configs[3]: HR-UBnormal-shaped windows, concat conditioning (U-Net on 6 frames, no encoder)"),
    "seq24": ("T12", 4096, 50, 8, "BASELINE configs[4]: synthetic N(0,1) windows, seq_len 24 = 12 cond + 12 denoised frames"),
    # shapes the reference accepts (mocodad.py:780-796) beyond BASELINE's configurations -- seeded random-init weights built here
    # (no reference-generated fixture holds these frame counts): ("rand:<strategy>:<seg_len>:<conditioning_indices>", ...)
    "seg10": ("rand:inject:10:2", 1024, 10, 5, "seg_len 10 = 5 cond + 5 denoised frames (inject), Avenue-shaped windows"),
    "seg16": ("rand:inject:16:2", 1024, 10, 5, "seg_len 16 = 8 cond + 8 denoised frames (inject; U-Net on 8 frames)"),
    "seg18": ("rand:inject:18:2", 1024, 10, 5, "seg_len 18 = 9 cond + 9 denoised frames (inject; U-Net on 9 frames)"),
    "seg20": ("rand:inject:20:2", 1024, 10, 5, "seg_len 20 = 10 cond + 10 denoised frames (inject), Avenue-shaped windows"),
    "seg4": ("rand:inject:4:2", 1024, 10, 5, "seg_len 4 = 2 cond + 2 denoised frames (inject; U-Net on 2 frames)"),
    "seg14": ("rand:inject:14:2", 1024, 10, 5, "seg_len 14 = 7 cond + 7 denoised frames (inject; U-Net on 7 frames)"),
    "seg22": ("rand:inject:22:2", 1024, 10, 5, "seg_len 22 = 11 cond + 11 denoised frames (inject; U-Net on 11 frames)"),
    "concat12": ("rand:concat:12:2", 1024, 10, 5, "seg_len 12, concat conditioning (U-Net on 12 frames, 6 of them denoised)"),
    "concat24": ("rand:concat:24:2", 1024, 10, 5, "seg_len 24, concat conditioning (U-Net on 24 frames: the slab-tiled MFMA kernel)"),
    "seg32": ("rand:inject:32:2", 1024, 10, 5, "seg_len 32 = 16 cond + 16 denoised frames (inject; U-Net on 16 frames: the slab-tiled MFMA kernel)"),
    "concat32": ("rand:concat:32:2", 1024, 10, 5, "seg_len 32, concat conditioning (U-Net on 32 frames: the slab-tiled MFMA kernel)"),
    # the 'E_unet' condition encoder (the U-Net's down path per window) at frame counts other than 3 / 6 / 12
    "seg10_eunet": ("rand:inject:10:2:E_unet", 1024, 10, 5, "seg_len 10 = 5 + 5 frames, 'E_unet' condition encoder"),
    "seg32_eunet": ("rand:inject:32:2:E_unet", 1024, 10, 5, "seg_len 32 = 16 + 16 frames, 'E_unet' condition encoder"),
}


def random_weights(strategy, seg_len, ci, arch="AE"):
    """Seeded random-init state_dict (the reference's key layout) for a shape without a fixture: BatchNorm statistics perturbed,
    last layer scaled so that the chain stays O(1) -- the recipe of tests/golden/gen_golden.py's benign fixtures."""
    import argparse
    from mocodad_amd.models.mocodad import MoCoDAD
    _, cfg = load_weights("inject")
    cfg = dict(cfg, conditioning_strategy=strategy, seg_len=seg_len, conditioning_indices=ci, conditioning_architecture=arch)
    cfg.setdefault("gt_path", cfg.get("test_path"))
    cfg.setdefault("ckpt_dir", "/tmp/mocodad_amd_ckpt")
    torch.manual_seed(1234)
    m = MoCoDAD(argparse.Namespace(**cfg))
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) + 0.5)
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=gen) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=gen) * 0.1)
        last = m.model.st_gcnnsu3[-1]
        last.tcn[0].weight.mul_(0.25)
        last.residual[0].weight.mul_(0.25)
    return {k: v.detach().clone() for k, v in m.state_dict().items()}, cfg


def load_weights(variant="inject"):
    if variant.startswith("rand:"):
        _, strategy, seg_len, ci, *arch = variant.split(":")
        return random_weights(strategy, int(seg_len), int(ci), *arch)
    d = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{variant}.npz"))
    w = {k: d[k] for k in d.files}
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    return {k: torch.from_numpy(v) for k, v in w.items()}, cfg


def frame_split(seg_len, ci, strat):
    if strat == "no_condition":
        return [], list(range(seg_len))
    if isinstance(ci, int):
        n = seg_len // ci
        return list(range(n)), list(range(n, seg_len))
    return list(ci), [i for i in range(seg_len) if i not in ci]


def synth_windows(n, seg_len, seed, iid=False):
    """HR-Avenue-shaped synthetic input: smooth per-joint random walks, robust-scaled-like, clipped to +-5
    (iid: i.i.d. N(0,1) as BASELINE configs[4] asks)."""
    g = torch.Generator().manual_seed(seed)
    if iid:
        return torch.randn(n, 2, seg_len, 17, generator=g).float().contiguous()
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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(sd, cfg, ns, S, budget_s=20.0):
    """The oracle (a PyTorch-CPU op-for-op port of the reference path) timed on this host's cores, on a bounded sample:
    a small probe sizes the timed sample to about `budget_s` seconds; a 1-thread figure on a smaller sample beside it."""
    from oracle import mocodad_oracle as O
    threads = min(usable_cores(), 64)
    seg_len, strat, ci = cfg["seg_len"], cfg["conditioning_strategy"], cfg["conditioning_indices"]
    _, xi = frame_split(seg_len, ci, strat)
    g = torch.Generator().manual_seed(123)

    def run(n, seed):
        data = synth_windows(n, seg_len, seed)
        noise = torch.randn(S, ns - 1, n, 2, len(xi), 17, generator=g)
        t0 = time.perf_counter()
        with torch.no_grad():
            O.score(sd, data, noise, noise_steps=ns, strategy=strat, conditioning_indices=ci, aggregation="best")
        return time.perf_counter() - t0

    torch.set_num_threads(threads)
    run(8, 1)                       # warm-up (thread pool, allocator)
    probe_n = 16
    probe = run(probe_n, 2)
    if probe < 0.5:                 # a light configuration: a larger probe sizes the sample more accurately
        probe_n = 128
        probe = run(probe_n, 2)
    n = int(min(16384, max(16, probe_n * 0.8 * budget_s / max(probe, 1e-3))))
    n = max(16, n // 16 * 16)
    dt = run(n, 3)
    torch.set_num_threads(1)
    n1 = int(min(256, max(4, 0.2 * budget_s * 1.5 * (n / dt) / threads)))      # about 0.2 x budget on one thread
    dt1 = run(n1, 4)
    torch.set_num_threads(threads)
    return {"value": round(n / dt, 2), "unit": "clips/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "one_thread": {"value": round(n1 / dt1, 2), "unit": "clips/s", "sample": f"{n1} windows, {dt1:.1f}s"},
            "sample": f"one batch of {n} windows, ns={ns}, S={S}, oracle/mocodad_oracle.py (PyTorch CPU, {threads} threads), {dt:.1f}s"}


def auc_vs_ref(sc, sd, cfg, ns, S, seeds=200, oracle_seeds=2, with_oracle=True):
    """`AUC vs ref` of BASELINE.json's metric, next to the throughput (untimed): a fixed synthetic clip set
    (mocodad_amd.data.synthetic.make_dataset: per-person trajectories with jerkier anomalous intervals, ground-truth frame
    masks) scored (a) by the benchmarked mode -- in-kernel Philox noise -- and (b) by the SAME kernel fed with torch.randn
    draws (the parity mode the golden trajectories pin to the reference within 1e-4 per score), `seeds` seeds each (the
    seed-to-seed spread of the AUC on a small clip set is ~0.03, so the means need that many; replicas of the clip set under
    distinct window ids share a launch); and (c), for `oracle_seeds` CPU-drawn noise tensors, by the CPU oracle (the reference
    path restated; checker only) against the kernel on the same draws.  Frame scores by mcd_frame_scores (mocodad.py:362-425 on
    device), AUC by sklearn's roc_auc_score (mocodad.py:428)."""
    from sklearn.metrics import roc_auc_score
    from mocodad_amd.data import synthetic
    from mocodad_amd.engine import FrameScoreAssembler
    seg_len, strat, ci = cfg["seg_len"], cfg["conditioning_strategy"], cfg["conditioning_indices"]
    _, xi = frame_split(seg_len, ci, strat)
    data, trans, meta, frames, gts = synthetic.make_dataset(n_clips=3, frames_per_clip=max(80, 3 * seg_len), persons_per_clip=2,
                                                            seg_len=seg_len, num_transform=2)
    N = data.shape[0]
    dev = sc.device
    asm = FrameScoreAssembler(gts, {}, num_transform=2, pad_size=-1, filter_kernel_size=3, frames_shift=2, device=dev)
    ddata, dtrans, dmeta, dframes = data.to(dev), trans.to(dev), meta.to(dev), frames.to(dev)
    auc = lambda best: float(roc_auc_score(asm.gt, asm(best, dtrans, dmeta, dframes)))
    # replicas per launch: bounded by the noise tensor of the parity mode (S x (ns-1) x N R x 2 x Tx x 17 floats <= ~1 GB)
    per_rep = S * max(ns - 1, 1) * N * 2 * len(xi) * 17 * 4
    R = int(max(1, min(50, seeds, (1 << 30) // per_rep)))
    calls = (seeds + R - 1) // R
    rep = ddata.repeat(R, 1, 1, 1)
    gen = torch.Generator(device=dev).manual_seed(2026)
    philox, randn = [], []
    for c in range(calls):
        z = torch.randn(S, max(ns - 1, 1), N * R, 2, len(xi), 17, device=dev, generator=gen)
        runs = (sc.score(rep, n_samples=S, noise_steps=ns, seed=4242, first_window_id=c * N * R)[0], sc.score(rep, n_samples=S, noise_steps=ns, noise=z)[0])
        for dst, loss in zip((philox, randn), runs):
            best = loss.min(1)[0].view(R, N)
            dst.extend(auc(best[r]) for r in range(R))
        del z
    a, b = np.array(philox), np.array(randn)
    n = len(a)
    se = float(np.sqrt((a.var(ddof=1) + b.var(ddof=1)) / n))
    out = {"hip": round(float(a.mean()), 5), "ref_noise": round(float(b.mean()), 5), "abs_diff": round(abs(float(a.mean() - b.mean())), 5),
           "std_err_of_diff": round(se, 5), "seeds": n, "seed_std": round(float(np.concatenate([a, b]).std(ddof=1)), 5), "windows": N,
           "note": "hip = benchmarked mode (in-kernel Philox); ref_noise = same kernel on torch.randn draws (the mode pinned to the "
                   "reference); means over the seeds of the AUC on a fixed synthetic clip set"}
    if with_oracle and oracle_seeds > 0:
        from oracle import mocodad_oracle as O
        ao, ah, same = [], [], []
        for k in range(oracle_seeds):
            g = torch.Generator().manual_seed(k)
            noise = torch.randn(S, max(ns - 1, 1), N, 2, len(xi), 17, generator=g)
            best = sc.score(ddata, n_samples=S, noise_steps=ns, noise=noise)[0].min(1)[0]
            with torch.no_grad():
                ref = O.score(sd, data, noise, noise_steps=ns, strategy=strat, conditioning_indices=ci, aggregation="best")[1]
            ao.append(round(auc(ref), 6)); ah.append(round(auc(best), 6)); same.append(float((best.cpu() - ref).abs().max()))
        out["oracle_same_noise"] = {"auc_oracle": ao, "auc_hip": ah, "max_abs_score_diff": float(f"{max(same):.3e}"), "seeds": oracle_seeds,
                                    "note": "CPU oracle (reference path restated) and the kernel on the same CPU-drawn noise: same scores, same AUC"}
    return out


class GpuSampler:
    """Shader clock (MHz) and socket power (W) of one GPU, sampled from a background thread while the kernels run: amdsmi
    (the library behind amd-smi) when it loads, else torch.cuda.clock_rate / power_draw (amdsmi through torch)."""

    def __init__(self, index, period_s=0.1):
        import threading
        self.index, self.period, self.samples, self._stop = index, period_s, [], threading.Event()
        self._thread = threading.Thread(target=self._loop, daemon=True)
        self._read = self._reader()

    def _reader(self):
        idx = self.index
        try:
            import amdsmi
            try:
                amdsmi.amdsmi_init()
            except Exception:
                pass
            h = amdsmi.amdsmi_get_processor_handles()[idx]

            def read():
                clk = pw = None
                try:
                    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
                    cl = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 65535]
                    clk = float(np.mean(cl)) if cl else (float(m["current_gfxclk"]) if isinstance(m.get("current_gfxclk"), (int, float)) and 0 < m["current_gfxclk"] < 65535 else None)
                    for k in ("current_socket_power", "average_socket_power"):
                        if isinstance(m.get(k), (int, float)) and 0 < m[k] < 65535:
                            pw = float(m[k])
                            break
                except Exception:
                    pass
                if clk is None:
                    try:
                        clk = float(amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX)["clk"])
                    except Exception:
                        pass
                if pw is None:
                    try:
                        pi = amdsmi.amdsmi_get_power_info(h)
                        for k in ("current_socket_power", "average_socket_power", "socket_power"):
                            if isinstance(pi.get(k), (int, float)) and pi[k] > 0:
                                pw = float(pi[k])
                                break
                    except Exception:
                        pass
                return clk, pw
            read()
            self.source = "amdsmi"
            return read
        except Exception:
            pass

        def read_torch():
            clk = pw = None
            try:
                clk = float(torch.cuda.clock_rate(idx))
            except Exception:
                pass
            try:
                pw = float(torch.cuda.power_draw(idx)) / 1e3
            except Exception:
                pass
            return clk, pw
        self.source = "torch.cuda"
        return read_torch

    def _loop(self):
        t0 = time.perf_counter()
        while not self._stop.is_set():
            clk, pw = self._read()
            self.samples.append((time.perf_counter() - t0, clk, pw))
            self._stop.wait(self.period)

    def __enter__(self):
        self._thread.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._thread.join(timeout=2)

    def summary(self):
        clk = [c for _, c, _ in self.samples if c]
        pw = [p for _, _, p in self.samples if p]
        st = lambda v: {"min": round(min(v), 1), "mean": round(float(np.mean(v)), 1), "max": round(max(v), 1)} if v else None
        return {"source": self.source, "samples": len(self.samples), "sclk_mhz": st(clk), "socket_power_w": st(pw),
                "sclk_mhz_first_last": [round(clk[0], 1), round(clk[-1], 1)] if clk else None}


def sustained_run(launch, seconds, est_step_ms, windows_per_step, flop_per_window, dev_index):
    """Back-to-back steps for at least `seconds` (never `value`): per-step HIP-event times, throughput of the first and of the
    last second, and the shader clock / socket power sampled while the kernels run -- what the burst of the headline run
    (tens of ms at the driver's --steps 20) cannot show: does the clock hold under seconds of fp32 MFMA?"""
    n_steps = max(8, int(np.ceil(seconds * 1e3 / max(est_step_ms, 1e-3) * 1.02)))
    chunk = max(1, int(np.ceil(250.0 / max(est_step_ms, 1e-3))))           # ~0.25 s of launches per chunk, two chunks in flight
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]
    for e in ev:
        e.record()
    torch.cuda.synchronize()
    with GpuSampler(dev_index) as smp:
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(n_steps):
            launch(i)
            ev[i + 1].record()
            if i % chunk == chunk - 1 and i >= 2 * chunk:
                ev[i + 1 - 2 * chunk].synchronize()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    ends = np.array([ev[0].elapsed_time(e) for e in ev[1:]])               # ms since the start of the region
    step = np.diff(np.concatenate([[0.0], ends]))
    total_ms = float(ends[-1])
    first = min(int(np.searchsorted(ends, 1000.0)) + 1, n_steps)        # (a region shorter than 1 s: the whole of it, both ways)
    last = min(n_steps - int(np.searchsorted(ends, total_ms - 1000.0)), n_steps)
    rate = lambda n, ms: windows_per_step * n / (ms * 1e-3)
    r_first = rate(first, float(ends[first - 1]))
    r_last = rate(last, total_ms - float(ends[n_steps - last - 1]) if last < n_steps else total_ms)
    frac = lambda r: round(r * flop_per_window / 1e12 / PEAK_FP32_TFLOPS, 4)
    return {"seconds": round(total_ms / 1e3, 3), "wall_seconds": round(wall, 3), "steps": n_steps,
            "value": round(rate(n_steps, total_ms), 1), "unit": "clips/s", "frac": frac(rate(n_steps, total_ms)),
            "first_second": {"value": round(r_first, 1), "frac": frac(r_first)}, "last_second": {"value": round(r_last, 1), "frac": frac(r_last)},
            "droop_last_vs_first": round(r_last / r_first - 1.0, 5),
            "step_ms": {"p5": round(float(np.percentile(step, 5)), 4), "p50": round(float(np.percentile(step, 50)), 4),
                        "p95": round(float(np.percentile(step, 95)), 4), "max": round(float(step.max()), 4)},
            "gpu": smp.summary(),
            "note": "same launches as the timed region, back to back for the stated time; HIP events per step; informational, never `value`"}


def e2e_run(sd, cfg, ns, S, batch, n_clips, frames_per_clip, dev, kernel_rate):
    """End to end, as eval_MoCoDAD.py runs it (the reference's caller: mocodad.py:230-274, eval_MoCoDAD.py:33-38): synthetic
    per-person trajectories uploaded once, windows + test-time transforms cut on the device, the test_step loop over batches of
    `batch` windows, then on_test_epoch_end = collation + mcd_frame_scores + roc_auc_score.  Never `value`."""
    import argparse
    import tempfile
    from mocodad_amd.data import synthetic
    from mocodad_amd.data.windows import TrajectoryWindows
    from mocodad_amd.models.mocodad import MoCoDAD
    import sklearn.metrics  # noqa: F401
    c = dict(cfg, noise_steps=ns, n_generated_samples=S, batch_size=batch, aggregation_strategy="best", model_return_value="loss",
             save_tensors=False, dataset_choice="synthetic", pad_size=-1, filter_kernel_size=3, frames_shift=2)
    trajs, gts = synthetic.make_trajectories(n_clips=n_clips, frames_per_clip=frames_per_clip, seed=999)
    gt_dir = tempfile.mkdtemp(prefix="mocodad_gt_")
    synthetic.write_gt(gt_dir, gts)
    c["gt_path"] = c["test_path"] = gt_dir
    c.setdefault("ckpt_dir", "/tmp/mocodad_amd_ckpt")
    t_build = time.perf_counter()
    tw = TrajectoryWindows(trajs, c["seg_len"], c["num_transform"]).to(dev)
    t_build = time.perf_counter() - t_build
    m = MoCoDAD(argparse.Namespace(**c)).to(dev)
    m.load_state_dict(sd)
    m.dataset_name = "synthetic"
    n = len(tw)
    warm = tw.batch(0, min(batch, n))
    m.on_test_epoch_start()
    with torch.no_grad():
        m.test_step(warm, 0)            # packs the weights, sizes the workspace (outside the timed region, like a checkpoint load)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    m.on_test_epoch_start()
    with torch.no_grad():
        for i, b in enumerate(tw.batches(batch)):
            m._calls = i * batch
            m.test_step(b, i)
    ev1.record()        # (no synchronisation: the epoch end's host-side first-use work runs under the batches still queued)
    auc = m.on_test_epoch_end()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t1 = t0 + ev0.elapsed_time(ev1) * 1e-3      # the scoring loop alone: device time from the first launch to the last batch's end
    return {"windows": n, "batch": batch, "clips": n_clips, "frames_per_clip": frames_per_clip, "seconds": round(t2 - t0, 4),
            "value": round(n / (t2 - t0), 1), "unit": "clips/s", "scoring_seconds": round(t1 - t0, 4), "epoch_end_seconds": round(max(t2 - t1, 0.0), 4),
            "vs_kernel_rate": round(n / (t2 - t0) / kernel_rate, 4) if kernel_rate else None, "auc": round(float(auc), 6),
            "host_window_index_seconds": round(t_build, 3),
            "note": "MoCoDAD.test_step loop over device-cut windows + on_test_epoch_end (gather, mcd_frame_scores, roc_auc_score); "
                    "trajectories resident in HBM; informational, never `value`"}


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) and relay rank 0's line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.exit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 200; 5 for seq24: 0.39 s per step)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps right before the timed ones (default 20; 2 for seq24)")
    ap.add_argument("--preroll-ms", type=float, default=200.0,
                    help="untimed steps run for this long BEFORE the warm-up, to bring the GPU out of its idle clocks (the sclk ramp from "
                         "~100 MHz takes tens of ms: tools/clock_probe.sh; with 3 warm-up steps = 7 ms the first timed steps ran below 2.4 GHz)")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="avenue", help="workload shape (default: BASELINE configs[1])")
    ap.add_argument("--batch", type=int, default=None, help="windows per step: per GPU (weak scaling) or in total (strong)")
    ap.add_argument("--noise-steps", type=int, default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every GPU scores --batch windows per step; strong: --batch windows per step are split over the GPUs")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams consecutive batches alternate over (2: the ramp of "
                    "batch i+1 fills the tail of batch i, +2 %%; per-launch durations then overlap, so the default keeps 1)")
    ap.add_argument("--split", type=int, default=0, help="tuning / A-B: workgroups per window group (0 = library's choice; 1 = one "
                    "launch with encoder and aggregation inside; n_samples = one trajectory per workgroup, 3 launches)")
    ap.add_argument("--variant", type=int, default=0, help="tuning / A-B: alternative workgroup shape of the trajectory kernel (MCD_OPT_VARIANT)")
    ap.add_argument("--ref-value", type=float, default=None, help="the 1-GPU `value` of the same configuration: the line then carries "
                    "scaling_efficiency = value / (n_gpus x ref) (weak) or value / ref / n_gpus (strong); tools/scale_check.sh passes it")
    ap.add_argument("--cond-generic", action="store_true", help="tuning / A-B: condition encoder as its own (runtime-channel-list) launch even "
                    "when the workgroups own whole windows (MCD_OPT_COND_GENERIC)")
    ap.add_argument("--phase", type=int, default=0, help="tuning experiment: start offset of the second half of the grid, x 1024 cycles")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-auc", action="store_true", help="skip the (untimed) AUC-vs-reference leg")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational H2D-inclusive and opt-in legs")
    ap.add_argument("--min-seconds", type=float, default=0.0, help="make the TIMED region at least this long (steps = max(--steps, what fills it)): "
                    "sustained throughput as `value`")
    ap.add_argument("--sustained-seconds", type=float, default=10.0, help="length of the informational `sustained` leg (0 = skip; skipped under --no-extras)")
    ap.add_argument("--e2e-windows", type=int, default=None, help="windows of the informational end-to-end leg (default: about 0.5 s of scoring; 0 = skip)")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of CPU work for the cpu_baseline sample (all threads + one thread)")
    ap.add_argument("--dist-backend", default="nccl", help="'nccl' (= RCCL over xGMI, the default) or 'gloo' (tests that "
                    "place several ranks on one GPU)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 5 if args.config == "seq24" else 200
    if args.warmup is None:
        args.warmup = 2 if args.config == "seq24" else 20

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    ndev = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))      # ranks on THIS node (bench.py itself is single-node)
    if local_world > ndev and args.dist_backend == "nccl":
        # RCCL refuses two ranks on one GPU (and would otherwise hang in its bootstrap)
        print(f"bench.py: {local_world} ranks on this node but {ndev} GPU(s) visible; one rank per GPU is required with the nccl backend", file=sys.stderr)
        sys.exit(2)
    dev_index = local_rank % ndev          # (several ranks per GPU only in the 1-GPU tests, with --dist-backend gloo)
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    # under torch.distributed.run (LOCAL_RANK set) the collective path is used even with one rank, so a 1-process launch
    # exercises exactly what the N-GPU launches do
    use_dist = world > 1 or "LOCAL_RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)
    coll_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    from mocodad_amd.engine import HipScorer
    from mocodad_amd.parallel import shard_range
    variant, B_cfg, ns_cfg, S_cfg, desc = CONFIGS[args.config]
    sd, cfg = load_weights(variant)
    ns = args.noise_steps or ns_cfg
    S = args.samples or S_cfg
    B_arg = args.batch or B_cfg
    seg_len, strat = cfg["seg_len"], cfg["conditioning_strategy"]
    ci, xi = frame_split(seg_len, cfg["conditioning_indices"], strat)
    sc = HipScorer(sd, strategy=strat, seg_len=seg_len, cond_idx=ci, corrupt_idx=xi,
                   cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], cond_unet=cfg.get("conditioning_architecture") == "E_unet",
                   device=dev, options=dict(split=args.split, phase=args.phase, variant=args.variant,
                                **({"cond_generic": 1} if args.cond_generic else {})))
    if args.scaling == "weak":
        # every rank owns its own shard of B windows per step (global window ids keep the Philox streams distinct and
        # independent of the number of GPUs)
        lo, hi = rank * B_arg, (rank + 1) * B_arg
        B_total = world * B_arg
    else:
        lo, hi = shard_range(B_arg, rank, world)
        B_total = B_arg
    B = hi - lo
    per = -(-B_total // world)      # common (padded) shard length of the single all-gather
    iid = args.config == "seq24"
    data_host = synth_windows(B_total if args.scaling == "strong" else B, seg_len, 1000 + (0 if args.scaling == "strong" else rank), iid)
    if args.scaling == "strong":
        data_host = data_host[lo:hi].contiguous()
    data = data_host.to(dev)
    # Window scores stay on their rank while the job runs; ONE all-gather after the last batch reassembles them before
    # the AUC (SURVEY.md 8e, and what eval_MoCoDAD.py does) -- it is inside the timed region.
    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else []
    gather_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))

    def run(scorer, n_steps, seed0, events=None, src=None):
        # same buffer / collective size in the warm-up and in the timed run (no size-dependent lazy set-up inside the latter)
        scores = torch.zeros(max(args.steps, n_steps), per, device=dev, dtype=torch.float32)
        main = torch.cuda.current_stream()
        for st in streams:
            st.wait_stream(main)
        for i in range(n_steps):
            with torch.cuda.stream(streams[i % len(streams)] if streams else main):
                x = data if src is None else src.to(dev, non_blocking=True)      # src: host windows (PCIe-inclusive leg)
                if events is not None:
                    events[i][0].record()
                if B > 0:      # condition encoder + all trajectories + 'best' over the samples: one launch
                    scorer.score_fused(x, n_samples=S, noise_steps=ns, aggregation="best", seed=seed0 + i, first_window_id=lo,
                                       out=scores[i, :B])
                if events is not None:
                    events[i][1].record()   # brackets the step's launch(es) on this stream
        for st in streams:
            main.wait_stream(st)
        if use_dist:
            gather_ev[0].record()
            flat = scores.view(-1) if coll_dev.type == "cuda" else scores.view(-1).cpu()
            gathered = torch.empty(world * flat.numel(), device=coll_dev, dtype=torch.float32)
            dist.all_gather_into_tensor(gathered, flat)   # RCCL over xGMI: the path's only exchange
            gather_ev[1].record()
            return gathered
        return scores

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    preroll_steps = 0
    est_step_ms = None
    if args.preroll_ms > 0 and B > 0:
        t_pre = time.perf_counter()
        pre_out = torch.zeros(per, device=dev, dtype=torch.float32)
        while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:      # local launches only: no collective, ranks may differ
            t_one = time.perf_counter()
            sc.score_fused(data, n_samples=S, noise_steps=ns, aggregation="best", seed=preroll_steps, first_window_id=lo, out=pre_out[:B])
            torch.cuda.synchronize()
            est_step_ms = (time.perf_counter() - t_one) * 1e3            # (the last one: clocks are up by then)
            preroll_steps += 1
    if args.min_seconds > 0:
        # a timed region of at least --min-seconds: the step count follows from the pre-roll's step time (every rank takes the
        # largest count, so the single all-gather keeps one size)
        if est_step_ms is None:
            raise SystemExit("--min-seconds needs the pre-roll (--preroll-ms > 0) to size the step count")
        want = int(np.ceil(args.min_seconds * 1e3 / est_step_ms * 1.03))
        if use_dist:
            w_t = torch.tensor([want], dtype=torch.int64, device=coll_dev)
            dist.all_reduce(w_t, op=dist.ReduceOp.MAX)
            want = int(w_t.item())
        args.steps = max(args.steps, want)
    # the timed steps' events exist (hipEventCreate, signal pool growth) before anything is timed: torch creates them lazily
    # at their first record(), which would otherwise happen inside the timed loop
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a_, b_ in ev:
        a_.record()
        b_.record()
    torch.cuda.synchronize()
    if args.warmup > 0:
        run(sc, args.warmup, 0)
    barrier()
    # Nothing but the launch loop runs on the host inside the timed region (a clock / power sampler thread here would contend
    # for the GIL with the launches and make default runs differ from --no-extras runs: ADVICE r5); the shader clock and the
    # socket power are read ONCE right behind the closing barrier, and sampled over time in the `sustained` leg only.
    t0 = time.perf_counter()
    best = run(sc, args.steps, 100, ev)
    barrier()
    dt_local = time.perf_counter() - t0
    gpu_after = None
    if rank == 0 and not args.no_extras:
        smp = GpuSampler(dev_index, 0.05)
        clk_, pw_ = smp._read()
        gpu_after = {"source": smp.source, "sclk_mhz": round(clk_, 1) if clk_ else None, "socket_power_w": round(pw_, 1) if pw_ else None,
                     "note": "one reading right behind the timed region's closing barrier (no sampler thread runs during it)"}
    dt = dt_local
    step_ms = [a.elapsed_time(b) for a, b in ev]
    kern_ms = float(np.mean(step_ms)) if B > 0 else 0.0
    rank_info = None
    if use_dist:
        gms = gather_ev[0].elapsed_time(gather_ev[1])
        # per rank, so that a slow rank is attributable: first / fastest / slowest timed step and the shader clock right after
        # the timed region (amdsmi through torch when it is importable; 0 = unknown)
        try:
            clk = float(torch.cuda.clock_rate(dev))
        except Exception:
            clk = 0.0
        sm = step_ms if step_ms else [0.0]
        mine = torch.tensor([dt_local, kern_ms, gms, float(B), sm[0], min(sm), max(sm), clk], dtype=torch.float64, device=coll_dev)
        allr = torch.empty(world * 8, dtype=torch.float64, device=coll_dev)
        dist.all_gather_into_tensor(allr, mine)
        allr = allr.view(world, 8).cpu().numpy()
        dt = float(allr[:, 0].max())          # the job's time = the slowest rank's
        rank_info = {"kernel_ms": [round(float(v), 4) for v in allr[:, 1]], "wall_s": [round(float(v), 5) for v in allr[:, 0]],
                     "all_gather_ms": [round(float(v), 4) for v in allr[:, 2]], "windows_per_step": [int(v) for v in allr[:, 3]],
                     "first_step_ms": [round(float(v), 4) for v in allr[:, 4]], "min_step_ms": [round(float(v), 4) for v in allr[:, 5]],
                     "max_step_ms": [round(float(v), 4) for v in allr[:, 6]], "clock_mhz": [int(v) for v in allr[:, 7]]}
    if streams:
        # launches on different streams overlap, so a launch's own start-to-end time counts its neighbour's work too;
        # the roofline then uses the timed region's average time per launch instead
        kern_ms = dt / args.steps * 1e3
    assert torch.isfinite(best).all()

    if rank == 0:
        total = B_total * args.steps
        P = S * (ns - 1)
        flop_per_window = P * f_unet(sc.t_unet) + (f_cond(sc.t_cond) if strat == "inject" else 0)
        achieved = B * flop_per_window / (kern_ms * 1e-3) / 1e12
        pmc = pmc_profile(args.config, B, ns, S, flop_per_window, kern_ms, sc.t_unet)
        nb = {3: "3,2,4", 6: "6,1,4", 12: "12,1,3", 4: "4,1,4", 8: "8,1,2", 5: "5,1,4", 10: "10,1,3", 7: "7,1,2", 9: "9,1,3", 11: "11,1,3", 1: "1,4,4", 2: "2,2,4"}.get(sc.t_unet)
        tiled = nb is None         # 13 .. 32 U-Net frames: the slab-tiled kernel, frame count padded to 16 (two workgroups per CU) / 24 / 32
        tp_of = lambda t: 16 if t <= 16 else 24 if t <= 24 else 32
        kname = f"score_kernel<{nb}>" if nb else "score_tiled_kernel<%d,1> (T_u=%d)" % (tp_of(sc.t_unet), sc.t_unet)
        if strat != "inject":
            enc = ""
        elif cfg.get("conditioning_architecture") == "E_unet":     # 1 .. 12 frames: LDS-resident; 13 .. 32: the slab-tiled stages (COND form)
            enc = f" + cond_unet_kernel<{sc.t_cond}>" if sc.t_cond <= 12 else " + score_tiled_kernel<%d,1,COND> (T_c=%d)" % (tp_of(sc.t_cond), sc.t_cond)
        else:                                                      # shipped encoder: MFMA up to 20 frames, the plain kernel above
            enc = f" + cond_fast_kernel<{sc.t_cond}>" if sc.t_cond <= 20 else " + cond_encode_kernel"
        split_used = sc.plan_split(B, S, ns) if B > 0 else 1         # what the library chose for this call (mcd_plan_split)
        # the shipped ('AE') encoder with as many condition frames as the U-Net has frames runs inside the one-launch kernel
        enc_inside = strat != "inject" or (cfg.get("conditioning_architecture") != "E_unet" and sc.t_cond == sc.t_unet)
        one_launch = split_used == 1 and not tiled
        launches = (1 if enc_inside else 2) if one_launch else (3 if strat == "inject" else 2)
        if args.variant:
            kname += f" variant {args.variant}"
        out = {
            "metric": f"pose-clips/sec (whole node) @ noise_steps={ns}, {S} seeds; AUC vs ref",
            "value": round(total / dt, 1), "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_ms": args.preroll_ms, "preroll_steps": preroll_steps,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{desc}, noise_steps={ns}, {S} generated samples, {strat} conditioning, 'best' aggregation",
                       "name": args.config, "windows_per_step_per_gpu": B if args.scaling == "weak" else None,
                       "windows_per_step_total": B_total, "denoiser_passes_per_window": P,
                       "weights": f"seeded random init ({'built in bench.py' if variant.startswith('rand:') else 'tests/golden/weights_' + variant + '.npz'})",
                       "noise": "in-kernel Philox4x32-10", "parallelism": f"windows sharded over {world} GPU(s), one all-gather of scores",
                       "streams": max(args.streams, 1)},
            "step_ms_median": round(float(np.median(step_ms)), 4) if B > 0 else None,
            "step_ms_pct": ({"p5": round(float(np.percentile(step_ms, 5)), 4), "p50": round(float(np.percentile(step_ms, 50)), 4),
                             "p95": round(float(np.percentile(step_ms, 95)), 4)} if B > 0 else None),
            "gpu_after_timed_region": gpu_after,
            "step_ms_first": [round(float(v), 4) for v in step_ms[:4]] if B > 0 else None,
            "step_ms_max": [round(float(np.max(step_ms)), 4), int(np.argmax(step_ms))] if B > 0 else None,    # [ms, step index]
            "roofline": {"bound": "mfma", "kernel": kname + (" (condition encoder and aggregation inside: one launch per step)" if one_launch and enc_inside
                                    else enc + " (aggregation inside the trajectory kernel: 2 launches per step)" if one_launch
                                    else enc + " + aggregate_kernel (one trajectory per workgroup: "
                                         f"{launches} launches per step)"),
                         "split": split_used, "launches_per_step": launches,
                         "achieved": round(achieved, 3),
                         "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / PEAK_FP32_TFLOPS, 4),
                         "flop_per_window": flop_per_window, "kernel_ms_per_step": round(kern_ms, 4),
                         "hbm_algorithmic_bytes_per_window": 2 * seg_len * 17 * 4 + 4,
                         # `traffic`: HBM bytes per step from the committed rocprofv3 PMC passes of this workload (FETCH_SIZE x 2,
                         # the guide's gfx950 correction, + WRITE_SIZE); null when the workload is not the profiled one
                         "traffic": pmc["hbm_bytes_per_step"] if pmc and "hbm_bytes_per_step" in pmc else None,
                         "traffic_source": pmc["source"] if pmc else None,
                         # the HBM roofline north_star asks to see beside it: those bytes / this run's launch time vs 8 TB/s
                         "hbm": ({"achieved_GBps": round(pmc["hbm_bytes_per_step"] / (kern_ms * 1e-3) / 1e9, 2), "peak_GBps": PEAK_HBM_GBS,
                                  "frac": round(pmc["hbm_bytes_per_step"] / (kern_ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 6)}
                                 if pmc and "hbm_bytes_per_step" in pmc else None),
                         # matrix-pipe statistics from the same profile: frac ~= mfma_pipe_busy_frac x useful_mfma_frac
                         "pmc": pmc},
        }
        if use_dist:
            out["ranks"] = dict(rank_info, backend="rccl" if args.dist_backend == "nccl" else args.dist_backend)
            out["rccl_ranks"] = world if args.dist_backend == "nccl" else 0
        if args.ref_value:
            out["scaling_efficiency"] = round(out["value"] / (args.ref_value * (world if args.scaling == "weak" else world)), 4)
            out["ref_value_1gpu"] = args.ref_value
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, cfg, ns, S, args.cpu_budget)
        if world == 1 and not args.no_extras and not args.no_auc:
            # `AUC vs ref` (BASELINE.json metric), untimed: see auc_vs_ref
            out["auc"] = auc_vs_ref(sc, sd, cfg, ns, S, with_oracle=not args.no_cpu_baseline and S * (ns - 1) <= 64)
        if world == 1 and not args.no_extras and args.sustained_seconds > 0 and B > 0:
            # informational only (never `value`): the same launches back to back for >= --sustained-seconds
            sus_out = torch.zeros(per, device=dev, dtype=torch.float32)
            out["sustained"] = sustained_run(lambda i: sc.score_fused(data, n_samples=S, noise_steps=ns, aggregation="best", seed=5000 + i,
                                                                      first_window_id=lo, out=sus_out[:B]),
                                             args.sustained_seconds, kern_ms, B, flop_per_window, dev_index)
        if world == 1 and not args.no_extras and args.e2e_windows != 0 and not variant.startswith("rand:"):
            # informational only: the caller's loop around the path (test_step + on_test_epoch_end), about 0.5 s of scoring by default
            n_e2e = args.e2e_windows or int(max(4 * B, min(400_000, 0.5 * out["value"])))
            fpc = 200 if seg_len <= 12 else 300
            per_clip = 3 * (fpc - 5 - seg_len + 1) * int(cfg.get("num_transform", 5))
            out["e2e"] = e2e_run(sd, cfg, ns, S, B, max(1, round(n_e2e / per_clip)), fpc, dev, out["value"])
        if world == 1 and not use_dist and not args.no_extras:
            # informational only (never `value`): the same steps fed from pinned HOST windows, the H2D copy inside the timed loop
            n_x = min(args.steps, 10)
            pinned = data_host.pin_memory()
            run(sc, 2, 300, src=pinned)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            run(sc, n_x, 400, src=pinned)
            torch.cuda.synchronize()
            out["value_incl_h2d"] = {"value": round(B * n_x / (time.perf_counter() - t1), 1), "unit": "clips/s",
                                     "note": "windows copied from pinned host memory inside the timed loop (PCIe-inclusive); informational"}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
