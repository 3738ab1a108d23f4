#!/usr/bin/env python3
"""Throughput of the non-headline configurations (BASELINE configs 4/5 shapes) — parity-test cases, not bench lines.
usage (GPU box): python tools/perf_configs.py"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mocodad_amd.engine import HipScorer  # noqa: E402

F_UNET = {3: 4_290_352, 6: 8_799_400, 12: 18_524_464}
F_COND = {3: 545_904, 12: 2_484_720}


def split(seg_len, ci, strat):
    if strat == "no_condition":
        return [], list(range(seg_len))
    if isinstance(ci, int):
        n = seg_len // ci
        return list(range(n)), list(range(n, seg_len))
    return list(ci), [i for i in range(seg_len) if i not in ci]


for variant, B, ns, S in (("inject", 1024, 10, 5), ("concat", 1024, 10, 5), ("T12", 512, 10, 5), ("T12", 256, 50, 8)):
    d = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{variant}.npz"))
    w = {k: d[k] for k in d.files}
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    strat = cfg["conditioning_strategy"]
    ci, xi = split(cfg["seg_len"], cfg["conditioning_indices"], strat)
    sc = HipScorer(sd, strategy=strat, seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
                   cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0")
    data = torch.randn(B, 2, cfg["seg_len"], 17).clamp_(-5, 5).cuda()
    for _ in range(2):
        sc.score(data, n_samples=S, noise_steps=ns, seed=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for i in range(reps):
        sc.score(data, n_samples=S, noise_steps=ns, seed=i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    flop = S * (ns - 1) * F_UNET[sc.t_unet] + (F_COND.get(sc.t_cond, 0) if strat == "inject" else 0)
    print(f"{variant:8s} T_u={sc.t_unet:2d} B={B} ns={ns} S={S}: {B/dt:10.0f} clips/s  {B*flop/dt/1e12:6.1f} TFLOP/s ({B*flop/dt/157.3e12*100:4.1f}% of fp32 peak)  {dt*1e3:.2f} ms")
