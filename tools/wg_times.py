#!/usr/bin/env python3
"""Per-workgroup start / end times of one scoring launch (needs a -DMCD_PROFILE build): how the trajectories of a launch
spread over the CUs in time.    usage (GPU box): MCD_PROF_LIB=... python tools/wg_times.py <split> [config]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("MCD_PROF_LIB") or os.path.join(ROOT, "mocodad_amd", "libmocodad_hip_prof.so")
import torch
from mocodad_amd import _lib
_lib.LIB_PATH = so
import bench
from mocodad_amd.engine import HipScorer

SPLIT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
CONFIG = sys.argv[2] if len(sys.argv) > 2 else "avenue"
variant_w, B, NS, S, _ = bench.CONFIGS[CONFIG]
sd, cfg = bench.load_weights(variant_w)
ci, xi = bench.frame_split(cfg["seg_len"], cfg["conditioning_indices"], cfg["conditioning_strategy"])
sc = HipScorer(sd, strategy=cfg["conditioning_strategy"], seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
               cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0", options={"split": SPLIT})
L = _lib.lib()
prof = torch.zeros(4096 + 3 * 6000, dtype=torch.int64, device="cuda:0")
data = bench.synth_windows(B, cfg["seg_len"], 1).cuda()
for _ in range(20):
    sc.score_fused(data, n_samples=S, noise_steps=NS, seed=1)
torch.cuda.synchronize()
L.mcd_debug_set_prof(C.c_void_p(prof.data_ptr()))
sc.score_fused(data, n_samples=S, noise_steps=NS, seed=1)
torch.cuda.synchronize()
p = prof.cpu().numpy()[4096:].reshape(-1, 3)
n = int((p[:, 0] != 0).sum())
t0 = p[:n, 0].astype(np.int64)
t1 = p[:n, 1].astype(np.int64)
hw = p[:n, 2]
cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 8) & 0xf) + 16 * ((hw >> 12) & 0x1) + 32 * ((hw >> 13) & 0x7)      # xcc, cu id, sh id, se id
tg = (hw >> 16) & 0xf
print("  HW_ID.TG_ID histogram:", np.bincount(tg.astype(np.int64), minlength=4)[:8], " WAVE_ID of wave 0:", np.bincount((hw & 0xf).astype(np.int64), minlength=4)[:10])
base = t0.min()
dur = (t1 - t0) / 100.0          # us
print(f"split={SPLIT}: {n} workgroups, launch span {(t1.max() - base) / 100.0:.1f} us")
print(f"  workgroup duration us: mean {dur.mean():.1f}  min {dur.min():.1f}  p5 {np.percentile(dur, 5):.1f}  p50 {np.percentile(dur, 50):.1f}  p95 {np.percentile(dur, 95):.1f}  max {dur.max():.1f}")
print(f"  start us: p50 {np.percentile(t0 - base, 50) / 100:.1f} max {(t0.max() - base) / 100:.1f};  end us: min {(t1.min() - base) / 100:.1f} p50 {np.percentile(t1 - base, 50) / 100:.1f} max {(t1.max() - base) / 100:.1f}")
cus = {}
for c, d, e in zip(cu, dur, t1):
    cus.setdefault(int(c), []).append((d, e))
per = np.array([len(v) for v in cus.values()])
last = np.array([max(e for _, e in v) - base for v in cus.values()]) / 100.0
print(f"  {len(cus)} CUs seen; workgroups per CU: min {per.min()} max {per.max()}; last end per CU us: min {last.min():.1f} p50 {np.percentile(last, 50):.1f} max {last.max():.1f}")
xs = {}
for c, e in zip(cu, t1):
    xs.setdefault(int(c) // 1000, []).append((e - base) / 100.0)
print("  last end per XCD us: " + " ".join(f"{k}:{max(v):.0f}" for k, v in sorted(xs.items())))
