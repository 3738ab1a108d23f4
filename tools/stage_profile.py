#!/usr/bin/env python3
"""Per-stage cycle breakdown of one workgroup of score_kernel (needs a -DMCD_PROFILE build).
usage (GPU box): python tools/stage_profile.py [variant]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.environ.get("MCD_PROF_LIB") or os.path.join(ROOT, "mocodad_amd", "libmocodad_hip_prof.so")
if not os.environ.get("MCD_PROF_LIB"):
    from mocodad_amd import build as _build
    _build.build_library(so, ["MCD_PROFILE"] + (["MCD_TUNING_VARIANTS"] if len(sys.argv) > 1 and sys.argv[1] != "0" else []))
VARIANT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
import torch
from mocodad_amd import _lib
_lib.LIB_PATH = so
import bench
from mocodad_amd.engine import HipScorer

CONFIG = sys.argv[2] if len(sys.argv) > 2 else "avenue"
variant_w, B, NS, S, _ = bench.CONFIGS[CONFIG]
sd, cfg = bench.load_weights(variant_w)
ci, xi = bench.frame_split(cfg["seg_len"], cfg["conditioning_indices"], cfg["conditioning_strategy"])
sc = HipScorer(sd, strategy=cfg["conditioning_strategy"], seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
               cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0", options={"variant": VARIANT, "split": S})
B = min(B, 1024)
NS = min(NS, 10)
L = _lib.lib()
prof = torch.zeros(4096, dtype=torch.int64, device="cuda:0")
data = bench.synth_windows(B, cfg["seg_len"], 1).cuda()
sc.score(data, n_samples=S, noise_steps=NS, seed=1)
torch.cuda.synchronize()
L.mcd_debug_set_prof(C.c_void_p(prof.data_ptr()))
sc.score(data, n_samples=S, noise_steps=NS, seed=1)
torch.cuda.synchronize()
p = prof.cpu().numpy().astype(float)
names = ["pass prologue", "-", "L0 sp1a", "L1 sd1.0", "L2 sd1.1", "down1", "L3 sd2.0", "L4 sd2.1", "down2", "L5 sd3.0",
         "L6 gemm(P)", "L6 mix+epi(+up3)", "up3+skip", "L7 su4.0", "L8 su4.1", "up2+skip", "L9 su3.0", "L10+ddpm"]
sub = p[32:32 + 33].reshape(11, 3)   # per layer: mix, gemm (epilogue time is accounted to the stage ids below)
for l in range(11):
    if sub[l, :2].sum() > 0:
        idx = {0: 2, 1: 3, 2: 4, 3: 6, 4: 7, 5: 9, 7: 13, 8: 14, 9: 16}[l]
        p[idx] += sub[l, 0] + sub[l, 1]
sub10 = p[18:22].copy()        # layer 10: FMA product | x-block zeroing + next pass's embeddings | barrier | mix + DDPM store (then barrier = p[17])
p[17] += sub10.sum()
NP = NS - 1
tot = p[:18].sum()
print(f"variant={VARIANT}  cycles per pass ({NP} passes): total {tot/NP:.0f}")
lay = {2: 0, 3: 1, 4: 2, 6: 3, 7: 4, 9: 5, 13: 7, 14: 8, 16: 9}
for i, (n, v) in enumerate(zip(names, p)):
    extra = ""
    if i in lay:
        l = lay[i]
        extra = f"   mix {sub[l,0]/NP:7.0f}  gemm {sub[l,1]/NP:7.0f}  epilogue {(v - sub[l,0] - sub[l,1])/NP:7.0f}"
    if i == 17:
        extra = f"   product {sub10[0]/NP:6.0f}  zero+emb {sub10[1]/NP:6.0f}  barrier {sub10[2]/NP:6.0f}  mix+barrier+tail {sub10[3]/NP:6.0f}  barrier {(v - sub10.sum())/NP:6.0f}"
    print(f"  {n:14s} {v/NP:9.0f}  {100*v/tot:5.1f}%{extra}")

# per-wave cycles spent waiting at each barrier of a pass (the wave with the smallest wait arrived last: the stage's critical path)
# PROF_NW of the build = the wave count of the unit that holds this frame count's kernel (MCD_UNIT_FLAGS_<n> of mcd_instances.hpp)
from mocodad_amd import build as _b
_sh = _b.shipped_shape(sc.t_unet)
_nw = [f for f in (_b.unit_flags().get(_sh[0], []) if _sh else []) if f.startswith("-DMCD_NWAVES=")]
NW = int(os.environ.get("MCD_NWAVES") or (_nw[0].split("=")[1] if _nw else 8))
bar = p[72 + NW:72 + NW + 30 * NW].reshape(30, NW) / NP
if bar.sum() > 0:
    print(f"\nbarrier waits per pass (cycles), {NW} waves: barrier | per wave | min  mean")
    for i in range(30):
        if bar[i].sum() > 0:
            print(f"  {i:2d} | " + " ".join(f"{x:6.0f}" for x in bar[i]) + f" | {bar[i].min():6.0f} {bar[i].mean():6.0f}")
    print(f"  sum of per-barrier minima {bar.min(1).sum():.0f}   mean wait per wave {bar.sum(0).mean():.0f}   (pass total {tot/NP:.0f})")

# per-wave time stamps of one pass (the second pass of workgroup 0's first trajectory), relative to the top of the pass:
# per layer: entry | mix done (before its barrier) | after the barrier | GEMM starts (next stage's coefficient loads issued) |
#            GEMM issued | stores landed | after the closing barrier
PROF_SLOTS = 72 + NW + 30 * NW
tr = p[PROF_SLOTS:PROF_SLOTS + 128 * NW].reshape(128, NW)
if tr[0].sum() > 0:
    t0 = tr[0].min()
    tr = (tr - t0) % 2.0 ** 32 + t0          # (32-bit stamps)
    print("\ntime stamps of one pass (cycles from the top of the pass), per wave; rows: layer / point")
    pts = ["entry", "mix done", "barrier", "gemm start", "gemm issued", "stores landed", "barrier"]
    for l in range(11):
        for k in range(7):
            row = tr[8 + 8 * l + k]
            if row.sum() > 0:
                print(f"  L{l:<2d} {pts[k]:14s} " + " ".join(f"{x - t0:7.0f}" for x in row) + f" | span {row.max() - row.min():6.0f}")
