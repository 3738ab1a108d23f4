#!/usr/bin/env python3
"""Per-stage cycle breakdown of workgroup 0 of score_tiled_kernel (needs a -DMCD_PROFILE build: MCD_PROF_LIB or
mocodad_amd/libmocodad_hip_prof.so).   usage (GPU box): python tools/tiled_stage_profile.py [config]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mocodad_amd import _lib
_lib.LIB_PATH = os.environ.get("MCD_PROF_LIB") or os.path.join(ROOT, "mocodad_amd", "libmocodad_hip_prof.so")
import bench
from mocodad_amd.engine import HipScorer

CONFIG = sys.argv[1] if len(sys.argv) > 1 else "concat32"
variant_w, B, NS, S, _ = bench.CONFIGS[CONFIG]
sd, cfg = bench.load_weights(variant_w)
ci, xi = bench.frame_split(cfg["seg_len"], cfg["conditioning_indices"], cfg["conditioning_strategy"])
sc = HipScorer(sd, strategy=cfg["conditioning_strategy"], seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
               cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0")
L = _lib.lib()
prof = torch.zeros(4096 + 8192 + 512, dtype=torch.int64, device="cuda:0")
data = bench.synth_windows(B, cfg["seg_len"], 1).cuda()
sc.score(data, n_samples=S, noise_steps=NS, seed=1)
torch.cuda.synchronize()
L.mcd_debug_set_prof(C.c_void_p(prof.data_ptr()))
sc.score(data, n_samples=S, noise_steps=NS, seed=1)
torch.cuda.synchronize()
L.mcd_debug_set_prof(None)
pall = prof.cpu().numpy().astype(float)
for wv, off in ((0, 2048), (7, 2048 + 64)):
    p = pall[off:off + 64]
    tot = p.sum()
    print(f"{CONFIG}: T_u={sc.t_unet}  workgroup 0, wave {wv}: {tot:.0f} cycles over the launch, {tot / 2.4e6:.2f} ms at 2.4 GHz")
    lname = ["L0", "L1", "L2", "L3", "L4", "L5", "L6", "L7", "L8", "L9"]      # (a layer's row sums its 32-channel parts and frame groups)
    print("  layer  X staged       mix   barrier  gemm+epi   (% of the launch)     [X staged: from the layer's / part's start to its X in LDS,")
    print("                                                                         incl. the wait for the other waves' previous stage]")
    for l in range(10):
        v = p[4 * l:4 * l + 4].copy()
        if l >= 3:
            v[0] += p[40 + 2 * (l - 3):42 + 2 * (l - 3)].sum()      # (X staged = its own slot + the two sub-slots below)
        print(f"  {lname[l]:4s} {v[0]:10.0f} {v[1]:9.0f} {v[2]:9.0f} {v[3]:9.0f}   {100 * v.sum() / tot:5.1f}%")
    print("  of the X staging of the later parts / chunks / skip rows:  wait for the other waves' previous stage | loads issued a stage ahead landing + LDS stores")
    for l in range(3, 10):
        v = p[40 + 2 * (l - 3):42 + 2 * (l - 3)]
        print(f"  {lname[l]:4s} {v[0]:10.0f} {v[1]:9.0f}   {100 * v.sum() / tot:5.1f}%")
    for i, n in [(55, "L10 product"), (56, "L10 mix"), (54, "L10 tail+update"), (58, "prologue: SiLU, noise"), (60, "prologue: emb rows"), (61, "layer tails")]:
        print(f"  {n:22s} {p[i]:12.0f}   {100 * p[i] / tot:5.1f}%")

# per-wave event trace of one pass (profile builds of score_tiled_kernel stamp it, see TLTR): for every event the spread of the
# waves' arrival, relative to the pass's first stamp
NW = int(os.environ.get("MCD_TRACE_WAVES", "12"))
ids = pall[4096 + 8192:4096 + 8192 + 512].astype(int)
ev = pall[4096:4096 + 8192].reshape(512, 16)[:, :NW]
n = int((ev[:, 0] > 0).sum())
if n:
    t0 = ev[0].min()
    names = {}
    for l in range(10):
        names.update({4 * l: f"L{l} X ready", 4 * l + 1: f"L{l} joint mix done", 4 * l + 2: f"L{l} barrier", 4 * l + 3: f"L{l} gemm(+epi) done",
                      100 + l: f"L{l} time mix done", 120 + l: f"L{l} barrier", 140 + l: f"L{l} gemm done", 160 + l: f"L{l} barrier (epi)", 180 + l: f"L{l} X staged, pre-barrier"})
        if l >= 3:
            names.update({40 + 2 * (l - 3): f"L{l} prev stage done (barrier)", 41 + 2 * (l - 3): f"L{l} loads committed"})
    names.update({54: "L10 + update", 60: "pass prologue", 61: "layer tails"})
    print(f"trace of one pass, {NW} waves: event, cycles since the pass's first stamp (earliest wave), then per wave its lag behind the earliest")
    prev = t0
    for e in range(n):
        r = ev[e]
        print(f"  {e:3d} {names.get(ids[e], str(ids[e])):32s} {r.min() - t0:9.0f} (+{r.min() - prev:7.0f})  " + " ".join(f"{x - r.min():6.0f}" for x in r))
        prev = r.min()
