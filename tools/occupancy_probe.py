#!/usr/bin/env python3
"""Launch time of the headline configuration vs. the number of workgroups (1 vs. 2 resident per CU): how much of one
workgroup's non-MFMA time a co-resident one fills.  usage (GPU box): python tools/occupancy_probe.py"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mocodad_amd.engine import HipScorer
d = np.load(os.path.join(ROOT, "tests", "golden", "weights_inject.npz"))
w = {k: d[k] for k in d.files}
cfg = json.loads(bytes(w.pop("__cfg__")).decode())
sd = {k: torch.from_numpy(v) for k, v in w.items()}
sc = HipScorer(sd, strategy="inject", seg_len=6, cond_idx=[0,1,2], corrupt_idx=[3,4,5], cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0",
               options={"split": 5})      # chain-major: one trajectory pair per workgroup, as when the probe was written
for B in (51, 102, 153, 204, 256, 307, 408, 1024):
    data = torch.randn(B, 2, 6, 17).clamp_(-5, 5).cuda()
    for _ in range(3): sc.score(data, n_samples=5, noise_steps=10, seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): sc.score(data, n_samples=5, noise_steps=10, seed=i)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"B={B} WGs={(B*5+1)//2} {ms*1e3:.0f} us  {B/ms:.0f} clips/ms")
