"""GPU: the opt-in split-bf16 GEMM path (option 'bf16x3'; hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate)
against the golden trajectories generated from the reference.  Same tolerance as the fp32 path: 1e-4 absolute on scores;
the measured error is printed.  (The shipped default and everything bench.py reports as `value` compute in fp32.)"""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = [("inject", 2, 1), ("inject", 10, 5), ("inject", 50, 8), ("concat", 10, 5), ("concat", 50, 2), ("T12", 10, 2),
         ("T12", 50, 8)]     # 3, 6 and 12 U-Net frames


@pytest.mark.parametrize("variant,ns,S", CASES)
def test_split_bf16_gemms_vs_golden_trajectories(variant, ns, S):
    from mocodad_amd.engine import HipScorer
    from oracle import mocodad_oracle as O
    w = load_golden(f"weights_{variant}.npz")
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    strat = cfg["conditioning_strategy"]
    ci, xi = O.split_indices(cfg["seg_len"], cfg["conditioning_indices"], strat)
    sc = HipScorer(sd, strategy=strat, seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
                   cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0", options={"bf16x3": 1})
    g = load_golden(f"traj_{variant}_ns{ns}_S{S}.npz")
    loss, poses = sc.score(torch.from_numpy(g["data"]), n_samples=S, noise_steps=ns,
                           noise=torch.from_numpy(g["noise"].astype(np.float32)), want_poses=True)
    e = float(np.abs(loss.cpu().numpy() - g["loss_all"]).max())
    p = float(np.abs(poses.cpu().numpy() - g["poses_all"]).max())
    print(f"{variant} ns={ns} S={S}: max|score - golden| = {e:.3e}  max|pose - golden| = {p:.3e}")
    assert e < 1e-4
    # and it really is a different arithmetic path than the default
    sc32 = HipScorer(sd, strategy=strat, seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
                     cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0")
    l32, _ = sc32.score(torch.from_numpy(g["data"]), n_samples=S, noise_steps=ns, noise=torch.from_numpy(g["noise"].astype(np.float32)))
    assert not torch.equal(l32, loss)
