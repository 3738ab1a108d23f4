"""GPU: every stage of the U-Net ALONE through the test entry mcd_layer_forward -- the production MFMA stage functions in
the trajectory kernel's LDS plan -- against the reference's own layer I/O (tests/golden/layers_{inject,concat,T12,hostile_*}.npz):
ST_GCNN_layer.forward (stsgcn.py:94-116) for the 11 layers, CNN_layer over the joint axis (stsgcn.py:187-199 as called at
stsae_unet.py:205,213,381,391) for the 4 resamplers.  3 U-Net frames (inject), 6 (concat) and 12 (T12)."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _scorer(variant):
    from mocodad_amd.engine import HipScorer
    from oracle import mocodad_oracle as O
    w = load_golden(f"weights_{variant}.npz")
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    strat = cfg["conditioning_strategy"]
    ci, xi = O.split_indices(cfg["seg_len"], cfg["conditioning_indices"], strat)
    return HipScorer(sd, strategy=strat, seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
                     cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0")


@pytest.mark.parametrize("variant", ["inject", "concat", "T12", "hostile_inject", "hostile_concat", "hostile_T12"])
def test_every_layer_vs_reference_layer_io(variant):
    """3, 6 and 12 U-Net frames (the 12-frame kernel has its own mix path, six output frames per unit); the `hostile_*` fixtures
    hold trained-scale weight statistics (BatchNorm gains 0.1x..10x, PReLU slopes 1.5 / -0.2 / 0.01 / 0).  Bound: 2e-5,
    relative to the stage's largest output where that exceeds 1."""
    sc = _scorer(variant)
    g = load_golden(f"layers_{variant}.npz")
    e = torch.from_numpy(g["emb_in"])
    worst = 0.0
    for i in range(11):
        out = sc.layer_forward(i, torch.from_numpy(g[f"L{i}_in"]), e).cpu().numpy()
        ref = g[f"L{i}_out"]
        scale = max(1.0, float(np.abs(ref).max()))
        worst = max(worst, np.abs(out - ref).max() / scale)
        np.testing.assert_allclose(out, ref, atol=2e-5 * scale, rtol=1e-5, err_msg=f"layer {i}")
    for sid, rn in ((11, "down1"), (12, "down2"), (13, "up3"), (14, "up2")):
        out = sc.layer_forward(sid, torch.from_numpy(g[f"{rn}_in"]), e).cpu().numpy()
        ref = g[f"{rn}_out"]
        scale = max(1.0, float(np.abs(ref).max()))
        worst = max(worst, np.abs(out - ref).max() / scale)
        np.testing.assert_allclose(out, ref, atol=2e-5 * scale, rtol=1e-5, err_msg=rn)
    print(f"{variant}: max scaled |stage output - reference| over the 15 stages = {worst:.3e}")


def test_layer_forward_ragged_batch_and_errors():
    sc = _scorer("inject")
    g = load_golden("layers_inject.npz")
    e = torch.from_numpy(g["emb_in"])
    x = torch.from_numpy(g["L5_in"])
    full = sc.layer_forward(5, x, e)
    part = sc.layer_forward(5, x[:3], e[:3])          # 3 windows: the second workgroup's second chain slot is empty
    assert torch.equal(full[:3], part)
    with pytest.raises(ValueError):
        sc.layer_forward(5, x[:, :10], e)
    with pytest.raises(KeyError):
        sc.layer_forward(15, x, e)
