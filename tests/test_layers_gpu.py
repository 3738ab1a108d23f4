"""GPU: every stage of the U-Net ALONE through the test entry mcd_layer_forward -- the production MFMA stage functions in
the trajectory kernel's LDS plan -- against the reference's own layer I/O (tests/golden/layers_{inject,concat,T12,hostile_*}.npz):
ST_GCNN_layer.forward (stsgcn.py:94-116) for the 11 layers, CNN_layer over the joint axis (stsgcn.py:187-199 as called at
stsae_unet.py:205,213,381,391) for the 4 resamplers.  3 U-Net frames (inject), 6 (concat) and 12 (T12)."""
import json

import numpy as np
import pytest
import torch
from mocodad_amd import _lib

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
        assert _lib.lib().mcd_debug_poison_lds(None) == 0      # (LDS full of NaN patterns: no stage entry may read what it did not write)
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


@pytest.mark.parametrize("seg_len", [18, 22, 24, 20])
def test_every_stage_at_twelve_wave_frame_counts_vs_oracle(seg_len):
    """9 and 11 U-Net frames (seg_len 18 / 22 split in halves) have no reference-generated stage fixture: every stage of their
    TWELVE-wave kernels alone (mcd_layer_forward runs score_kernel<9|10|11|12, 1, 3, LT> built with the flags of the shipped
    kernel) against the ORACLE's stage functions, which are pinned to the reference's layer I/O at 3 .. 32 frames
    (test_oracle_golden.py::test_layers).  10 and 12 frames ride along (they have reference fixtures too: test_extra6_gpu.py and above)."""
    from oracle import mocodad_oracle as O
    from test_hip_parity import _random_model
    m, sd, gen = _random_model("inject", seg_len, 2)
    sc = m.to("cuda:0").scorer()
    T = seg_len // 2
    assert sc.t_unet == T
    sdo = O.to_torch_state({k: v.numpy() for k, v in sd.items()})
    B = 3
    e = torch.randn(B, 16, generator=gen)
    blocks = O.UNET_DOWN + O.UNET_MID1 + O.UNET_MID2 + O.UNET_UP4 + O.UNET_UP3
    worst = 0.0
    with torch.no_grad():
        for i, (b, li) in enumerate(blocks):
            cin, vin, _, _ = sc._STAGES[i]
            x = torch.randn(B, cin, T, vin, generator=gen)
            ref = O.st_gcnn_layer(sdo, f"model.{b}.{li}", x, e).numpy()
            assert _lib.lib().mcd_debug_poison_lds(None) == 0
            out = sc.layer_forward(i, x, e).cpu().numpy()
            scale = max(1.0, float(np.abs(ref).max()))
            worst = max(worst, np.abs(out - ref).max() / scale)
            np.testing.assert_allclose(out, ref, atol=2e-5 * scale, rtol=1e-5, err_msg=f"T_u {T} layer {i}")
        for sid, rn in ((11, "down1"), (12, "down2"), (13, "up3"), (14, "up2")):
            cin, vin, _, _ = sc._STAGES[sid]
            x = torch.randn(B, cin, T, vin, generator=gen)
            ref = O.joint_resample(sdo, f"model.{rn}", x).numpy()
            out = sc.layer_forward(sid, x, e).cpu().numpy()
            scale = max(1.0, float(np.abs(ref).max()))
            worst = max(worst, np.abs(out - ref).max() / scale)
            np.testing.assert_allclose(out, ref, atol=2e-5 * scale, rtol=1e-5, err_msg=f"T_u {T} {rn}")
    print(f"T_u = {T}: max scaled |stage output - oracle| over the 15 stages = {worst:.3e}")
