#!/usr/bin/env python3
"""Error study for DESIGN.md's split-bf16 lever (NOT a test, not collected by pytest; CPU only, uses the oracle).

The channel GEMMs (the 1x1 convolutions tcn.0 / residual.0 of every ST-GCN layer: 85 % of the path's MFMAs) are
re-computed with both operands split into bf16 pairs, x = hi + lo, and the products hi*hi + hi*lo + lo*hi (or fewer
terms) accumulated in fp32 -- what three (two) bf16 MFMAs per fp32 MFMA would compute -- and the window scores compared
with the golden vectors generated from the reference.  usage: python tests/studies/bf16x3_error.py"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mocodad_oracle as O  # noqa: E402

TERMS = 3
_orig = O._conv_bn


def _split(t):
    hi = t.bfloat16().float()
    return hi, (t - hi).bfloat16().float()


def _conv_bn_split(sd, conv, bn, x):
    if not (conv.endswith("tcn.0") or conv.endswith("residual.0")):
        return _orig(sd, conv, bn, x)
    w = O._t(sd, conv + ".weight")
    b = O._t(sd, conv + ".bias")
    # the kernel folds the eval-mode BatchNorm into the conv before packing: split the FOLDED weights
    g = O._t(sd, bn + ".weight") / torch.sqrt(O._t(sd, bn + ".running_var") + O.BN_EPS)
    wf = w * g[:, None, None, None]
    bf = (b - O._t(sd, bn + ".running_mean")) * g + O._t(sd, bn + ".bias")
    wh, wl = _split(wf)
    xh, xl = _split(x)
    y = F.conv2d(xh, wh)
    if TERMS >= 2:
        y = y + F.conv2d(xl, wh)
    if TERMS >= 3:
        y = y + F.conv2d(xh, wl)
    if TERMS >= 4:
        y = y + F.conv2d(xl, wl)
    return y + bf[None, :, None, None]


def run(variant, ns, S):
    d = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{variant}.npz"))
    w = {k: d[k] for k in d.files}
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    g = np.load(os.path.join(ROOT, "tests", "golden", f"traj_{variant}_ns{ns}_S{S}.npz"))
    data = torch.from_numpy(g["data"])
    noise = torch.from_numpy(g["noise"].astype(np.float32))
    with torch.no_grad():
        p, corrupt = O.reverse_diffusion(sd, data, noise, noise_steps=ns, strategy=cfg["conditioning_strategy"],
                                         conditioning_indices=cfg["conditioning_indices"])
        loss = O.window_losses(p, corrupt).t().numpy()
    ref = g["loss_all"]
    return float(np.abs(loss - ref).max()), float(np.abs(ref).max()), float(np.abs(p.transpose(0, 1).numpy() - g["poses_all"]).max())


if __name__ == "__main__":
    torch.set_num_threads(8)
    for variant, ns, S in (("inject", 10, 5), ("inject", 50, 8), ("concat", 10, 5), ("T12", 10, 2)):
        for terms in (0, 1, 2, 3, 4):
            TERMS = terms
            O._conv_bn = _orig if terms == 0 else _conv_bn_split
            e, m, pe = run(variant, ns, S)
            name = {0: "fp32 oracle", 1: "bf16 (hi*hi)", 2: "2 terms (+lo_x*hi_w)", 3: "3 terms (+hi_x*lo_w)", 4: "4 terms (all)"}[terms]
            print(f"{variant:7s} ns={ns:2d} S={S}: {name:24s} max|score - golden| = {e:.3e} (scores up to {m:.3f}), max|pose err| = {pe:.3e}")
        O._conv_bn = _orig
