#!/usr/bin/env python3
"""Error of each arithmetic against an fp64 run of the same algorithm (NOT a test, not collected by pytest).

Reference = the oracle (the CPU restatement pinned to the reference-generated vectors) run in float64 with the same weights,
windows and noise.  Compared with it, max |score - fp64 score| of
  (i)   the oracle in fp32 (what the reference computes, torch CPU),
  (ii)  the HIP fp32 kernel (fp32 MFMA) -- the shipped, benchmarked path              [needs a GPU]
  (iii) emulations of split-bf16 channel GEMMs in the oracle: 3 terms (hi*hi + hi*lo + lo*hi -- the arithmetic of the opt-in
        kernel rounds 1-3 carried and round 4 removed: 10-20x the fp32 error on trained-scale weights) and 6 terms (three
        bf16 limbs per operand, every product whose weight is >= 2^-24: an fp32-equivalent significand)
on the golden trajectories (3, 6 and 12 U-Net frames; noise_steps 10 and 50) and on "trained-scale" variants of the same
models: BatchNorm gains spread over 0.1x..10x, PReLU slopes 0.01 / 1.5, windows pushed against the +-5 clip.

usage: python tests/studies/fp64_error.py [> profiles/r02_fp64_error.txt]"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import mocodad_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
_orig = O._conv_bn
TERMS = 0


def _limbs(t, n):
    out, r = [], t
    for _ in range(n):
        h = r.bfloat16().float()
        out.append(h)
        r = r - h
    return out


def _conv_bn_split(sd, conv, bn, x):
    """channel GEMMs (tcn.0 / residual.0, BatchNorm folded first like the kernel's packer) as sums of bf16 x bf16 products"""
    if TERMS == 0 or x.dtype != torch.float32 or not (conv.endswith("tcn.0") or conv.endswith("residual.0")):
        return _orig(sd, conv, bn, x)
    w, b = O._t(sd, conv + ".weight"), O._t(sd, conv + ".bias")
    g = O._t(sd, bn + ".weight") / torch.sqrt(O._t(sd, bn + ".running_var") + O.BN_EPS)
    wf = w * g[:, None, None, None]
    bf = (b - O._t(sd, bn + ".running_mean")) * g + O._t(sd, bn + ".bias")
    n = 2 if TERMS == 3 else 3
    ws, xs = _limbs(wf, n), _limbs(x, n)
    pairs = [(0, 0), (1, 0), (0, 1)] if TERMS == 3 else [(0, 0), (1, 0), (0, 1), (2, 0), (1, 1), (0, 2)]   # (x limb, w limb)
    y = None
    for i, j in pairs:
        t = F.conv2d(xs[i], ws[j])
        y = t if y is None else y + t
    return y + bf[None, :, None, None]


O._conv_bn = _conv_bn_split
_pe = O.pos_encoding
DTYPE = torch.float32
O.pos_encoding = lambda t, ch: _pe(t, ch).to(DTYPE)     # (the fp32 sinusoid table is part of the model: same values in the fp64 run)


def load(variant):
    d = np.load(os.path.join(ROOT, "tests", "golden", f"weights_{variant}.npz"))
    w = {k: d[k] for k in d.files}
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    return {k: torch.from_numpy(v) for k, v in w.items()}, cfg


def trained_scale(sd, cfg, seed):
    """BatchNorm gains x 10^U(-1,1), PReLU slopes alternating 0.01 / 1.5; the last layer rescaled so that the U-Net's
    eps-prediction stays O(1) (otherwise the 200x..1000x gain of the reverse chain overflows, with any arithmetic)."""
    g = torch.Generator().manual_seed(seed)
    out = {k: v.clone() for k, v in sd.items()}
    i = 0
    for k in sorted(out):
        if k.endswith(".tcn.1.weight") or k.endswith(".residual.1.weight") or k.endswith(".block.1.weight"):
            out[k] = out[k] * 10 ** (torch.rand(out[k].shape, generator=g) * 2 - 1)
        if k.endswith(".prelu.weight"):
            out[k] = torch.full_like(out[k], 0.01 if i % 2 == 0 else 1.5)
            i += 1
    Tu = cfg["seg_len"] if cfg["conditioning_strategy"] != "inject" else len(O.split_indices(cfg["seg_len"], cfg["conditioning_indices"], "inject")[1])
    for _ in range(3):
        x = torch.randn(16, 2, Tu, 17, generator=g)
        cond = torch.randn(16, 16, generator=g) * 0.5 if cfg["conditioning_strategy"] == "inject" else None
        eps = O.unet_forward(out, x, torch.full((16,), 5, dtype=torch.long), cond) - x
        s = float(eps.std())
        for n in ("tcn.0.weight", "tcn.0.bias", "residual.0.weight", "residual.0.bias", "tcn.1.bias", "residual.1.bias"):
            k = "model.st_gcnnsu3.1." + n
            if k.endswith("1.bias"):
                out[k] = out[k] * (0.5 / s)
            elif k.endswith("0.weight") or k.endswith("0.bias"):
                out[k] = out[k] * (0.5 / s)
        for n in ("tcn.1.running_mean", "residual.1.running_mean"):
            out["model.st_gcnnsu3.1." + n] = out["model.st_gcnnsu3.1." + n] * (0.5 / s)
    return out


def scores(sd, cfg, data, noise, ns, dtype):
    global DTYPE
    DTYPE = dtype
    sdd = {k: (v.to(dtype) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    p, corrupt = O.reverse_diffusion(sdd, data.to(dtype), noise.to(dtype), noise_steps=ns, strategy=cfg["conditioning_strategy"],
                                     conditioning_indices=cfg["conditioning_indices"])
    return O.window_losses(p, corrupt).t().double().numpy()


def hip_scores(sd, cfg, data, noise, ns, S):
    from mocodad_amd.engine import HipScorer
    strat = cfg["conditioning_strategy"]
    ci, xi = O.split_indices(cfg["seg_len"], cfg["conditioning_indices"], strat)
    sc = HipScorer(sd, strategy=strat, seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
                   cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0")
    return sc.score(data, n_samples=S, noise_steps=ns, noise=noise)[0].double().cpu().numpy()


def main():
    global TERMS
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    gpu = torch.cuda.is_available()
    print("# max |score - fp64 score| per case (scores are O(0.1 .. 2)); reference = oracle in float64, same weights / windows / noise")
    print(f"# HIP columns: {'measured on ' + torch.cuda.get_device_name(0) if gpu else 'no GPU in this run'}")
    hdr = f"{'case':34s} {'fp32 oracle':>12s} {'HIP fp32':>12s} {'emul 3 terms':>13s} {'emul 6 terms':>13s}  max score"
    print(hdr)
    cases = [("inject", 10, 5), ("inject", 50, 8), ("concat", 10, 5), ("concat", 50, 2), ("T12", 10, 2), ("T12", 50, 8)]
    for scale in (False, True):
        for variant, ns, S in cases:
            sd, cfg = load(variant)
            g = np.load(os.path.join(ROOT, "tests", "golden", f"traj_{variant}_ns{ns}_S{S}.npz"))
            data = torch.from_numpy(g["data"])
            noise = torch.from_numpy(g["noise"].astype(np.float32))
            if scale:
                sd = trained_scale(sd, cfg, 11)
                data = (data * 2.5).clamp(-5, 5)
            TERMS = 0
            ref = scores(sd, cfg, data, noise, ns, torch.float64)
            cols = [np.abs(scores(sd, cfg, data, noise, ns, torch.float32) - ref).max()]
            cols.append(np.abs(hip_scores(sd, cfg, data, noise, ns, S) - ref).max() if gpu else float("nan"))
            for t in (3, 6):
                TERMS = t
                cols.append(np.abs(scores(sd, cfg, data, noise, ns, torch.float32) - ref).max())
            TERMS = 0
            name = f"{variant} ns={ns} S={S}" + (" trained-scale" if scale else "")
            print(f"{name:34s} " + " ".join(f"{c:12.3e}" for c in cols[:2]) + " " + " ".join(f"{c:13.3e}" for c in cols[2:]) + f"  {np.abs(ref).max():8.3f}")


if __name__ == "__main__":
    main()
