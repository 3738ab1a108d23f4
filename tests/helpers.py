import argparse
import json
import os

import numpy as np
import torch

from conftest import load_golden


def golden_weights(variant):
    w = load_golden(f"weights_{variant}.npz")
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    return {k: torch.from_numpy(v) for k, v in w.items()}, cfg


def make_args(cfg, **over):
    d = dict(cfg)
    d.update(over)
    d.setdefault("gt_path", d.get("test_path"))
    d.setdefault("ckpt_dir", "/tmp/mocodad_amd_ckpt")
    return argparse.Namespace(**d)


def write_gt_dir(tmpdir, g):
    gts = {}
    for k in g:
        if k.startswith("gt_"):
            np.save(os.path.join(tmpdir, k[3:] + ".npy"), g[k])
            sc, cl = k[3:].split("_")
            gts[(int(sc), int(cl))] = g[k]
    return gts
