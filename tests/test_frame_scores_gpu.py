"""GPU: the frame-score assembly that follows the path (mcd_frame_scores: scatter-max, pad_scores, person aggregation, HR
masks, shift + Gaussian smoothing, mean over transforms -- mocodad.py:362-425, eval_utils.py:27-34,100-106,133-149) against
the reference's own AUCs (tests/golden/postproc.npz) and against the oracle's per-frame scores on synthetic clips with
absences, masks and sparse person ids."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import golden_weights, make_args, write_gt_dir

pytestmark = pytest.mark.gpu


def _gts(g):
    gts = {}
    for k in g:
        if k.startswith("gt_"):
            sc, cl = k[3:].split("_")
            gts[(int(sc), int(cl))] = g[k]
    return gts


@pytest.mark.parametrize("tag,dataset", [("avenue", "HR-Avenue"), ("stc", "HR-STC")])
def test_device_post_processing_reproduces_the_reference_auc(tmp_path, tag, dataset):
    from mocodad_amd.models.mocodad import MoCoDAD
    from oracle import mocodad_oracle as O
    g = load_golden("postproc.npz")
    write_gt_dir(str(tmp_path), g)
    pad, ks, shift = [int(v) for v in g[f"params_{tag}"]]
    _, cfg = golden_weights("inject")
    m = MoCoDAD(make_args(cfg, gt_path=str(tmp_path), dataset_choice=dataset, pad_size=pad, filter_kernel_size=ks, frames_shift=shift,
                          num_transform=5)).to("cuda:0")
    auc = m.post_processing(g["out"].copy(), None, g["trans"], g["meta"], g["frames"])
    assert abs(auc - float(g[f"auc_{tag}"])) < 1e-9, (auc, float(g[f"auc_{tag}"]))
    # the per-frame scores themselves, against the oracle's (float64 both)
    asm = m._frame_assembler()
    pds = asm(g["out"], g["trans"], g["meta"], g["frames"])
    _, ref, gt = O.post_processing(g["out"].copy(), g["trans"], g["meta"], g["frames"], _gts(g), num_transform=5,
                                   pad_size=pad, filter_kernel_size=ks, frames_shift=shift)
    assert np.array_equal(gt, asm.gt)
    np.testing.assert_allclose(pds, ref, rtol=1e-12, atol=1e-15)
    # device-resident inputs give the same numbers
    pds2 = asm(torch.from_numpy(g["out"]).cuda(), torch.from_numpy(g["trans"]).cuda(), torch.from_numpy(g["meta"]).cuda(),
               torch.from_numpy(g["frames"]).cuda())
    assert np.array_equal(pds, pds2)


@pytest.mark.parametrize("pad,sigma,shift", [(-1, 3.0, 1), (0, 2.0, 2), (1, 5.0, 3), (4, 1.0, 5), (12, 30.0, 6)])
def test_device_frame_scores_vs_oracle_with_absences_and_masks(pad, sigma, shift):
    """Random clips of different lengths, persons with sparse ids who leave and re-enter (pad_scores' intervals of absence,
    touching the first / last frames or not), HR-style keep masks, windows of unknown clips and out-of-range transforms."""
    from mocodad_amd.engine import FrameScoreAssembler
    from oracle import mocodad_oracle as O
    rng = np.random.default_rng(100 + pad + shift)
    seg_len, T = 6, 3
    clips = {(1, 3): 70, (1, 12): 45, (2, 1): 130, (10, 2): 24}
    gts = {k: (rng.random(n) < 0.3).astype(np.int64) for k, n in clips.items()}
    for g in gts.values():
        g[0], g[-1] = 0, 1
    masks_o = {(1, 12): rng.random(45) < 0.7, (2, 1): rng.random(130) < 0.8}
    rows = []
    for tr in range(T):
        for (sc, cl), n in clips.items():
            persons = sorted(rng.choice(np.arange(1, 40), size=int(rng.integers(1, 5)), replace=False))
            for pi, person in enumerate(persons):
                present = np.ones(n, dtype=bool)
                for _ in range(int(rng.integers(0, 4)) if pi else 0):      # intervals of absence (the first person stays: every
                    #                                                        (transform, clip) block needs at least one window)
                    a = int(rng.integers(0, n))
                    present[a:a + int(rng.integers(1, 15))] = False
                if pi and rng.random() < 0.3:
                    present[:int(rng.integers(1, 8))] = False
                if pi and rng.random() < 0.3:
                    present[-int(rng.integers(1, 8)):] = False
                for f0 in range(1, n - seg_len + 2):
                    if present[f0 - 1:f0 - 1 + seg_len].all():
                        rows.append((tr, sc, cl, person, f0))
    rows.append((0, 9, 9, 1, 1))            # a clip without ground truth: ignored
    rows.append((T, 1, 3, 1, 1))            # a transform index beyond num_transform: ignored
    rows = np.array(rows)
    rng.shuffle(rows)
    out = rng.gamma(2.0, 0.05, size=len(rows)).astype(np.float32)
    trans, meta = rows[:, 0].copy(), rows[:, 1:5].copy()
    frames = (rows[:, 4:5] + np.arange(seg_len)[None]).astype(np.int32)
    # every (transform, clip) needs at least one person with a window (the reference fails otherwise)
    for tr in range(T):
        for (sc, cl) in clips:
            assert ((trans == tr) & (meta[:, 0] == sc) & (meta[:, 1] == cl)).any()
    _, ref, gt = O.post_processing(out.copy(), trans, meta, frames, gts, num_transform=T, pad_size=pad, filter_kernel_size=sigma,
                                   frames_shift=shift, masks=masks_o)
    asm = FrameScoreAssembler(gts, masks_o, num_transform=T, pad_size=pad, filter_kernel_size=sigma, frames_shift=shift, device="cuda:0")
    pds = asm(out, trans, meta, frames)
    assert np.array_equal(gt, asm.gt) and pds.shape == ref.shape
    np.testing.assert_allclose(pds, ref, rtol=1e-12, atol=1e-15)


def test_device_frame_scores_error_and_fallback():
    from mocodad_amd.engine import FrameScoreAssembler
    gts = {(1, 1): np.array([0, 0, 1, 1, 0, 0, 0, 0]), (1, 2): np.array([0, 1, 0, 0, 0, 0, 0])}
    asm = FrameScoreAssembler(gts, {}, num_transform=1, pad_size=-1, filter_kernel_size=2, frames_shift=1, device="cuda:0")
    meta = np.array([[1, 1, 1, 1], [1, 1, 1, 2]])            # clip (1, 2) has no window at all
    frames = (meta[:, 3:4] + np.arange(6)[None]).astype(np.int32)
    with pytest.raises(ValueError, match="at least one array"):
        asm(np.array([0.5, 0.7], dtype=np.float32), np.zeros(2, dtype=np.int64), meta, frames)
    asm.MAX_WORKSPACE = 16              # a person-id range too large for the dense table: the caller falls back to the host
    assert asm(np.array([0.5, 0.7], dtype=np.float32), np.zeros(2, dtype=np.int64), meta, frames) is None
