"""CPU: host logic of the drop-in surface (no GPU needed): state_dict layout, schedules, post-processing,
collation, error behaviour, and the C-ABI library's exported symbols."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from helpers import golden_weights, make_args, write_gt_dir
from mocodad_amd.models.mocodad import MoCoDAD
from mocodad_amd.utils import diffusion_utils as DU
from mocodad_amd.utils import eval_utils as EU
from mocodad_amd.utils.model_utils import processing_data
from oracle import mocodad_oracle as O


@pytest.mark.parametrize("variant", ["inject", "concat", "T12", "injtail", "nocond", "encE", "encU", "imp2", "implist", "rndimp"])
def test_state_dict_layout_matches_reference_checkpoint(variant):
    sd, cfg = golden_weights(variant)
    m = MoCoDAD(make_args(cfg))
    own = m.state_dict()
    assert set(own.keys()) == set(sd.keys())
    for k in sd:
        assert tuple(own[k].shape) == tuple(sd[k].shape), k
    m.load_state_dict(sd)           # a Lightning ckpt's 'state_dict' loads verbatim
    assert torch.equal(m.state_dict()["model.st_gcnnsd3.0.gcn.A"], sd["model.st_gcnnsd3.0.gcn.A"])
    assert m.n_frames_condition + m.n_frames_corrupt == cfg["seg_len"]


@pytest.mark.parametrize("ns", [2, 10, 50])
def test_diffusion_tables_bit_exact(ns):
    g = load_golden("schedule.npz")
    d = DU.Diffusion(noise_steps=ns, device="cpu")
    assert np.array_equal(d.beta.numpy(), g[f"beta_{ns}"])
    assert np.array_equal(d.alpha.numpy(), g[f"alpha_{ns}"])
    assert np.array_equal(d.alpha_hat.numpy(), g[f"alpha_hat_{ns}"])
    tab = DU.step_table(ns, 16)
    b, a, ah = O.schedule(ns)
    assert torch.equal(tab[:, 0], 1 / torch.sqrt(a)) and torch.equal(tab[:, 1], (1 - a) / torch.sqrt(1 - ah))
    assert torch.equal(tab[:, 2], torch.sqrt(b))
    pe = O.pos_encoding(torch.arange(ns, dtype=torch.float32)[:, None], 16)
    assert torch.equal(tab[:, 4:], pe)


def test_diffusion_api_surface():
    d = DU.Diffusion(noise_steps=10, device="cpu")
    x = torch.randn(5, 2, 3, 17)
    t = d.sample_timesteps(5)
    assert t.min() >= 1 and t.max() < 10
    for fn in (d.noise_images, d.noise_graph):
        xt, eps = fn(x, t)
        ah = d.alpha_hat[t][:, None, None, None]
        assert torch.allclose(xt, torch.sqrt(ah) * x + torch.sqrt(1 - ah) * eps)
    z = torch.randn(5, 8)
    zt, eps = d.noise_latent(z, t)
    assert zt.shape == z.shape and d.prepare_noise_schedule().shape == (10,)
    bet = DU.betas_for_alpha_bar(10, DU._cosine_alpha_bar)
    assert np.array_equal(bet, O.betas_for_alpha_bar(10))


@pytest.mark.parametrize("tag,dataset", [("avenue", "HR-Avenue"), ("stc", "HR-STC")])
def test_post_processing_auc_matches_reference(tmp_path, tag, dataset):
    g = load_golden("postproc.npz")
    write_gt_dir(str(tmp_path), g)
    _, cfg = golden_weights("inject")
    pad, ks, shift = [int(v) for v in g[f"params_{tag}"]]
    m = MoCoDAD(make_args(cfg, gt_path=str(tmp_path), dataset_choice=dataset, pad_size=pad, filter_kernel_size=ks,
                          frames_shift=shift, num_transform=5))
    auc = m.post_processing(g["out"].copy(), None, g["trans"], g["meta"], g["frames"])
    assert abs(auc - float(g[f"auc_{tag}"])) < 1e-9


def test_frame_assembly_matches_oracle_pieces():
    rng = np.random.default_rng(0)
    for _ in range(20):
        n = int(rng.integers(20, 60))
        gt = rng.integers(0, 2, n)
        score = rng.random(n) * (rng.random(n) > 0.4)
        a = EU.pad_scores(score.copy(), gt, 5)
        b = O.pad_scores(score.copy(), gt, 5)
        assert np.array_equal(a, b)
    pos = rng.random(7)
    fr = (rng.integers(1, 30, 7)[:, None] + np.arange(6)[None]).astype(np.int32)
    mat = EU.compute_var_matrix(pos, fr, 40)
    ref = np.zeros((7, 40))
    for i in range(7):
        ref[i, fr[i] - 1] = pos[i]
    assert np.array_equal(mat, ref)
    assert len(EU.get_avenue_mask()[1]) == 1439 and len(EU.get_avenue_mask()[16]) == 740


def test_processing_data_concatenates_batches():
    a = [torch.arange(3.0), torch.zeros(3, 2, 6, 17), torch.arange(3), torch.zeros(3, 4, dtype=torch.long), torch.zeros(3, 6, dtype=torch.int32)]
    b = [torch.arange(2.0), torch.ones(2, 2, 6, 17), torch.arange(2), torch.ones(2, 4, dtype=torch.long), torch.ones(2, 6, dtype=torch.int32)]
    out, gt, tr, meta, fr = processing_data([a, b])
    ro = O.processing_data([a, b])
    for x, y in zip((out, gt, tr, meta, fr), ro):
        assert np.array_equal(x, y)


def test_frame_splits_and_random_imp_masks_follow_the_reference():
    """_frame_split for the imputation strategies, and the per-window frame sets of 'random_imp': the module draws one
    torch.randperm per window on the default generator like _select_frames does (golden: sets the reference drew)."""
    from conftest import load_golden
    _, cfg = golden_weights("imp2")
    assert MoCoDAD(make_args(cfg))._frame_split() == ([0, 2, 4], [1, 3, 5])
    _, cfg = golden_weights("implist")
    assert MoCoDAD(make_args(cfg))._frame_split() == ([1, 4], [0, 2, 3, 5])
    _, cfg = golden_weights("rndimp")
    m = MoCoDAD(make_args(cfg))
    assert (m.n_frames_condition, m.n_frames_corrupt, m.input_n_frames) == (2, 4, 6)
    g = load_golden("traj_rndimp_ns4_S2.npz")
    torch.manual_seed(int(g["rng_seed"][0]))
    mask = m.draw_random_imp_mask(g["data"].shape[0])
    assert torch.equal(mask, torch.from_numpy(g["cond_mask"]))
    assert all(bin(int(v)).count("1") == 2 for v in mask)


def test_error_behaviour():
    _, cfg = golden_weights("inject")
    with pytest.raises(NotImplementedError):
        MoCoDAD(make_args(cfg, conditioning_architecture="bogus"))
    with pytest.raises(KeyError):
        MoCoDAD(make_args(cfg, conditioning_strategy="bogus"))
    with pytest.raises(AssertionError):
        MoCoDAD(make_args(cfg, conditioning_indices=[0, 2, 3]))
    with pytest.raises(AssertionError):
        MoCoDAD(make_args(cfg, conditioning_indices=[1, 2, 3]))
    m = MoCoDAD(make_args(cfg))
    batch = [torch.zeros(2, 2, 6, 17), torch.zeros(2), torch.zeros(2, 4), torch.zeros(2, 6)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.forward(batch)          # the product path never silently falls back to the CPU
    with pytest.raises(NotImplementedError):
        m.training_step(batch, 0)
    assert m._frame_split() == ([0, 1, 2], [3, 4, 5])
    m2 = MoCoDAD(make_args(cfg, conditioning_indices=2, seg_len=24))
    assert (m2.n_frames_condition, m2.n_frames_corrupt, m2.input_n_frames) == (12, 12, 12)
    m3 = MoCoDAD(make_args(cfg, conditioning_strategy="cat"))
    assert m3.conditioning_strategy == "concat" and m3.input_n_frames == 6 and m3.condition_encoder is None


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mocodad_hip.h")).read()
    declared = set(re.findall(r"\b(mcd_[a-z_0-9]+)\s*\(", hdr))
    assert {"mcd_pack_weights", "mcd_score", "mcd_unet_forward", "mcd_cond_encode", "mcd_aggregate"} <= declared
    so = os.path.join(ROOT, "mocodad_amd", "libmocodad_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mocodad_hip.h but not exported"
    lib.mcd_abi_version.restype = ctypes.c_int32
    from mocodad_amd import _lib
    assert lib.mcd_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define MCD_ABI_VERSION (\d+)", hdr).group(1))
    assert set(_lib.EXPORTS) == declared


def test_pack_weights_reports_missing_tensor_without_a_gpu():
    from mocodad_amd import _lib
    L = _lib.lib()
    cfg = _lib.ModelCfg(num_coords=2, n_joints=17, t_unet=3, t_cond=3, emb_dim=16, strategy=0, cond_layers=4)
    for i, c in enumerate([32, 16, 32, 32]):
        cfg.cond_channels[i] = c
    arr = (_lib.Tensor * 1)()
    buf = torch.zeros(4)
    arr[0].name, arr[0].data, arr[0].numel = b"model.nothing", buf.data_ptr(), 4
    h = ctypes.c_void_p()
    rc = L.mcd_pack_weights(arr, 1, ctypes.byref(cfg), 0, ctypes.byref(h))
    assert rc == -2 and b"missing tensor" in L.mcd_last_error()
    cfg.t_unet = 33            # beyond MCD_MAX_FRAMES (any count up to 32 is served: specialised kernels or the runtime-shape one)
    assert L.mcd_pack_weights(arr, 1, ctypes.byref(cfg), 0, ctypes.byref(h)) == -4


def test_window_views_materialize_like_the_reference_dataset():
    """TrajectoryWindows / WindowBatch (views into trajectories + affine on load) reproduce what the reference's
    dataset materialises: the golden transformed windows, in transform-major order."""
    from mocodad_amd.data.windows import TrajectoryWindows
    g = load_golden("transforms.npz")
    base = g["base"]                                    # (N,2,T,V): treat each window as a 6-frame trajectory
    trajs = {(1, 1, i + 1): (1, np.ascontiguousarray(base[i].transpose(1, 0, 2))) for i in range(base.shape[0])}
    tw = TrajectoryWindows(trajs, seg_len=6, num_transform=5)
    assert len(tw) == 5 * base.shape[0] and tw.n_samples == base.shape[0]
    mat = tw.materialize().numpy()
    for tr in range(5):
        np.testing.assert_allclose(mat[tr * base.shape[0]:(tr + 1) * base.shape[0]], g[f"out_{tr}"], atol=1e-6, rtol=0)
    assert tw.trans.tolist() == sorted(tw.trans.tolist()) and tw.meta.shape == (len(tw), 4)
    # sliding windows over a longer trajectory: window s starts at frame first_frame + s
    long = {(1, 2, 1): (5, np.arange(10 * 2 * 17, dtype=np.float32).reshape(10, 2, 17))}
    tw2 = TrajectoryWindows(long, seg_len=6, num_transform=1)
    m2 = tw2.materialize()
    assert len(tw2) == 5 and tw2.frames[2].tolist() == [7, 8, 9, 10, 11, 12] and tw2.meta[2].tolist() == [1, 2, 1, 7]
    assert torch.equal(m2[2, 1, 0], torch.from_numpy(long[(1, 2, 1)][1][2, 1]))


def test_build_is_reproducible_across_output_paths(tmp_path):
    """The library must be a function of the sources and flags alone: tools/profile_set.sh stamps its sha256 into every profile
    and bench.py attaches PMC numbers only to the library they were taken on.  clang bakes a compilation-unit id derived from the
    command line -- output path included -- into each object; mocodad_amd/build.py pins it (-cuid).  One small unit of kernel
    instantiations compiled to two different paths (as two builds with private temporaries do) must give the same object."""
    import shutil
    import subprocess
    from mocodad_amd import build as B
    if shutil.which(B.HIPCC) is None:
        pytest.skip(f"no {B.HIPCC} on this host")
    probe = subprocess.run([B.HIPCC, "--offload-arch=gfx950", "--cuda-device-only", "-x", "hip", "-c", "/dev/null", "-o", os.devnull],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    if probe.returncode != 0:
        pytest.skip("this hipcc cannot target gfx950")
    flags = B.BASE_FLAGS + ["-DMCD_INST_UNIT=22"]
    src = os.path.join(B.CSRC, "mcd_inst.hip")
    outs = [str(tmp_path / "a" / "unit.o"), str(tmp_path / "b" / "unit.o.tmp4242")]
    for o in outs:
        os.makedirs(os.path.dirname(o))
        subprocess.run([B.HIPCC] + flags + ["-cuid=mcd_test_unit22", "-c", src, "-o", o], check=True, cwd=B.ROOT,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert open(outs[0], "rb").read() == open(outs[1], "rb").read()
    # ... and build_library passes such an id for every object it compiles
    import inspect
    assert "-cuid=" in inspect.getsource(B.build_library)
