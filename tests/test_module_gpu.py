"""GPU: the drop-in module surface (MoCoDAD.forward / test_step / on_test_epoch_end) against golden vectors and the
oracle, the device scatter-max, and size-independent properties at the BASELINE batch size."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import golden_weights, make_args

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def _model(variant, **over):
    from mocodad_amd.models.mocodad import MoCoDAD
    sd, cfg = golden_weights(variant)
    m = MoCoDAD(make_args(cfg, **over)).to("cuda:0")
    m.load_state_dict(sd)
    return m, sd, cfg


@pytest.mark.parametrize("variant,ns,S", [("inject", 10, 5), ("concat", 10, 5), ("injtail", 10, 2), ("T12", 10, 2)])
def test_forward_matches_reference_outputs(variant, ns, S):
    g = load_golden(f"traj_{variant}_ns{ns}_S{S}.npz")
    m, _, _ = _model(variant, noise_steps=ns, n_generated_samples=S)
    batch = [torch.from_numpy(g[k]) for k in ("data", "trans", "meta", "frames")]
    noise = torch.from_numpy(g["noise"].astype(np.float32))
    for aggr in ("best", "worst", "mean", "median", "mean_pose", "median_pose", "quantile:0.3", "all"):
        key = aggr.replace(":", "_").replace(".", "p")
        out = m.forward(batch, aggr_strategy=aggr, return_="all", noise=noise)
        assert len(out) == 6 and out[2].shape == batch[0].shape and out[3] is batch[1]
        np.testing.assert_allclose(out[0].cpu().numpy(), g[f"loss_{key}"], atol=ATOL, rtol=0, err_msg=aggr)
        if f"pose_{key}" in g:
            np.testing.assert_allclose(out[1].cpu().numpy(), g[f"pose_{key}"], atol=ATOL, rtol=1e-5, err_msg=aggr)
        elif aggr == "all":
            np.testing.assert_allclose(out[1].cpu().numpy(), g["poses_all"], atol=ATOL, rtol=1e-5)
        else:
            assert out[1] is None
    only_loss = m.forward(batch, aggr_strategy="best", return_="loss", noise=noise)
    assert len(only_loss) == 5
    with pytest.raises(ValueError):
        m.forward(batch, aggr_strategy="nope", noise=noise)
    m.model_return_value = None
    with pytest.raises(ValueError):
        m.forward(batch, noise=noise)


def test_reload_state_dict_repacks():
    m, sd, _ = _model("inject", noise_steps=4, n_generated_samples=2)
    gen = torch.Generator().manual_seed(3)
    batch = [torch.randn(8, 2, 6, 17, generator=gen), torch.zeros(8), torch.zeros(8, 4), torch.zeros(8, 6)]
    noise = torch.randn(2, 3, 8, 2, 3, 17, generator=gen)
    a = m.forward(batch, noise=noise)[0].clone()
    sd2 = {k: (v * 1.01 if v.dtype.is_floating_point and "running_var" not in k else v) for k, v in sd.items()}
    m.load_state_dict(sd2)
    b = m.forward(batch, noise=noise)[0]
    assert not torch.allclose(a, b)
    m.load_state_dict(sd)
    assert torch.equal(a, m.forward(batch, noise=noise)[0])


def test_test_loop_end_to_end_auc_vs_oracle(tmp_path):
    """test_step -> on_test_epoch_end -> AUC, against the oracle's scores pushed through the oracle's
    post-processing (the reference's own loop: mocodad.py:230-274)."""
    from mocodad_amd.data import synthetic
    from oracle import mocodad_oracle as O
    data, trans, meta, frames, gts = synthetic.make_dataset(n_clips=2, frames_per_clip=40, persons_per_clip=2, num_transform=2)
    synthetic.write_gt(str(tmp_path), gts)
    ns, S = 4, 2
    m, sd, _ = _model("inject", noise_steps=ns, n_generated_samples=S, gt_path=str(tmp_path), num_transform=2,
                      dataset_choice="HR-STC", pad_size=-1, filter_kernel_size=3, frames_shift=2, save_tensors=False)
    gen = torch.Generator().manual_seed(11)
    m.on_test_epoch_start()
    ref_scores = []
    for batch in synthetic.batches((data, trans, meta, frames), 100):
        B = batch[0].shape[0]
        noise = torch.randn(S, ns - 1, B, 2, 3, 17, generator=gen)
        m._test_output_list.append(m.forward(batch, noise=noise))
        with torch.no_grad():
            ref_scores.append(O.score(sd, batch[0], noise, noise_steps=ns, aggregation="best")[1].numpy())
    auc = m.on_test_epoch_end()
    ref = np.concatenate(ref_scores)
    auc_ref, _, _ = O.post_processing(ref, trans.numpy(), meta.numpy(), frames.numpy(), gts, num_transform=2, pad_size=-1,
                                      filter_kernel_size=3, frames_shift=2)
    assert abs(auc - auc_ref) < 1e-3, (auc, auc_ref)     # north_star: AUC within +-0.1 points (0.001 absolute)
    assert m.logged["AUC"] == auc


def test_device_scatter_max_vs_the_reference_loop():
    """mcd_scatter_max against compute_var_matrix + np.nanmax restated here as the reference writes it (one zero matrix per
    (transform, clip, person), rows = windows, max over rows: eval_utils.py:27-34, mocodad.py:392-393)."""
    from mocodad_amd.utils import eval_utils as EU
    m, _, _ = _model("inject")
    sc = m.scorer()
    g = load_golden("postproc.npz")
    gts = {}
    for k in g:
        if k.startswith("gt_"):
            s_, c_ = k[3:].split("_")
            gts[(int(s_), int(c_))] = g[k]
    mat, keys = EU.frame_score_rows(g["out"], g["trans"], g["meta"], g["frames"], gts, 5, scatter_max=sc.scatter_max)
    assert len(keys) == 5 * 2 * 3
    for r, (tr, scn, cl, person) in enumerate(keys):
        sel = (g["trans"] == tr) & (g["meta"][:, 0] == scn) & (g["meta"][:, 1] == cl) & (g["meta"][:, 2] == person)
        n = len(gts[(scn, cl)])
        var = np.zeros((int(sel.sum()), n))
        for w, (val, fr) in enumerate(zip(g["out"][sel], g["frames"][sel])):
            var[w, fr - 1] = val
        np.testing.assert_array_equal(mat[r, :n], np.nanmax(var, axis=0).astype(np.float32).astype(np.float64))
        assert (mat[r, n:] == 0).all()


def test_full_size_properties():
    """BASELINE configs[1] batch (1024 windows, ns=10, S=5): deterministic, invariant to how the batch is split
    (noise keyed by global window id), finite, and best <= mean <= worst over the S samples."""
    m, _, _ = _model("inject", noise_steps=10, n_generated_samples=5)
    sc = m.scorer()
    gen = torch.Generator().manual_seed(0)
    data = torch.randn(1024, 2, 6, 17, generator=gen).clamp_(-5, 5)
    loss, _ = sc.score(data, n_samples=5, noise_steps=10, seed=7)
    loss2, _ = sc.score(data, n_samples=5, noise_steps=10, seed=7)
    assert torch.equal(loss, loss2) and torch.isfinite(loss).all()
    lo, _ = sc.score(data[:300], n_samples=5, noise_steps=10, seed=7, first_window_id=0)
    hi, _ = sc.score(data[300:], n_samples=5, noise_steps=10, seed=7, first_window_id=300)
    assert torch.equal(torch.cat([lo, hi]), loss)
    best = sc.aggregate(data, loss, None, "best", noise_steps=10)[1]
    mean = sc.aggregate(data, loss, None, "mean", noise_steps=10)[1]
    worst = sc.aggregate(data, loss, None, "worst", noise_steps=10)[1]
    assert torch.equal(best, loss.min(1)[0]) and torch.equal(worst, loss.max(1)[0])
    assert (best <= mean + 1e-6).all() and (mean <= worst + 1e-6).all()


@pytest.mark.parametrize("variant,B,ns,S,seg_len", [("concat", 1024, 10, 5, 6), ("T12", 4096, 50, 8, 24)])
def test_full_size_properties_other_configs(variant, B, ns, S, seg_len):
    """BASELINE configs[3] (concat, 1024 windows) and configs[4] shape (seg_len 24, 12 condition + 12 denoised frames,
    ns=50, S=8, 4096 windows): finite, deterministic, independent of the batch split, 1-sample-prefix consistent
    (the chains of sample s do not depend on how many samples are drawn), best <= mean <= worst."""
    m, _, _ = _model(variant, noise_steps=ns, n_generated_samples=S)
    sc = m.scorer()
    gen = torch.Generator().manual_seed(1)
    data = torch.randn(B, 2, seg_len, 17, generator=gen).clamp_(-5, 5)
    loss, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=11)
    assert torch.isfinite(loss).all()
    cut = B // 3
    lo, _ = sc.score(data[:cut], n_samples=S, noise_steps=ns, seed=11, first_window_id=0)
    hi, _ = sc.score(data[cut:], n_samples=S, noise_steps=ns, seed=11, first_window_id=cut)
    assert torch.equal(torch.cat([lo, hi]), loss)
    one, _ = sc.score(data[:64], n_samples=1, noise_steps=ns, seed=11)
    assert torch.equal(one[:, 0], loss[:64, 0])
    best = sc.aggregate(data, loss, None, "best", noise_steps=ns)[1]
    mean = sc.aggregate(data, loss, None, "mean", noise_steps=ns)[1]
    worst = sc.aggregate(data, loss, None, "worst", noise_steps=ns)[1]
    assert (best <= mean + 1e-6).all() and (mean <= worst + 1e-6).all()


@pytest.mark.parametrize("name", ["nocond", "encE", "l1", "mse", "imp2", "implist", "cattail", "encU", "rndimp", "cattail2"])
def test_extra_variants_vs_reference(name):
    """no_condition strategy (U-Net on all 6 frames), 'E' encoder with channels [24,40]+8 (generic condition-encoder
    kernel), l1 / mse losses, in-between imputation (every 2nd frame / an explicit list conditions), concat with the
    condition at the end of the window, 'E_unet' condition encoder, random imputation (per-window frame sets drawn
    from torch's RNG) — against vectors generated by the reference."""
    g = load_golden(f"traj_{name}_ns4_S2.npz")
    m, _, cfg = _model(name)
    B = g["data"].shape[0]
    batch = [torch.from_numpy(g["data"]), torch.zeros(B), torch.zeros(B, 4), torch.zeros(B, 6)]
    noise = torch.from_numpy(g["noise"].astype(np.float32))
    for aggr in ("all", "best", "mean"):
        if "rng_seed" in g:       # random_imp: the module draws the per-window frame sets like the reference does
            torch.manual_seed(int(g["rng_seed"][0]))
            assert torch.equal(m.draw_random_imp_mask(B), torch.from_numpy(g["cond_mask"]))
            torch.manual_seed(int(g["rng_seed"][0]))
        out = m.forward(batch, aggr_strategy=aggr, return_="all", noise=noise)
        np.testing.assert_allclose(out[0].cpu().numpy(), g[f"loss_{aggr}"], atol=ATOL, rtol=0, err_msg=aggr)
        if f"pose_{aggr}" in g:
            np.testing.assert_allclose(out[1].cpu().numpy(), g[f"pose_{aggr}"], atol=ATOL, rtol=1e-5, err_msg=aggr)
    if "cond_emb" in g:
        emb = m.scorer().cond_encode(batch[0][:, :, m._frame_split()[0], :]).cpu().numpy()
        np.testing.assert_allclose(emb, g["cond_emb"], atol=2e-5, rtol=1e-5)


def test_window_views_score_like_materialised_windows(tmp_path):
    """Windows read in place from trajectory buffers with the test-time transform applied on load (mcd_score_view)
    give the scores of the host-materialised (B,C,T,V) windows, and the test loop gives the same AUC."""
    from mocodad_amd.data import synthetic
    from mocodad_amd.data.windows import TrajectoryWindows
    trajs, gts = synthetic.make_trajectories(n_clips=2, frames_per_clip=40, persons_per_clip=2)
    synthetic.write_gt(str(tmp_path), gts)
    tw = TrajectoryWindows(trajs, seg_len=6, num_transform=5)
    dense = tw.materialize()
    m, _, _ = _model("inject", noise_steps=4, n_generated_samples=2, gt_path=str(tmp_path), num_transform=5,
                     dataset_choice="HR-STC", pad_size=-1, filter_kernel_size=3, frames_shift=2, save_tensors=False)
    sc = m.scorer()
    tw.to("cuda:0")
    a, pa = sc.score(tw.batch(0, len(tw))[0], n_samples=2, noise_steps=4, seed=3, want_poses=True)
    b, pb = sc.score(dense, n_samples=2, noise_steps=4, seed=3, want_poses=True)
    np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), atol=1e-5, rtol=0)
    np.testing.assert_allclose(pa.cpu().numpy(), pb.cpu().numpy(), atol=1e-5, rtol=1e-5)
    aucs = []
    for batches in (tw.batches(128), synthetic.batches((dense, tw.trans.long(), tw.meta, tw.frames), 128)):
        m.on_test_epoch_start()
        for i, batch in enumerate(batches):
            m._calls = i * 128
            m.test_step(batch, i)
        aucs.append(m.on_test_epoch_end())
    assert abs(aucs[0] - aucs[1]) < 1e-6, aucs


def test_c_abi_error_paths():
    """Argument / capability errors come back as negative codes + a message (raised as RuntimeError by the ctypes layer),
    never as a crash: too many frames, random_imp without its per-window masks, a wrong frame split, S > 64."""
    from mocodad_amd.engine import HipScorer
    from mocodad_amd.models.mocodad import MoCoDAD
    sd, cfg = golden_weights("inject")
    # more frames than MCD_MAX_FRAMES -> MCD_EUNSUPPORTED at pack time (every count up to 32 is served)
    m = MoCoDAD(make_args(cfg, seg_len=40, conditioning_strategy="concat", conditioning_indices=[0, 1, 2])).to("cuda:0")
    with pytest.raises(RuntimeError, match="must be in 1..32"):
        m.scorer()
    # random_imp needs the per-window condition-frame masks
    sdr, cfgr = golden_weights("rndimp")
    sc = HipScorer(sdr, strategy="random_imp", seg_len=6, cond_idx=[0, 1], corrupt_idx=[2, 3, 4, 5], device="cuda:0")
    with pytest.raises(ValueError, match="cond_mask"):
        sc.score(torch.zeros(2, 2, 6, 17), n_samples=1, noise_steps=3)
    # a frame split the checkpoint was not trained for: the gcn.T / gcn.A tensors have the wrong size -> MCD_EMISSING
    with pytest.raises(RuntimeError, match="elements, expected"):
        HipScorer(sd, strategy="inject", seg_len=6, cond_idx=[0, 1], corrupt_idx=[2, 3, 4, 5],
                  cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0")
    # aggregation over more than 64 samples is NOT an error (round 4 refused it; the reference has no cap, mocodad.py:454-520)
    sc3, = (HipScorer(sd, strategy="inject", seg_len=6, cond_idx=[0, 1, 2], corrupt_idx=[3, 4, 5],
                      cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0"),)
    la = torch.rand(2, 65, device="cuda:0")
    assert torch.equal(sc3.aggregate(torch.zeros(2, 2, 6, 17), la, None, "best", noise_steps=3)[1], la.min(1)[0])
    # ... and a (B,S) view with other strides is made contiguous by the wrapper (the C ABI takes dense tensors)
    assert torch.equal(sc3.aggregate(torch.zeros(2, 2, 6, 17), la.t().contiguous().t(), None, "worst", noise_steps=3)[1], la.max(1)[0])


def test_integration_md_ctypes_stub_runs_as_written(monkeypatch):
    """INTEGRATION.md section 2 -- the ctypes binding a maintainer of the reference would paste into models/mocodad.py -- is
    extracted from the document and executed VERBATIM: its struct layouts, call signatures and argument order against the
    built library, its result against a reference-generated trajectory (the reference's own attribute names on a stand-in
    object: mocodad.py:46-81)."""
    import os
    import types
    from mocodad_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = md.split("\n## 2.")[1].split("\n## 3.")[0]
    code = sec.split("```python\n")[1].split("\n```")[0]
    assert "mcd_pack_weights" in code and "mcd_score" in code and "mcd_aggregate" in code
    monkeypatch.setenv("MOCODAD_HIP_LIB", _lib.LIB_PATH)
    ns_ = {}
    exec(compile(code, "INTEGRATION.md#2", "exec"), ns_)
    sd, cfg = golden_weights("inject")
    g = load_golden("traj_inject_ns10_S5.npz")
    ref = types.SimpleNamespace(                       # what the reference's MoCoDAD.__init__ sets (mocodad.py:46-81)
        num_coords=2, n_joints=17, n_frames=cfg["seg_len"], input_n_frames=3, n_frames_condition=3, n_frames_corrupt=3,
        embedding_dim=cfg["embedding_dim"], conditioning_strategy="inject", conditioning_indices=list(cfg["conditioning_indices"]),
        cond_channels=list(cfg["channels"]), cond_h_dim=cfg["h_dim"], n_generated_samples=5, noise_steps=10,
        device=torch.device("cuda:0"), state_dict=lambda: sd)
    ns_["pack"](ref)
    data = torch.from_numpy(g["data"]).cuda().contiguous()
    noise = torch.from_numpy(g["noise"].astype(np.float32)).cuda().contiguous()
    best = ns_["hot_loop"](ref, data, noise=noise)
    torch.cuda.synchronize()
    np.testing.assert_allclose(best.cpu().numpy(), g["loss_best"], atol=ATOL, rtol=0)
    # perf mode through the stub == the shipped module's scorer with the same keys
    a = ns_["hot_loop"](ref, data, seed=7, first_window_id=100)
    m, _, _ = _model("inject", noise_steps=10, n_generated_samples=5)
    b, _, _ = m.scorer().score_fused(data, n_samples=5, noise_steps=10, aggregation="best", seed=7, first_window_id=100)
    assert torch.equal(a, b)
    ns_["lib"].mcd_free_weights(ref._h)
