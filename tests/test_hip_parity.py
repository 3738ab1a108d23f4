"""GPU: the HIP path (through the C ABI) vs. the golden vectors generated from the reference and vs. the
oracle on fresh seeded inputs.  Tolerance: 1e-4 absolute on scores (north_star), same bound on poses/eps."""
import json

import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
ATOL = 1e-4


def _scorer(variant):
    from mocodad_amd.engine import HipScorer
    from oracle import mocodad_oracle as O
    w = load_golden(f"weights_{variant}.npz")
    cfg = json.loads(bytes(w.pop("__cfg__")).decode())
    sd = {k: torch.from_numpy(v) for k, v in w.items()}
    strat = cfg["conditioning_strategy"]
    ci, xi = O.split_indices(cfg["seg_len"], cfg["conditioning_indices"], strat)
    sc = HipScorer(sd, strategy=strat, seg_len=cfg["seg_len"], cond_idx=ci, corrupt_idx=xi,
                   cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0")
    return sc, sd, cfg


@pytest.mark.parametrize("variant", ["inject", "concat", "T12", "injtail"])
def test_unet_pass_vs_golden(variant):
    sc, _, _ = _scorer(variant)
    g = load_golden(f"pass_{variant}.npz")
    x = torch.from_numpy(g["x"])
    cond = torch.from_numpy(g["cond"]) if "cond" in g else None
    for tv in (1, 9):
        eps = sc.unet_forward(x, tv, cond, noise_steps=10).cpu().numpy()
        np.testing.assert_allclose(eps, g[f"eps_t{tv}"], atol=ATOL, rtol=1e-5)


@pytest.mark.parametrize("variant", ["inject", "T12", "injtail"])
def test_cond_encoder_vs_golden(variant):
    sc, _, cfg = _scorer(variant)
    name = {"inject": "traj_inject_ns10_S5.npz", "T12": "traj_T12_ns10_S2.npz", "injtail": "traj_injtail_ns10_S2.npz"}[variant]
    g = load_golden(name)
    data = torch.from_numpy(g["data"])
    emb = sc.cond_encode(data[:, :, sc.cond_idx, :]).cpu().numpy()
    np.testing.assert_allclose(emb, g["cond_emb"], atol=2e-5, rtol=1e-5)
    if variant == "inject":
        gl = load_golden("layers_inject.npz")
        emb = sc.cond_encode(torch.from_numpy(gl["cond_in"])).cpu().numpy()
        np.testing.assert_allclose(emb, gl["cond_emb"], atol=2e-5, rtol=1e-5)


CASES = [("inject", 2, 1), ("inject", 10, 5), ("inject", 50, 8), ("concat", 10, 5), ("T12", 10, 2), ("injtail", 10, 2),
         ("T12", 50, 8), ("concat", 50, 2)]      # the last two: the long chains (gain 1014x) of the 12- and 6-frame U-Nets


@pytest.mark.parametrize("variant,ns,S", CASES)
def test_trajectory_vs_golden(variant, ns, S):
    sc, _, _ = _scorer(variant)
    g = load_golden(f"traj_{variant}_ns{ns}_S{S}.npz")
    data = torch.from_numpy(g["data"])
    noise = torch.from_numpy(g["noise"].astype(np.float32))
    loss, poses = sc.score(data, n_samples=S, noise_steps=ns, noise=noise, want_poses=True)
    np.testing.assert_allclose(poses.cpu().numpy(), g["poses_all"], atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss_all"], atol=ATOL, rtol=0)
    for aggr in ("best", "worst", "mean", "median", "mean_pose", "median_pose", "quantile:0.3"):
        key = aggr.replace(":", "_").replace(".", "p")
        sel, l = sc.aggregate(data, loss, poses, aggr, noise_steps=ns)
        np.testing.assert_allclose(l.cpu().numpy(), g[f"loss_{key}"], atol=ATOL, rtol=0, err_msg=aggr)
        if sel is not None:
            np.testing.assert_allclose(sel.cpu().numpy(), g[f"pose_{key}"], atol=ATOL, rtol=1e-5, err_msg=aggr)


HOSTILE = ["hostile_inject", "hostile_concat", "hostile_T12"]


@pytest.mark.parametrize("variant", HOSTILE)
def test_hostile_weights_vs_reference(variant):
    """Reference-generated vectors from weights with TRAINED-SCALE statistics (tests/golden/gen_golden.py --extra5): folded
    BatchNorm gains spread over 0.1x..10x per channel (some negative), PReLU slopes 1.5 / -0.2 / 0.01 / 0 on different layers
    (slope > 1 and slope < 0 take the other branch of the kernel's med3 form of PReLU), the last layer not scaled down,
    windows pushed against the +-5 clip.  Single passes, whole trajectories (B 4, ns 10, S 2) and every aggregation, on the
    specialised kernel (3, 6 and 12 frames), the one-launch fused form and the runtime-shape kernel.
    Bound: 1e-4 relative to the largest score / pose / eps value of the fixture."""
    sc, _, _ = _scorer(variant)
    gp = load_golden(f"pass_{variant}.npz")
    cond = torch.from_numpy(gp["cond"]) if "cond" in gp else None
    g = load_golden(f"traj_{variant}_ns10_S2.npz")
    data = torch.from_numpy(g["data"])
    noise = torch.from_numpy(g["noise"].astype(np.float32))
    tol = 1e-4 * float(np.abs(g["loss_all"]).max())
    ptol = 1e-4 * float(np.abs(g["poses_all"]).max())
    if "cond_emb" in g:
        emb = sc.cond_encode(data[:, :, sc.cond_idx, :]).cpu().numpy()
        np.testing.assert_allclose(emb, g["cond_emb"], atol=1e-4 * float(np.abs(g["cond_emb"]).max()), rtol=0)
    worst = 0.0
    for generic in (0, 1):
        sc.set_option("generic_unet", generic)
        for tv in (1, 9):
            eps = sc.unet_forward(torch.from_numpy(gp["x"]), tv, cond, noise_steps=10).cpu().numpy()
            np.testing.assert_allclose(eps, gp[f"eps_t{tv}"], atol=1e-4 * float(np.abs(gp[f"eps_t{tv}"]).max()), rtol=0)
        loss, poses = sc.score(data, n_samples=2, noise_steps=10, noise=noise, want_poses=True)
        worst = max(worst, float(np.abs(loss.cpu().numpy() - g["loss_all"]).max()))
        np.testing.assert_allclose(poses.cpu().numpy(), g["poses_all"], atol=ptol, rtol=0)
        np.testing.assert_allclose(loss.cpu().numpy(), g["loss_all"], atol=tol, rtol=0)
        for aggr in ("best", "worst", "mean", "median", "mean_pose", "median_pose", "quantile:0.3"):
            key = aggr.replace(":", "_").replace(".", "p")
            sel, l = sc.aggregate(data, loss, poses, aggr, noise_steps=10)
            np.testing.assert_allclose(l.cpu().numpy(), g[f"loss_{key}"], atol=tol, rtol=0, err_msg=aggr)
            if sel is not None:
                np.testing.assert_allclose(sel.cpu().numpy(), g[f"pose_{key}"], atol=ptol, rtol=0, err_msg=aggr)
    sc.set_option("generic_unet", 0)
    for split in (0, 1, 2):
        sc.set_option("split", split)
        agg, all_, _ = sc.score_fused(data, n_samples=2, noise_steps=10, aggregation="best", noise=noise, want_all=True)
        np.testing.assert_allclose(agg.cpu().numpy(), g["loss_best"], atol=tol, rtol=0)
        np.testing.assert_allclose(all_.cpu().numpy(), g["loss_all"], atol=tol, rtol=0)
    print(f"{variant}: max |score - reference| = {worst:.3e} on scores up to {float(np.abs(g['loss_all']).max()):.2f}")


def test_quantile_out_of_range_is_rejected():
    """torch.quantile raises for q outside [0, 1] (mocodad.py:513-516); so do the host wrapper and the C ABI."""
    sc, _, _ = _scorer("inject")
    data = torch.randn(4, 2, 6, 17)
    for bad in ("quantile:90", "quantile:-0.1", "quantile:nan"):
        with pytest.raises(ValueError):
            sc.score_fused(data, n_samples=3, noise_steps=4, aggregation=bad)
        with pytest.raises(ValueError):
            sc.aggregate(data, torch.zeros(4, 3, device="cuda:0"), None, bad, noise_steps=4)
    import ctypes as C
    from mocodad_amd import _lib
    cfg = sc._score_cfg(4, 3, 4, "smooth_l1")
    la = torch.zeros(4, 3, device="cuda:0")
    out = torch.zeros(4, device="cuda:0")
    rc = sc.L.mcd_aggregate(C.byref(cfg), 2, 17, _lib.AGGR["quantile"], C.c_float(1.5), C.c_void_p(la.data_ptr()), None, None,
                            C.c_void_p(out.data_ptr()), None, None)
    assert rc == -1 and b"quantile" in sc.L.mcd_last_error()


FUSED_AGGR = ("best", "worst", "mean", "median", "quantile:0.3")


@pytest.mark.parametrize("variant,ns,S", [("inject", 10, 5), ("inject", 50, 8), ("concat", 10, 5), ("T12", 10, 2), ("injtail", 10, 2)])
def test_fused_scoring_vs_golden(variant, ns, S):
    """mcd_score_fused (condition encoder + trajectories + aggregation over the samples in one call) against the reference's
    aggregated losses, for the three ways the call can be cut into workgroups: the library's choice, window-major (a
    workgroup runs all samples of its windows: ONE launch, encoder and aggregation inside) and chain-major (one trajectory
    per workgroup, aggregate_kernel afterwards).  Per-sample losses are bit-identical across the three."""
    sc, _, _ = _scorer(variant)
    g = load_golden(f"traj_{variant}_ns{ns}_S{S}.npz")
    data = torch.from_numpy(g["data"])
    noise = torch.from_numpy(g["noise"].astype(np.float32))
    ref_all = None
    for split in (0, 1, S):
        sc.set_option("split", split)
        for aggr in FUSED_AGGR:
            key = aggr.replace(":", "_").replace(".", "p")
            agg, all_, _ = sc.score_fused(data, n_samples=S, noise_steps=ns, aggregation=aggr, noise=noise, want_all=True)
            np.testing.assert_allclose(agg.cpu().numpy(), g[f"loss_{key}"], atol=ATOL, rtol=0, err_msg=f"{aggr} split={split}")
            np.testing.assert_allclose(all_.cpu().numpy(), g["loss_all"], atol=ATOL, rtol=0)
            ref_all = all_.clone() if ref_all is None else ref_all
            assert torch.equal(all_, ref_all), f"split={split}"
            only, none_, _ = sc.score_fused(data, n_samples=S, noise_steps=ns, aggregation=aggr, noise=noise)
            assert none_ is None and torch.equal(only, agg)
    # the stand-alone condition-encoder launch (runtime-channel-list kernel) feeds the same trajectories
    if variant != "concat":
        sc.set_option("split", 0)
        sc.set_option("cond_generic", 1)
        agg, _, _ = sc.score_fused(data, n_samples=S, noise_steps=ns, aggregation="best", noise=noise)
        np.testing.assert_allclose(agg.cpu().numpy(), g["loss_best"], atol=ATOL, rtol=0)


@pytest.mark.parametrize("B", [1, 2, 3, 5, 511, 513])
def test_fused_scoring_ragged_batches_perf_mode(B):
    """Batches that leave workgroup window slots empty / do not fill the device, perf mode: fused == unfused, any split."""
    sc, _, _ = _scorer("inject")
    gen = torch.Generator().manual_seed(B)
    data = torch.randn(B, 2, 6, 17, generator=gen)
    loss, _ = sc.score(data, n_samples=3, noise_steps=4, seed=5, first_window_id=17)
    for split in (0, 1, 3):
        sc.set_option("split", split)
        agg, all_, _ = sc.score_fused(data, n_samples=3, noise_steps=4, aggregation="best", seed=5, first_window_id=17, want_all=True)
        assert torch.equal(all_, loss) and torch.equal(agg, loss.min(1)[0])
        mean, _, _ = sc.score_fused(data, n_samples=3, noise_steps=4, aggregation="mean", seed=5, first_window_id=17)
        np.testing.assert_allclose(mean.cpu().numpy(), loss.mean(1).cpu().numpy(), rtol=1e-6)


@pytest.mark.parametrize("B", [1, 3, 4, 5, 37])
def test_ragged_batch_vs_oracle(B):
    """Batch sizes that do not fill the last workgroup's chain slots: parity vs. the oracle."""
    from oracle import mocodad_oracle as O
    sc, sd, cfg = _scorer("inject")
    gen = torch.Generator().manual_seed(100 + B)
    data = torch.randn(B, 2, 6, 17, generator=gen)
    S, ns = 3, 4
    noise = torch.randn(S, ns - 1, B, 2, 3, 17, generator=gen)
    loss, poses = sc.score(data, n_samples=S, noise_steps=ns, noise=noise, want_poses=True)
    with torch.no_grad():
        p_ref, corrupt = O.reverse_diffusion(sd, data, noise, noise_steps=ns, strategy="inject", conditioning_indices=[0, 1, 2])
        l_ref = O.window_losses(p_ref, corrupt)
    np.testing.assert_allclose(poses.cpu().numpy(), p_ref.transpose(0, 1).numpy(), atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy(), l_ref.t().numpy(), atol=ATOL, rtol=0)


def _random_model(strategy, seg_len, ci, arch="AE", seed=5):
    """A randomly initialised model with perturbed BatchNorm statistics (no reference-generated fixture covers these shapes)."""
    from helpers import golden_weights, make_args
    from mocodad_amd.models.mocodad import MoCoDAD
    _, cfg = golden_weights("inject")
    torch.manual_seed(seed)
    m = MoCoDAD(make_args(cfg, conditioning_strategy=strategy, seg_len=seg_len, conditioning_indices=ci, noise_steps=4,
                          n_generated_samples=2, conditioning_architecture=arch))
    gen = torch.Generator().manual_seed(17)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) + 0.5)
                mod.weight.copy_(torch.rand(mod.weight.shape, generator=gen) + 0.5)
                mod.bias.copy_(torch.randn(mod.bias.shape, generator=gen) * 0.1)
        last = m.model.st_gcnnsu3[-1]              # keep the random eps-prediction O(1) over the chain
        last.tcn[0].weight.mul_(0.25)
        last.residual[0].weight.mul_(0.25)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m, sd, gen


@pytest.mark.parametrize("strategy,seg_len,ci,arch", [
    ("inject", 8, 2, "AE"), ("concat", 8, [0, 1, 2, 3], "AE"), ("inject", 12, 3, "AE"),          # 4 and 8 U-Net frames: specialised kernels
    # 5 and 10 U-Net frames (seg_len 10 / 20 split in halves, seg_len 10 with every frame in the U-Net): specialised since round 3
    ("inject", 10, 2, "AE"), ("no_condition", 5, None, "AE"), ("inject", 10, 2, "E_unet"), ("inbetween_imp", 10, 2, "AE"),
    ("inject", 20, 2, "AE"), ("concat", 10, [0, 1, 2], "AE"),
    # 7, 9 and 11 U-Net frames: specialised too (one output frame per mix unit)
    ("concat", 7, [0, 1, 2], "AE"), ("inject", 18, 2, "AE"), ("inject", 22, 2, "E_unet"), ("inbetween_imp", 9, 3, "AE"),
    # 1 and 2 U-Net frames (seg_len 4 / 2 split in halves, a 1-frame condition): specialised as well
    ("inject", 4, 2, "AE"), ("inject", 2, 2, "E_unet"), ("concat", 2, [0], "AE"), ("no_condition", 1, None, "AE"),
    # 13 .. 32 U-Net frames: the slab-tiled MFMA kernel (frame count padded to 16 / 24 / 32), cross-checked with the plain-FMA kernel
    ("inject", 32, 2, "AE"), ("concat", 24, [0, 1, 2, 3], "AE"), ("concat", 13, [0, 1, 2], "AE"), ("inject", 26, 2, "E_unet"),
    ("concat", 20, [0, 1], "AE"), ("inbetween_imp", 30, 3, "AE"), ("no_condition", 17, None, "AE"), ("concat", 32, [28, 29, 30, 31], "AE"),
    # 25 .. 31 condition frames: the plain condition encoder with one of its three activation buffers in global scratch
    ("inject", 32, list(range(28)), "AE"), ("inject", 30, list(range(4, 30)), "AE"), ("inject", 32, list(range(31)), "E_unet"),
    ("inject", 32, list(range(25)), "AE"),
    # 13 .. 20 condition frames of the shipped encoder: cond_fast_kernel (MFMA, one window per workgroup); 'E_unet' at 13 .. 32
    # condition frames: the slab-tiled stages in their COND form (16: above; here the 24-frame padding)
    ("inject", 20, list(range(17)), "AE"), ("inject", 24, list(range(20)), "AE"), ("inject", 26, list(range(7, 26)), "AE"),
    ("inject", 28, list(range(20)), "E_unet")])
def test_other_frame_counts_vs_oracle(strategy, seg_len, ci, arch):
    """U-Net frame counts that no reference-generated fixture covers, HIP vs. oracle: 1, 2, 4, 5, 7 .. 11 frames on the
    specialised kernels (seg_len 10 split 5 + 5, seg_len 20 split 10 + 10, concat over 10 frames, a 1-frame window, ...); 13 .. 32
    frames (concat over 13 / 20 / 24 / 32 frames, 16 + 16, in-between imputation over 30, ...) on the slab-tiled MFMA kernel; both
    cross-checked against the plain-FMA runtime-shape kernel; 'E_unet' encoder included.
    (concat with a 4-frame condition at the END of a 32-frame window chains 28 predictions: bound relative to the poses there.)"""
    from oracle import mocodad_oracle as O
    m, sd, gen = _random_model(strategy, seg_len, ci, arch)
    m = m.to("cuda:0")
    B, S, ns = 5, 2, 4
    data = torch.randn(B, 2, seg_len, 17, generator=gen).clamp_(-3, 3)
    Tx = m.n_frames_corrupt
    noise = torch.randn(S, ns - 1, B, 2, Tx, 17, generator=gen)
    batch = [data, torch.zeros(B), torch.zeros(B, 4), torch.zeros(B, seg_len)]
    out = m.forward(batch, aggr_strategy="all", return_="all", noise=noise)
    with torch.no_grad():
        p_ref, corrupt = O.reverse_diffusion(sd, data, noise, noise_steps=ns, strategy=strategy, conditioning_indices=ci)
        l_ref = O.window_losses(p_ref, corrupt)
    # the bound is the absolute 1e-4 everywhere except on the long chains of predictions (concat / imputation over more than 12
    # U-Net frames, where a prediction's error feeds the next frame's input): relative to the largest pose there
    long_chain = strategy in ("concat", "inbetween_imp") and m.input_n_frames > 12
    scale = max(1.0, float(p_ref.abs().max())) if long_chain else 1.0
    np.testing.assert_allclose(out[1].cpu().numpy(), p_ref.transpose(0, 1).numpy(), atol=ATOL * scale, rtol=1e-5)
    np.testing.assert_allclose(out[0].cpu().numpy(), l_ref.t().numpy(), atol=ATOL * scale, rtol=0)
    # perf mode (in-kernel Philox) == the same kernel fed with the exported draws
    sc = m.scorer()
    a, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=5, first_window_id=3)
    z = sc.philox_noise(B, n_samples=S, noise_steps=ns, seed=5, first_window_id=3)
    b, _ = sc.score(data, n_samples=S, noise_steps=ns, noise=z)
    assert torch.equal(a, b)
    # frame counts with an MFMA kernel (4, 5, 7 .. 11 specialised; 13 .. 32 slab-tiled): the plain-FMA runtime-shape kernel
    # forced on the same call agrees
    if m.input_n_frames not in (3, 6, 12):
        sc.set_option("generic_unet", 1)
        c, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=5, first_window_id=3)
        sc.set_option("generic_unet", 0)
        np.testing.assert_allclose(c.cpu().numpy(), a.cpu().numpy(), atol=ATOL, rtol=0)


@pytest.mark.parametrize("strategy,seg_len,ci,arch,generic", [
    ("inject", 6, 3, "AE", 0), ("inject", 12, 3, "E_unet", 0), ("concat", 12, [0, 1, 2], "AE", 0), ("inject", 24, 2, "AE", 0),
    ("inject", 10, 2, "AE", 0), ("inject", 20, 2, "E_unet", 0), ("concat", 8, [0, 1, 2, 3], "AE", 0), ("inject", 8, 2, "AE", 0),
    ("inject", 32, 2, "E_unet", 0), ("concat", 24, [0, 1, 2, 3], "AE", 0), ("concat", 13, [0, 1, 2], "AE", 0),
    ("inbetween_imp", 30, 3, "AE", 0), ("no_condition", 17, None, "AE", 0), ("concat", 7, [0, 1, 2], "AE", 0),
    ("inject", 22, 2, "E_unet", 0), ("inject", 4, 2, "AE", 0),
    ("inject", 6, 3, "E_unet", 1), ("concat", 24, [0, 1, 2, 3], "AE", 1)])
def test_no_uninitialised_reads(strategy, seg_len, ci, arch, generic):
    """Every kernel family with the memory it does not own turned hostile: the LDS of every CU filled with NaN patterns
    (mcd_debug_poison_lds) and the allocator's free blocks -- which the scorer's workspace is carved from -- filled with NaNs,
    before the FIRST call of a fresh scorer.  A kernel that reads shared memory or scratch it never wrote (a pad row met by a
    zero coefficient is enough: 0 * NaN) then returns NaNs or differs from the clean second call; both are checked.
    (Found one in round 3: the slab-tiled kernel's last layer read 16-channel blocks past its 4-channel product buffer.)"""
    from mocodad_amd import _lib
    m, sd, gen = _random_model(strategy, seg_len, ci, arch)
    m = m.to("cuda:0")
    sc = m.scorer()
    if generic:
        sc.set_option("generic_unet", 1)
        sc.set_option("cond_generic", 1)
    B, S, ns = 7, 3, 4                  # (21 trajectories: an odd count for the kernels that take them in pairs)
    data = torch.randn(B, 2, seg_len, 17, generator=gen).clamp_(-3, 3)
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    junk = torch.full((96 << 20,), float("nan"), device="cuda:0")          # 384 MB of NaNs handed back to the caching allocator
    del junk
    L = _lib.lib()
    assert L.mcd_debug_poison_lds(None) == 0
    torch.cuda.synchronize()
    a, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=11, first_window_id=2)
    a = a.cpu().numpy()
    assert np.isfinite(a).all()
    b, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=11, first_window_id=2)
    assert np.array_equal(a, b.cpu().numpy())
    assert L.mcd_debug_poison_lds(None) == 0
    c, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=11, first_window_id=2)
    assert np.array_equal(a, c.cpu().numpy())


@pytest.mark.parametrize("variant,ns,S", [("inject", 10, 5), ("concat", 10, 5), ("T12", 10, 2), ("injtail", 10, 2)])
def test_runtime_shape_kernel_vs_golden_and_specialised(variant, ns, S):
    """The runtime-shape fallback forced (option 'generic_unet') on shapes the specialised kernels serve: it reproduces the
    reference-generated trajectories, and agrees with the specialised kernel in perf mode (same Philox keys) to rounding."""
    sc, _, _ = _scorer(variant)
    g = load_golden(f"traj_{variant}_ns{ns}_S{S}.npz")
    data = torch.from_numpy(g["data"])
    noise = torch.from_numpy(g["noise"].astype(np.float32))
    fast, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=9)
    sc.set_option("generic_unet", 1)
    loss, poses = sc.score(data, n_samples=S, noise_steps=ns, noise=noise, want_poses=True)
    np.testing.assert_allclose(poses.cpu().numpy(), g["poses_all"], atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss_all"], atol=ATOL, rtol=0)
    slow, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=9)
    np.testing.assert_allclose(slow.cpu().numpy(), fast.cpu().numpy(), atol=ATOL, rtol=0)
    gp = load_golden(f"pass_{variant}.npz")
    cond = torch.from_numpy(gp["cond"]) if "cond" in gp else None
    eps = sc.unet_forward(torch.from_numpy(gp["x"]), 9, cond, noise_steps=10).cpu().numpy()
    np.testing.assert_allclose(eps, gp["eps_t9"], atol=ATOL, rtol=1e-5)


@pytest.mark.parametrize("name", ["cattail", "cattail2", "imp2", "nocond", "encU"])
def test_runtime_shape_kernel_vs_extra_goldens(name):
    """The fallback forced on the reference-generated vectors of the frame layouts with index quirks: concat with the
    condition at the END (predictions read at the corrupt frames' original indices; with 2 + 4 frames a prediction drives a
    frame that is another prediction's input), in-between imputation, no condition, and the 'E_unet' encoder through
    cond_unet_generic_kernel (option 'cond_generic')."""
    from oracle import mocodad_oracle as O
    sc, _, cfg = _scorer(name) if name != "encU" else (None, None, None)
    if name == "encU":
        from mocodad_amd.engine import HipScorer
        w = load_golden("weights_encU.npz")
        cfg = json.loads(bytes(w.pop("__cfg__")).decode())
        sd = {k: torch.from_numpy(v) for k, v in w.items()}
        sc = HipScorer(sd, strategy="inject", seg_len=6, cond_idx=[0, 1, 2], corrupt_idx=[3, 4, 5], cond_unet=True, device="cuda:0")
        sc.set_option("cond_generic", 1)
    sc.set_option("generic_unet", 1)
    g = load_golden(f"traj_{name}_ns4_S2.npz")
    loss, poses = sc.score(torch.from_numpy(g["data"]), n_samples=2, noise_steps=4,
                           noise=torch.from_numpy(g["noise"].astype(np.float32)), want_poses=True)
    np.testing.assert_allclose(poses.cpu().numpy(), g["pose_all"], atol=ATOL, rtol=1e-5)
    np.testing.assert_allclose(loss.cpu().numpy(), g["loss_all"], atol=ATOL, rtol=0)


def test_empty_batch():
    sc, _, _ = _scorer("inject")
    loss, poses = sc.score(torch.zeros(0, 2, 6, 17), n_samples=2, noise_steps=4, want_poses=True)
    assert loss.shape == (0, 2) and poses.shape == (0, 2, 2, 3, 17)


def test_philox_noise_statistics_and_determinism():
    """Perf mode (in-kernel Philox): deterministic in (seed, window id), independent of batch split,
    and the generated x_T ~ N(0,1) (checked through a noise_steps=2 ... no: through the sample mean/var
    of many chains' final poses being finite and seed-dependent)."""
    sc, _, _ = _scorer("inject")
    gen = torch.Generator().manual_seed(5)
    data = torch.randn(64, 2, 6, 17, generator=gen)
    a, _ = sc.score(data, n_samples=4, noise_steps=5, seed=11)
    b, _ = sc.score(data, n_samples=4, noise_steps=5, seed=11)
    c, _ = sc.score(data, n_samples=4, noise_steps=5, seed=12)
    assert torch.equal(a, b)
    assert not torch.equal(a, c)
    assert torch.isfinite(a).all()
    # window-id keyed: scoring the second half alone with first_window_id=32 reproduces the same scores
    h, _ = sc.score(data[32:], n_samples=4, noise_steps=5, seed=11, first_window_id=32)
    assert torch.equal(a[32:], h)


@pytest.mark.parametrize("variant,B,S,ns", [("inject", 1024, 5, 10), ("inject", 700, 3, 6), ("concat", 512, 2, 6), ("T12", 64, 2, 4)])
def test_repeated_calls_are_bit_identical(variant, B, S, ns):
    """Race detector.  The trajectory kernels drop workgroup barriers wherever a stage reads only what the same wave wrote
    (wave-aligned units); a barrier removed where SLOWER waves still read what a faster one overwrites is a race that the golden
    tests pass (it needs the right timing) and that shows up as run-to-run differences on a busy GPU -- round 6 removed one
    barrier too many and exactly this kind of test caught it.  Same call 12 times, full grid, both schedules: all bit-identical."""
    sc, _, _ = _scorer(variant)
    gen = torch.Generator().manual_seed(123)
    data = torch.randn(B, 2, sc.seg_len, 17, generator=gen).clamp_(-3, 3).cuda()
    for split in (0, 1):
        sc.set_option("split", split)
        ref = sc.score(data, n_samples=S, noise_steps=ns, seed=11)[0].clone()
        assert torch.isfinite(ref).all()
        for _ in range(11):
            assert torch.equal(sc.score(data, n_samples=S, noise_steps=ns, seed=11)[0], ref), (variant, split)


def test_overlapping_launches_on_two_streams():
    """Batches scored on alternating HIP streams (bench.py --streams 2) overlap on the GPU; each stream has its own
    condition-embedding workspace, so the results equal the one-stream ones bit for bit."""
    sc, _, _ = _scorer("inject")
    gen = torch.Generator().manual_seed(9)
    batches = [torch.randn(700, 2, 6, 17, generator=gen).cuda() for _ in range(6)]
    ref = [sc.score(b, n_samples=3, noise_steps=6, seed=40 + i)[0].clone() for i, b in enumerate(batches)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    out = []
    for i, b in enumerate(batches):
        with torch.cuda.stream(streams[i % 2]):
            out.append(sc.score(b, n_samples=3, noise_steps=6, seed=40 + i)[0])
    torch.cuda.synchronize()
    for a, b in zip(ref, out):
        assert torch.equal(a, b)
