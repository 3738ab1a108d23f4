"""GPU: the PERF mode of the path (noise == NULL: in-kernel Philox4x32-10 + Box-Muller), i.e. the mode bench.py and
eval_MoCoDAD.py run.  The reference draws torch.randn_like (mocodad.py:162,176) from an unseeded global generator
(eval_MoCoDAD.py sets no seed), so parity here is (1) the generator's draws are N(0,1), independent across every key
axis; (2) the perf mode IS the parity mode fed with those draws, bit for bit, and the oracle fed with the same draws
reproduces its scores within 1e-4; (3) the score distribution / AUC over seeds equals that of the oracle drawing
torch.randn (AUC within +-0.1 points = 0.001 absolute)."""
import numpy as np
import pytest
import torch

from helpers import golden_weights

pytestmark = pytest.mark.gpu


def _scorer():
    from mocodad_amd.engine import HipScorer
    sd, cfg = golden_weights("inject")
    sc = HipScorer(sd, strategy="inject", seg_len=6, cond_idx=[0, 1, 2], corrupt_idx=[3, 4, 5],
                   cond_channels=list(cfg["channels"]) + [cfg["h_dim"]], device="cuda:0")
    return sc, sd


def test_philox_normals_are_standard_normal_and_uncorrelated():
    from scipy import stats
    sc, _ = _scorer()
    S, ns, B = 5, 10, 1024
    z = sc.philox_noise(B, n_samples=S, noise_steps=ns, seed=20260928, first_window_id=12345).cpu().double()   # (S,K,B,2,3,17)
    n = z.numel()
    assert n >= 4_000_000
    flat = z.reshape(-1).numpy()
    m, v = flat.mean(), flat.var()
    sk, ku = stats.skew(flat), stats.kurtosis(flat)
    print(f"n={n} mean={m:.3e} var={v:.6f} skew={sk:.3e} excess kurtosis={ku:.3e} max|z|={np.abs(flat).max():.3f}")
    assert abs(m) < 4 / np.sqrt(n) and abs(v - 1) < 4 * np.sqrt(2 / n)
    assert abs(sk) < 4 * np.sqrt(6 / n) and abs(ku) < 4 * np.sqrt(24 / n)
    # Kolmogorov-Smirnov against N(0,1) on 1e6 draws of each kind (x_T: one Philox call per element; steps: four normals per call)
    for name, part in (("x_T", z[:, 0]), ("step noise", z[:, 1:])):
        sub = part.reshape(-1).numpy()[:1_000_000]
        d, p = stats.kstest(sub, "norm")
        print(f"KS {name}: D={d:.3e} p={p:.3f}")
        assert p > 1e-3, (name, d, p)
    # tails: P(|z| > 3) = 2.6998e-3
    frac3 = (np.abs(flat) > 3).mean()
    assert abs(frac3 - 2.6998e-3) < 4 * np.sqrt(2.6998e-3 / n), frac3

    def corr(a, b):
        a, b = a.reshape(-1), b.reshape(-1)
        return float(((a - a.mean()) * (b - b.mean())).mean() / (a.std() * b.std()))
    lim = lambda k: 4.5 / np.sqrt(k)
    pairs = {
        "lag-1 over joints": (z[..., :-1], z[..., 1:]),
        "x vs y coordinate": (z[:, :, :, 0], z[:, :, :, 1]),
        "consecutive frames": (z[:, :, :, :, :-1], z[:, :, :, :, 1:]),
        "consecutive steps": (z[:, :-1], z[:, 1:]),
        "consecutive samples": (z[:-1], z[1:]),
        "consecutive windows": (z[:, :, :-1], z[:, :, 1:]),
        "squares, x vs y (Box-Muller pair)": (z[:, :, :, 0] ** 2, z[:, :, :, 1] ** 2),
        "squares, joint pairs sharing a Philox call": (z[:, 1:, ..., 0:16:2] ** 2, z[:, 1:, ..., 1:17:2] ** 2),
    }
    for name, (a, b) in pairs.items():
        c = corr(a, b)
        print(f"corr {name}: {c:.3e} (limit {lim(a.numel()):.1e})")
        assert abs(c) < lim(a.numel()), name
    # a different seed / window offset gives different, equally distributed draws; the same keys reproduce
    z2 = sc.philox_noise(B, n_samples=S, noise_steps=ns, seed=20260929, first_window_id=12345).cpu().double()
    assert abs(corr(z, z2)) < lim(n)
    z3 = sc.philox_noise(B // 2, n_samples=S, noise_steps=ns, seed=20260928, first_window_id=12345 + B // 2).cpu().double()
    assert torch.equal(z3, z[:, :, B // 2:])


@pytest.mark.parametrize("ns,S", [(10, 5), (2, 1), (3, 2)])
def test_perf_mode_is_parity_mode_with_the_exported_noise(ns, S):
    """scores(noise=NULL, seed) == scores(noise = mcd_philox_noise(seed)) bit for bit, and the CPU oracle fed with that
    tensor reproduces them within the north_star tolerance: the benchmarked mode computes the reference's algorithm."""
    from oracle import mocodad_oracle as O
    sc, sd = _scorer()
    gen = torch.Generator().manual_seed(3)
    B = 96
    data = (torch.randn(B, 2, 1, 17, generator=gen) + torch.cumsum(torch.randn(B, 2, 6, 17, generator=gen) * 0.15, 2)).clamp_(-5, 5)
    perf, pperf = sc.score(data, n_samples=S, noise_steps=ns, seed=77, first_window_id=1000, want_poses=True)
    z = sc.philox_noise(B, n_samples=S, noise_steps=ns, seed=77, first_window_id=1000)
    par, ppar = sc.score(data, n_samples=S, noise_steps=ns, noise=z, want_poses=True)
    assert torch.equal(perf, par) and torch.equal(pperf, ppar)
    with torch.no_grad():
        poses, corrupt = O.reverse_diffusion(sd, data, z.cpu(), noise_steps=ns, strategy="inject", conditioning_indices=[0, 1, 2])
        ref = O.window_losses(poses, corrupt).t()
    err = (perf.cpu() - ref).abs().max().item()
    print(f"ns={ns} S={S}: max |perf-mode score - oracle(same draws)| = {err:.3e}")
    assert err < 1e-4


def test_score_distribution_and_auc_over_seeds_match_the_oracle(tmp_path):
    """BASELINE configs[1] shape (ns=10, S=5, 'best'), 5 seeds each: HIP perf mode (Philox) vs the oracle drawing torch.randn.
    Per-window mean score within 3 sigma of the seed-to-seed spread; AUC on the synthetic clips within +-0.001 (or 3 standard
    errors of the seed spread if that is larger)."""
    from mocodad_amd.data import synthetic
    from oracle import mocodad_oracle as O
    sc, sd = _scorer()
    data, trans, meta, frames, gts = synthetic.make_dataset(n_clips=3, frames_per_clip=80, persons_per_clip=2, num_transform=2)
    N, ns, S, seeds = data.shape[0], 10, 5, 5
    hip, ref = [], []
    for seed in range(seeds):
        loss, _ = sc.score(data, n_samples=S, noise_steps=ns, seed=1000 + seed)
        hip.append(loss.min(1)[0].cpu().numpy())
        g = torch.Generator().manual_seed(seed)
        noise = torch.randn(S, ns - 1, N, 2, 3, 17, generator=g)
        with torch.no_grad():
            ref.append(O.score(sd, data, noise, noise_steps=ns, aggregation="best")[1].numpy())
    hip, ref = np.stack(hip), np.stack(ref)            # (seeds, N)
    sd_seed = np.sqrt(0.5 * (hip.var(0, ddof=1) + ref.var(0, ddof=1)))       # per-window seed-to-seed spread
    zscore = (hip.mean(0) - ref.mean(0)) / (sd_seed * np.sqrt(2.0 / seeds) + 1e-12)
    print(f"N={N}: mean score hip {hip.mean():.6f} ref {ref.mean():.6f}; per-window z: mean {zscore.mean():.3f} rms {np.sqrt((zscore**2).mean()):.3f} max {np.abs(zscore).max():.2f}")
    # unbiased: the per-window z-scores average to 0 with the spread of a t statistic
    assert abs(zscore.mean()) < 6 / np.sqrt(N) and np.sqrt((zscore ** 2).mean()) < 1.6      # (t with 8 dof: rms 1.15)
    assert abs(hip.mean() - ref.mean()) < 3 * sd_seed.mean() / np.sqrt(seeds * N) * 3 + 1e-4
    kw = dict(num_transform=2, pad_size=-1, filter_kernel_size=3, frames_shift=2)
    auc_h = np.array([O.post_processing(h, trans.numpy(), meta.numpy(), frames.numpy(), gts, **kw)[0] for h in hip])
    auc_r = np.array([O.post_processing(r, trans.numpy(), meta.numpy(), frames.numpy(), gts, **kw)[0] for r in ref])
    se = np.sqrt((auc_h.var(ddof=1) + auc_r.var(ddof=1)) / seeds)
    print(f"AUC hip {auc_h.round(5)} median {np.median(auc_h):.5f}; oracle {auc_r.round(5)} median {np.median(auc_r):.5f}; |diff of means| {abs(auc_h.mean()-auc_r.mean()):.5f} (se {se:.5f})")
    assert abs(auc_h.mean() - auc_r.mean()) < max(1e-3, 3 * se)


def test_auc_over_thousands_of_seeds_philox_vs_torch_randn(tmp_path):
    """The acceptance reading of SURVEY.md 8d (AUC within 0.001 absolute) needs more seeds than a CPU oracle can run: the
    seed-to-seed AUC spread on a small clip set is ~0.03.  The same HIP kernel is therefore run in parity mode with
    torch.randn draws (the mode pinned to the reference within 1e-4 per score by the golden trajectories) and in perf mode
    (Philox) for 3000 seeds each -- BASELINE configs[1] shape, 'best' of 5 samples -- and the seed-averaged AUCs are compared:
    |difference| < max(0.001, 3 standard errors); likewise the seed-averaged score of every window."""
    from sklearn.metrics import roc_auc_score
    from mocodad_amd.data import synthetic
    from mocodad_amd.engine import FrameScoreAssembler
    sc, _ = _scorer()
    data, trans, meta, frames, gts = synthetic.make_dataset(n_clips=3, frames_per_clip=80, persons_per_clip=2, num_transform=2)
    N, ns, S, R, calls = data.shape[0], 10, 5, 50, 60
    asm = FrameScoreAssembler(gts, {}, num_transform=2, pad_size=-1, filter_kernel_size=3, frames_shift=2, device="cuda:0")
    dtrans, dmeta, dframes = trans.cuda(), meta.cuda(), frames.cuda()
    rep = data.cuda().repeat(R, 1, 1, 1)                       # R seeds per launch: replicas of the clip set under distinct window ids
    gen = torch.Generator(device="cuda").manual_seed(2026)
    auc = {"philox": [], "randn": []}
    mean = {"philox": torch.zeros(N, device="cuda", dtype=torch.float64), "randn": torch.zeros(N, device="cuda", dtype=torch.float64)}
    sq = {k: torch.zeros_like(v) for k, v in mean.items()}
    for c in range(calls):
        z = torch.randn(S, ns - 1, N * R, 2, 3, 17, device="cuda", generator=gen)
        runs = {"philox": sc.score(rep, n_samples=S, noise_steps=ns, seed=4242, first_window_id=c * N * R)[0],
                "randn": sc.score(rep, n_samples=S, noise_steps=ns, noise=z)[0]}
        for k, loss in runs.items():
            best = loss.min(1)[0].view(R, N)
            mean[k] += best.double().sum(0)
            sq[k] += (best.double() ** 2).sum(0)
            for r in range(R):
                auc[k].append(roc_auc_score(asm.gt, asm(best[r], dtrans, dmeta, dframes)))
    n = R * calls
    a, b = np.array(auc["philox"]), np.array(auc["randn"])
    se = np.sqrt((a.var(ddof=1) + b.var(ddof=1)) / n)
    print(f"{n} seeds each: AUC philox {a.mean():.5f} +- {a.std(ddof=1):.4f}, torch.randn {b.mean():.5f} +- {b.std(ddof=1):.4f}; "
          f"|diff of means| {abs(a.mean() - b.mean()):.5f} (se {se:.5f})")
    assert abs(a.mean() - b.mean()) < max(1e-3, 3 * se)
    mp, mr = (mean["philox"] / n).cpu().numpy(), (mean["randn"] / n).cpu().numpy()
    vp = (sq["philox"] / n).cpu().numpy() - mp ** 2
    vr = (sq["randn"] / n).cpu().numpy() - mr ** 2
    zw = (mp - mr) / np.sqrt((vp + vr) / n)
    print(f"per-window seed-averaged 'best' score: z mean {zw.mean():.3f} rms {np.sqrt((zw ** 2).mean()):.3f} max {np.abs(zw).max():.2f}; "
          f"variance ratio philox/randn {np.mean(vp / vr):.4f}")
    assert abs(zw.mean()) < 5 / np.sqrt(N) and np.sqrt((zw ** 2).mean()) < 1.2 and np.abs(zw).max() < 5
    assert abs(np.mean(vp / vr) - 1) < 0.02
