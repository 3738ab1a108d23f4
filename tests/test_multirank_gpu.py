"""GPU: the multi-process path end to end.  On a box with ONE device two ranks share cuda:0 (RCCL refuses two ranks on one
GPU, so the collective runs on gloo there); on a box with >= 2 devices the SAME tests also run with one rank per GPU on the
nccl backend (= RCCL over xGMI) -- the configuration of the driver's 2/4/8-GPU scaling runs:
  - eval_MoCoDAD.py with WORLD_SIZE=2 gives the single-process scores and AUC bit for bit (noise keyed by global window id,
    contiguous shards, one all-gather, AUC on rank 0);
  - `python bench.py --gpus 2` launches its own ranks and prints ONE JSON line for the 2-rank job, weak and strong scaling."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    e = dict(os.environ)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e["OMP_NUM_THREADS"] = "4"
    return e


def _n_gpus():
    import torch
    return torch.cuda.device_count()


def _rank_layouts():
    """(backend, ranks): gloo with two ranks on cuda:0 always; nccl with one rank per visible GPU (up to 8) when there are >= 2."""
    return [("gloo", 2), ("nccl", None)]


def _resolve(backend, ranks):
    if backend == "nccl":
        n = min(_n_gpus(), 8)
        if n < 2:
            pytest.skip("the RCCL variant needs >= 2 GPUs (one rank per GPU); this box has %d" % n)
        return n
    return ranks


@pytest.mark.parametrize("backend,ranks", _rank_layouts())
@pytest.mark.parametrize("extra", [[], ["--device-windows"]])
def test_eval_two_ranks_equals_one_process(tmp_path, extra, backend, ranks):
    ranks = _resolve(backend, ranks)
    cfg = os.path.join(ROOT, "configs", "hr_avenue_test.yaml")
    common = ["-c", cfg, "--synthetic", "3", "--frames-per-clip", "90", "--random-init"] + extra
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "eval_MoCoDAD.py")] + common + ["--dump-scores", one],
                        capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr", "127.0.0.1",
                         "--master-port", str(_port()), os.path.join(ROOT, "eval_MoCoDAD.py")] + common +
                        ["--dist-backend", backend, "--dump-scores", two], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    a, b = np.load(one), np.load(two)
    assert a["scores"].shape == b["scores"].shape and a["scores"].size > 1000
    assert np.array_equal(a["scores"], b["scores"])
    assert float(a["auc"]) == float(b["auc"])
    assert r2.stdout.count("AUC:") == 1          # rank 0 alone computes and prints it


@pytest.mark.parametrize("backend,ranks", _rank_layouts())
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_self_launches_two_ranks(scaling, backend, ranks):
    ranks = _resolve(backend, ranks)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--batch", "256",
                        "--dist-backend", backend, "--scaling", scaling], capture_output=True, text=True, timeout=600, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 3 and d["scaling"] == scaling and d["value"] > 0
    assert len(d["ranks"]["kernel_ms"]) == ranks and len(d["ranks"]["all_gather_ms"]) == ranks
    for key in ("first_step_ms", "min_step_ms", "max_step_ms", "clock_mhz"):      # what makes a slow rank attributable
        assert len(d["ranks"][key]) == ranks
    assert all(lo <= hi for lo, hi in zip(d["ranks"]["min_step_ms"], d["ranks"]["max_step_ms"]))
    total = 256 * ranks if scaling == "weak" else 256
    assert d["config"]["windows_per_step_total"] == total
    assert sum(d["ranks"]["windows_per_step"]) == total
    assert d["rccl_ranks"] == (ranks if backend == "nccl" else 0)
    assert "roofline" in d and "cpu_baseline" not in d


def test_nccl_backend_refuses_more_ranks_than_gpus():
    """One rank per GPU is required with RCCL; asking for more must fail fast (exit code 2 / a message), not hang."""
    n = _n_gpus()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "1", "--warmup", "0", "--batch", "64"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0 and "one rank per GPU" in (r.stdout + r.stderr)
    cfg = os.path.join(ROOT, "configs", "hr_avenue_test.yaml")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n + 1}", "--master-addr", "127.0.0.1",
                        "--master-port", str(_port()), os.path.join(ROOT, "eval_MoCoDAD.py"), "-c", cfg, "--synthetic", "1", "--random-init"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0 and "one rank per GPU" in (r.stdout + r.stderr)


def test_missing_checkpoint_is_an_error():
    cfg = os.path.join(ROOT, "configs", "hr_avenue_test.yaml")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "eval_MoCoDAD.py"), "-c", cfg, "--synthetic", "1"],
                       capture_output=True, text=True, timeout=300, env=_env(), cwd=ROOT)
    assert r.returncode != 0 and "--random-init" in (r.stdout + r.stderr)


def test_scale_check_script_output_format():
    """tools/scale_check.sh on the GPUs that exist (one here): a header and one line per (scaling mode, N) in the format the
    8-GPU run will be read in -- the script must not rot before such a box appears."""
    import re
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "scale_check.sh"), "5"], capture_output=True, text=True, timeout=900, env=_env(), cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    n = _n_gpus()
    per_mode = sum(1 for k in (1, 2, 4, 8) if k <= n)
    assert lines[0] == f"# {n} GPU(s) visible" and len(lines) == 1 + 3 * per_mode, r.stdout
    pat = re.compile(r"^(weak|strong) N=(\d+): (\d+) clips/s  eff ([0-9.]+)  rccl_ranks (\d+)  kernel ms/rank (None|\[[0-9., ]+\])  all_gather ms (None|\[[0-9., ]+\])$")
    seen = []
    for l in lines[1:]:
        m = pat.match(l)
        assert m, l
        seen.append((m.group(1), int(m.group(2))))
        assert int(m.group(3)) > 0 and (int(m.group(2)) > 1 or float(m.group(4)) == 1.0)
    # weak (avenue), strong (stc, 16 384 windows), strong (seq24 = BASELINE configs[4], the designated scaling shape)
    assert seen == [(mode, k) for mode in ("weak", "strong", "strong") for k in (1, 2, 4, 8) if k <= n]
