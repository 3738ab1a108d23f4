import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


# ---- the parity that was MEASURED, not just gated: on a GPU box every np.testing.assert_allclose of the session is recorded
# (test id, quantity, max |got - ref|, max |ref|, the gate) and written to gpurun_out/parity_errors.txt at the end; the copy the
# round was judged on is tracked as profiles/r05_parity_errors.txt.
@pytest.fixture(scope="session", autouse=True)
def _parity_error_log(request):
    try:
        import torch
        on_gpu = torch.cuda.is_available()
    except Exception:
        on_gpu = False
    if not on_gpu:
        yield
        return
    rec = {}
    orig = np.testing.assert_allclose

    def recording(actual, desired, rtol=1e-7, atol=0, **kw):
        try:
            a, d = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
            if a.shape == d.shape and a.size:
                test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0].replace("tests/", "")
                key = (test, str(kw.get("err_msg", "") or ""))
                err, ref = float(np.nanmax(np.abs(a - d))), float(np.nanmax(np.abs(d)))
                e = rec.setdefault(key, [0.0, 0.0, float("inf"), 0.0, 0, 0])
                e[0], e[1], e[2], e[3], e[4], e[5] = max(e[0], err), max(e[1], ref), min(e[2], float(atol)), max(e[3], float(rtol)), e[4] + 1, e[5] + a.size
        except Exception:
            pass
        return orig(actual, desired, rtol=rtol, atol=atol, **kw)

    np.testing.assert_allclose = recording
    yield
    np.testing.assert_allclose = orig
    if not rec:
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_errors.txt"), "w") as f:
        f.write("# every np.testing.assert_allclose of this `pytest -m gpu` session: HIP path (or oracle) vs reference-generated vectors / oracle\n")
        f.write("# columns: max|got-ref|  max|ref|  err/max(1,|ref|)  atol(min)  rtol(max)  comparisons  elements  test :: quantity\n")
        worst = 0.0
        for (test, what), (err, ref, atol, rtol, n, el) in sorted(rec.items()):
            rel = err / max(1.0, ref)
            worst = max(worst, rel)
            f.write(f"{err:10.3e} {ref:10.3e} {rel:10.3e} {atol:9.1e} {rtol:9.1e} {n:5d} {el:9d}  {test} :: {what}\n")
        f.write(f"# {len(rec)} rows; largest error relative to max(1, |ref|): {worst:.3e}\n")
