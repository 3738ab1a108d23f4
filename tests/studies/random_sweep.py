"""One-off widening of tests/test_random_configs_gpu.py: the same draws / checks over seed ranges the suite does not hold.
usage (GPU box): python tests/studies/random_sweep.py [n_short] [n_long]   -> one line per failure + a summary line"""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_random_configs_gpu as R

n_short = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_long = int(sys.argv[2]) if len(sys.argv) > 2 else 200
seeds = list(range(40, 40 + n_short)) + [-25 - k for k in range(n_long)]
t0 = time.time()
bad = []
for i, s in enumerate(seeds):
    try:
        R.test_random_configuration_vs_oracle(s)
    except Exception as e:      # noqa: BLE001  (a study: report and go on)
        c = R._draw(s) if s >= 0 else R._draw_long(-s)
        bad.append(s)
        print(f"FAIL seed {s}: {c}\n  {type(e).__name__}: {str(e)[:400]}", flush=True)
        if os.environ.get("SWEEP_TRACE"):
            traceback.print_exc()
print(f"random sweep: {len(seeds)} configurations ({n_short} short, {n_long} long windows), {len(bad)} failed {bad}, {time.time() - t0:.0f} s")
