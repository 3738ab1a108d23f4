#!/bin/bash
# Second call on a final library (after tools/profile_set.sh and `PMC_SET` of bench.py point at the set): the bench lines WITH the
# set's PMC files attached (roofline.traffic / roofline.pmc), 60-second sustained runs, the random-configuration sweep.
#   usage: gpurun -- bash tools/second_call.sh <tag>      -> gpurun_out/<tag>2/
TAG=${1:-r06zz}
cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG}2
mkdir -p $O
for c in avenue stc ubnormal_concat seq24 concat24 concat32; do
  timeout 600 python bench.py --config $c > $O/bench_$c.json 2>/dev/null
done
for c in avenue seq24; do
  timeout 600 python bench.py --config $c --min-seconds 60 --no-cpu-baseline --no-auc --sustained-seconds 0 --e2e-windows 0 > $O/sustained60_$c.json 2>/dev/null
done
timeout 1500 python tests/studies/random_sweep.py 2>&1 | grep -v amdgpu.ids > $O/random_sweep.txt
tail -1 $O/random_sweep.txt
python - "$O" <<'PY'
import json, sys
o = sys.argv[1]
for c in ["avenue", "stc", "ubnormal_concat", "seq24", "concat24", "concat32"]:
    d = json.loads(open(f"{o}/bench_{c}.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print(c, d["value"], r["frac"], r.get("traffic"), (r.get("pmc") or {}).get("mfma_pipe_busy_frac"), (r.get("pmc") or {}).get("refused"))
for c in ["avenue", "seq24"]:
    d = json.loads(open(f"{o}/sustained60_{c}.json").read().strip().splitlines()[-1])
    print("60 s", c, d["value"], d["roofline"]["frac"], d["steps"], d.get("gpu_after_timed_region"))
PY
