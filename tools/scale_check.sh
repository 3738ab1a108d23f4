#!/bin/bash
# The multi-GPU scaling check in one command: bench.py at N = 1, 2, 4, 8 GPUs of this node (those that exist), weak scaling
# (the driver's SCALE command: every GPU scores 1024 windows per step) and strong scaling of BASELINE configs[2] ("HR-STC ...
# sharded over clips": a FIXED batch split over the ranks), each line with per-rank kernel ms, the all-gather ms, rccl_ranks and
# scaling_efficiency = value_N / (N x value_1).      usage: bash tools/scale_check.sh [steps]       (about 30 s per N)
cd "$(dirname "$0")/.." || exit 1
STEPS=${1:-100}
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "# $NGPU GPU(s) visible"
# ... and the shape BASELINE names as the scaling run (configs[4]: seq_len 24, ns 50, 8 samples), strong: 32 768 windows per step
# split over the ranks (0.36 s per 1024 windows per GPU: 3 steps)
for mode in "weak --config avenue" "strong --config stc --batch 16384" "strong --config seq24 --batch 32768 --steps 3 --warmup 1"; do
  set -- $mode; scaling=$1; shift
  ref=""
  for n in 1 2 4 8; do
    [ "$n" -gt "$NGPU" ] && continue
    line=$(timeout 900 python bench.py --gpus $n --steps $STEPS --warmup 10 --scaling $scaling "$@" --no-cpu-baseline --no-extras ${ref:+--ref-value $ref} | tail -1)
    [ -z "$ref" ] && ref=$(echo "$line" | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "$line" | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d.get('ranks') or {}
print(f\"$scaling N={d['n_gpus']}: {d['value']:.0f} clips/s  eff {d.get('scaling_efficiency', 1.0)}  rccl_ranks {d.get('rccl_ranks', 0)}  kernel ms/rank {r.get('kernel_ms')}  all_gather ms {r.get('all_gather_ms')}\")"
  done
done
