#!/bin/bash
# A/B bench of tuning builds inside ONE gpurun call (boxes of the pool differ by +-2 %):
#   bash tools/ab_bench.sh "<bench args>" mocodad_amd/libA.so mocodad_amd/libmocodad_hip.so ...   (each library benched 3x, interleaved)
args="$1"; shift
for rep in 1 2 3; do
  for f in "$@"; do
    echo -n "$(basename $f) [$args]: "
    MCD_LIB=$PWD/$f timeout 300 python bench.py $args --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
  done
done
