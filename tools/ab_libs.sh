#!/bin/bash
# A/B of built libraries on ONE box: bash tools/ab_libs.sh "<config list>" libA.so libB.so ...   (each config x lib, 2 repetitions)
cfgs="$1"; shift
for rep in 1 2; do
  for c in $cfgs; do
    for l in "$@"; do
      echo -n "[$c $l]: "
      MCD_LIB=$PWD/mocodad_amd/$l timeout 600 python bench.py --config $c --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
    done
  done
done
