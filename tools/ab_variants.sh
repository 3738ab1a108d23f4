#!/bin/bash
# A/B of bench.py argument sets on ONE box (boxes of the pool differ by +-2 %), each 3x, interleaved:
#   bash tools/ab_variants.sh "<common args>" "<args A>" "<args B>" ...
common="$1"; shift
for rep in 1 2 3; do
  for a in "$@"; do
    echo -n "[$common $a]: "
    timeout 600 python bench.py $common $a --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'])"
  done
done
