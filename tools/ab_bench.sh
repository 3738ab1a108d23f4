#!/bin/bash
# A/B bench of tuning builds inside ONE gpurun call (boxes of the pool differ by +-2 %):
#   bash tools/ab_bench.sh mocodad_amd/libexp_a.so mocodad_amd/libexp_b.so ...   (each listed library is benched twice, interleaved)
for rep in 1 2; do
  for f in "$@"; do
    echo -n "$(basename $f): "
    MCD_LIB=$PWD/$f timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'])"
  done
done
