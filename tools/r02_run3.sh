# round 2, GPU call 3: full GPU test suite + bench lines of the four configurations + kernel trace of the default one
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -E "passed|failed|rc=|FAILED|Error" $O/pytest.log | tail -30
for c in avenue stc ubnormal_concat seq24; do
  st=20; [ $c = seq24 ] && st=5
  timeout 400 python bench.py --config $c --steps $st > $O/bench_$c.json 2> $O/bench_$c.err
  python - <<PY
import json
d=json.load(open('$O/bench_$c.json')); print('$c', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_step'], d.get('value_incl_h2d',{}).get('value'))
PY
done
timeout 300 python bench.py --batch 4096 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B4096', d['value'], d['roofline']['frac'])"
timeout 300 python bench.py --batch 1000 --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('B1000', d['value'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_avenue -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $O/prof_avenue.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/prof_avenue -name "*_results.db" | head -1) > $O/avenue_kernel_stats.txt
rm -rf $O/prof_avenue
cat $O/avenue_kernel_stats.txt | head -12
