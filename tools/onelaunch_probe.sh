cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "--split 5" "--split 1" "--split 1 --cond-generic"; do
  tag=$(echo $v | tr -d ' -')
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -- python $R/bench.py --config avenue $v --no-cpu-baseline --no-extras > /tmp/prof_$tag.log 2>&1
  echo "== $v"; tail -1 /tmp/prof_$tag.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step'])"
  python $R/tools/rocpd_summary.py $(find /tmp/prof_$tag -name "*_results.db" | head -1) | head -6 | cut -c1-110
done
