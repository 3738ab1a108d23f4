mkdir -p gpurun_out/r05zz2
python bench.py > gpurun_out/r05zz2/bench_avenue.json 2>/dev/null
for c in stc ubnormal_concat seq24 concat24 concat32; do python bench.py --config $c --no-cpu-baseline > gpurun_out/r05zz2/bench_$c.json 2>/dev/null; done
for c in avenue stc ubnormal_concat seq24 concat24 concat32; do python - <<PY
import json
d=json.loads(open('gpurun_out/r05zz2/bench_$c.json').read().strip().splitlines()[-1])
r=d['roofline']; print('$c', d['value'], r['frac'], 'traffic', r.get('traffic'), (r.get('pmc') or {}).get('mfma_pipe_busy_frac'), (r.get('pmc') or {}).get('refused'))
PY
done
timeout 1200 python tests/studies/random_sweep.py 200 120 > gpurun_out/r05zz2/random_sweep.txt 2>&1; tail -1 gpurun_out/r05zz2/random_sweep.txt
