mkdir -p gpurun_out/r05v
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05v/pytest_gpu.txt
cat gpurun_out/r05v/pytest_gpu.txt
cp gpurun_out/parity_errors.txt gpurun_out/r05v/parity_errors.txt 2>/dev/null
python bench.py > gpurun_out/r05v/bench_default.json 2> gpurun_out/r05v/bench_default.err; tail -c 1200 gpurun_out/r05v/bench_default.json
for c in ubnormal_concat seq24; do python bench.py --config $c --no-cpu-baseline --no-extras > gpurun_out/r05v/bench_$c.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r05v/bench_$c.json').read().strip().splitlines()[-1]);print('$c',d['value'],d['roofline']['frac'])"; done
