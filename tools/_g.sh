mkdir -p gpurun_out/r05zc
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05zc/pytest_gpu.txt
cat gpurun_out/r05zc/pytest_gpu.txt
cp gpurun_out/parity_errors.txt gpurun_out/r05zc/parity_errors.txt 2>/dev/null
for c in seg32 concat24 concat32 seg32_eunet; do
  echo -n "$c: "; timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%10.1f clips/s  frac %.4f  kernel ms/step %8.4f' % (d['value'], r['frac'], r['kernel_ms_per_step']))"
done > gpurun_out/r05zc/shapes_tiled.txt 2>&1
cat gpurun_out/r05zc/shapes_tiled.txt
