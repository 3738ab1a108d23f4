mkdir -p gpurun_out/r05r
timeout 1500 python -m pytest tests/test_extra6_gpu.py tests/test_hip_parity.py tests/test_random_configs_gpu.py tests/test_layers_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05r/pytest_tiled.txt
cat gpurun_out/r05r/pytest_tiled.txt
for c in seg32 concat24 concat32 seg32_eunet seg20 seg14; do
  echo -n "$c: "; timeout 300 python bench.py --config $c --no-cpu-baseline --no-extras | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%10.1f clips/s  frac %.4f  kernel ms/step %8.4f' % (d['value'], r['frac'], r['kernel_ms_per_step']))"
done > gpurun_out/r05r/shapes_tiled.txt 2>&1
cat gpurun_out/r05r/shapes_tiled.txt
