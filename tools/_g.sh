mkdir -p gpurun_out/r05zf
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/r05zf/pytest_gpu.txt
cat gpurun_out/r05zf/pytest_gpu.txt
cp gpurun_out/parity_errors.txt gpurun_out/r05zf/parity_errors.txt
bash tools/profile_set.sh r05zz '' d64baa8 > gpurun_out/profile_set.log 2>&1; tail -3 gpurun_out/profile_set.log
