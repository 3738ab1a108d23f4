timeout 1200 python -m pytest tests/test_layers_gpu.py tests/test_extra6_gpu.py -m gpu -q 2>&1 | tail -4
