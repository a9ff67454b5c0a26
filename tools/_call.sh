python -m pytest tests/test_gpu_bench_parity.py -x -q 2>&1 | tail -3
