python -m pytest tests/test_zz_gpu_bench_cli.py -x -q -k "two_ranks_share or two_gpus" 2>&1 | tail -3
