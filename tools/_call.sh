python -m pytest tests/test_gpu_wf4.py -x -q 2>&1 | tail -3
bash tools/wf4_tail_ab.sh run 2>&1 | tail -30
