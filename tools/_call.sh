for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], 'frac', r['frac'], 'b2b', r['frac_back_to_back'], 'rocprof', r['frac_rocprof'], 'marker_us', r['marker_us'], 'avg_launch_ms', r['avg_launch_ms'], r['avg_launch_ms_back_to_back'], 'eff', r['effective_frac'])"; done
python -m pytest tests/test_zz_gpu_bench_cli.py -x -q -k "default_command" 2>&1 | tail -3
cat gpurun_out/timing_warnings.jsonl 2>/dev/null | tail -2
