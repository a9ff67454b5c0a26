mkdir -p gpurun_out/r6l
for i in 1 2; do python bench.py --workload conv2 --steps 50 --warmup 10 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('conv2', d['value'], d['ms_per_step'], d['config']['streams'])"; done
python tools/latency_bench.py resnet18 32 2>/dev/null | tail -1
for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-sclk 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('value %8.1f  %s  call %s submit %s host %s' % (d['value'], c['repeat_values']['all'], c.get('net_call_images_per_sec'), c.get('net_submit_images_per_sec'), c.get('net_submit_host_images_per_sec')))"; done
python -m pytest tests/test_gpu_bench_parity.py tests/test_gpu_nets.py -x -q 2>&1 | tail -3
