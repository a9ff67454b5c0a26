mkdir -p gpurun_out/r6n
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
for spec in "base|" "hwq8|GPU_MAX_HW_QUEUES=8" "base|" "hwq8|GPU_MAX_HW_QUEUES=8"; do
  tag=${spec%%|*}; envs=${spec#*|}
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-sclk 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('%-6s value %8.1f  %s submit %s host %s call %s' % ('$tag', d['value'], c['repeat_values']['all'], c.get('net_submit_images_per_sec'), c.get('net_submit_host_images_per_sec'), c.get('net_call_images_per_sec')))"
done 2>&1 | tee gpurun_out/r6n/ab_hwq.txt
cat gpurun_out/timing_warnings.jsonl | tail -3
