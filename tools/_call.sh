mkdir -p gpurun_out/r6k
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for spec in "base|" "stem7|PLANER_HIP_STEM_ROWS14=0" "pipe15|PLANER_HIP_STREAMS=pipe15" "pipe5|PLANER_HIP_STREAMS=pipe5" "base|" "pipe11|PLANER_HIP_STREAMS=pipe11" "pipe9|PLANER_HIP_STREAMS=pipe9"; do
  tag=${spec%%|*}; envs=${spec#*|}
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra --no-sclk 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s value %8.1f  %s %s' % ('$tag', d['value'], d['config']['repeat_values']['all'], d['config']['streams']))"
done 2>&1 | tee gpurun_out/r6k/ab_env.txt
