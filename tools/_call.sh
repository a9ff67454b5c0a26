mkdir -p gpurun_out/r6i
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r6i/gputest.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6i/bench.json 2> gpurun_out/r6i/bench.err; tail -2 gpurun_out/r6i/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6i/bench.json').read().strip().splitlines()[-1])
c=d['config']; rf=d['roofline']
print('value', d['value'], c['repeat_values']['all'], 'frac', rf['frac'], rf.get('frac_rocprof'), 'pcie', c.get('pcie_inclusive_images_per_sec'), 'submit', c.get('net_submit_images_per_sec'), 'call', c.get('net_call_images_per_sec'))
PY
cat gpurun_out/timing_warnings.jsonl 2>/dev/null | tail -3
