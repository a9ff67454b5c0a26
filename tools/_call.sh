mkdir -p gpurun_out/prof
python bench.py --steps 50 --warmup 10 > gpurun_out/prof/r06_bench_line.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof/r06_bench_driver_form.json 2>/dev/null
python - <<'PY'
import json
for f in ('r06_bench_line','r06_bench_driver_form'):
    d=json.loads(open('gpurun_out/prof/%s.json'%f).read().strip().splitlines()[-1]); r=d['roofline']; c=d['config']
    print(f, d['value'], d['ms_per_step'], c['repeat_values']['all'], r['frac'], r['frac_back_to_back'], r['frac_rocprof'], c['pcie_inclusive_images_per_sec'], c['net_submit_images_per_sec'], c['net_call_images_per_sec'], c['net_call_host_images_per_sec'], r['whole_forward_timed_run']['mfma_util'])
PY
