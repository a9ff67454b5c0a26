mkdir -p gpurun_out/r6a
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6a/bench_base.json 2> gpurun_out/r6a/bench_base.err
tail -c 600 gpurun_out/r6a/bench_base.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6a/bench_base.json').read().strip().splitlines()[-1])
print('BASE value', d['value'], d['ms_per_step'], d['config'].get('streams'), d['roofline']['frac'], d['config'].get('pcie_inclusive_images_per_sec'), d['config'].get('net_submit_images_per_sec'))
PY
PLANER_HIP_STREAMS=1x1 python bench.py --batch 256 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-e2e --no-extra --no-sclk > gpurun_out/r6a/bench_b256_1s.json 2> gpurun_out/r6a/bench_b256_1s.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6a/bench_b256_1s.json').read().strip().splitlines()[-1])
print('B256 1-stream value', d['value'], d['ms_per_step'])
tot=0
for r in d['per_layer']:
    print('  %-28s %8.1f us per 256 = %6.2f per 32  %s' % (r['layer'], r['us'], r['us']/8, r.get('kernel','')[:60])); tot+=r['us']
print('sum per 32:', tot/8)
PY
bash tools/wf4_stalls.sh
