#!/bin/bash
# Run ON THE GPU BOX: the secondary workload lines of this build -> gpurun_out/prof/<tag>_other_workloads.jsonl
# (BASELINE config 2 single conv, config 5 YOLO-v3 at batch 1 pipelined and one image at a time, ResNet-18 on one stream
# and one batch at a time).  Kernel choices: the shipped tuning database.
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r03}
out=$R/gpurun_out/prof
mkdir -p $out
f=$out/${tag}_other_workloads.jsonl
: > $f
cd $R
python bench.py --workload conv2 --steps 50 --warmup 10 2>/dev/null | tail -1 >> $f
PLANER_HIP_STREAMS=1x1 python bench.py --workload conv2 --steps 50 --warmup 10 2>/dev/null | tail -1 >> $f
python bench.py --workload yolov3 --steps 50 --warmup 10 2>/dev/null | tail -1 >> $f
python tools/latency_bench.py yolov3 1 2>/dev/null | tail -1 >> $f
python tools/latency_bench.py resnet18 32 2>/dev/null | tail -1 >> $f
PLANER_HIP_STREAMS=1x1 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({'workload': 'resnet18 batch 32, ONE stream (throughput mode)', 'value': d['value'], 'ms_per_step': d['ms_per_step'],
                  'tune_source': d['config']['tune_source'], 'roofline_frac': d['roofline']['frac'], 'by_kernel': d['roofline']['by_kernel']}))" >> $f
python - <<PY
import json
for ln in open("$f"):
    d = json.loads(ln)
    print({k: d[k] for k in d if k in ("metric", "workload", "value", "ms_per_step", "latency_ms_median", "images_per_sec_one_at_a_time", "streams", "batch")},
          d.get("config", {}).get("streams"), d.get("config", {}).get("tune_source"))
PY
