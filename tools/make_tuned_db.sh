#!/bin/bash
# Run ON THE GPU BOX (via gpurun): build the tuning database that ships in planer_amd/tuned/ for this device.
# Every BASELINE workload is compiled once with the shipped database switched off, so all launch plans (C side),
# conv algorithms and stream plans (planer_amd.net) are found by measurement and written to
#   gpurun_out/tuned/<arch>_cu<CUs>.plans  and  .algo.json
# Copy both into planer_amd/tuned/ afterwards (the files are small text).  Runs of this build then load them by
# default and time nothing: bench.py's config.tune_source says "shipped".
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/tuned
mkdir -p $out
cd $R
stem=$(python -c "import planer_amd; c = planer_amd.hip.context(); print('%s_cu%d' % (c.arch.split(':')[0], c.cu_count))")
export PLANER_HIP_TUNED=0 PLANER_HIP_TUNE_CACHE=$out/$stem.plans
rm -f $out/$stem.plans $out/$stem.plans.algo.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/fill_resnet18.json 2> $out/fill_resnet18.err
python bench.py --workload yolov3 --steps 20 --warmup 5 > $out/fill_yolov3.json 2> $out/fill_yolov3.err
python bench.py --workload conv2 --steps 20 --warmup 5 > $out/fill_conv2.json 2> $out/fill_conv2.err
python tools/latency_bench.py > $out/fill_latency.log 2>&1
# more shapes than the three BASELINE workloads: other batch sizes of ResNet-18, other resolutions of YOLO-v3 (latency and
# throughput plan each; tools/tune_fill.py prints the cold compile time of each)
: > $out/fill_more.jsonl
for b in 1 8 16 64 256; do python tools/tune_fill.py resnet18 $b 2>>$out/fill_more.err | tail -1 >> $out/fill_more.jsonl; done
for sz in 320 608; do python tools/tune_fill.py yolov3 1 $sz 2>>$out/fill_more.err | tail -1 >> $out/fill_more.jsonl; done
python tools/tune_fill.py customnet 1 2>>$out/fill_more.err | tail -1 >> $out/fill_more.jsonl      # BASELINE config 1 (bench.py's extra.customnet_b1)
cat $out/fill_more.jsonl
# a second pass must find everything: its tune_source may not mention "autotuned"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $out/check_resnet18.json 2> $out/check_resnet18.err
# picks judged under the seven-replica pipeline (table "algo_throughput": asked first by throughput plans);
# the search reads and extends the cache this script has just filled
STREAMS=pipe7 python tools/pipeline_search.py --cands 2,8,4,7,9,11 --write $out/$stem.plans.algo.json > $out/pipeline_search.log 2>&1; tail -2 $out/pipeline_search.log
mv $out/$stem.plans.algo.json $out/$stem.algo.json
python - <<PY
import json
d = json.load(open("$out/check_resnet18.json"))
print("check run:", d["value"], d["config"]["tune_source"], d["config"]["streams"])
PY
wc -l $out/$stem.plans; cat $out/$stem.algo.json | head -40
