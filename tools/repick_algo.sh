#!/bin/bash
# Run ON THE GPU BOX (via gpurun): re-pick the conv algorithm of the shipped database's entries whose input map has one of the
# given heights (e.g. after a new algorithm became a candidate for those maps), everything else staying as shipped.
# The box's scratch copy of planer_amd/tuned/<stem>.algo.json loses those entries; ResNet-18 at the given batches then
# compiles on top of the shipped database with a user cache, which ends up holding the merged database:
#   gpurun_out/tuned/<stem>.plans  and  .algo.json     (copy both into planer_amd/tuned/ afterwards)
#   usage: tools/repick_algo.sh "14" "8 16 32 64 256"
R=${GRAFT_REPO_ROOT:-/root/repo}
heights=${1:-14}; batches=${2:-"8 16 32 64 256"}
out=$R/gpurun_out/tuned
mkdir -p $out
cd $R
stem=$(python -c "import planer_amd; c = planer_amd.hip.context(); print('%s_cu%d' % (c.arch.split(':')[0], c.cu_count))")
python - "$R/planer_amd/tuned/$stem.algo.json" $heights <<'PY'
import ast, json, sys
path, hs = sys.argv[1], [int(h) for h in sys.argv[2:]]
d = json.load(open(path))
keep = {k: v for k, v in d["algo"].items() if ast.literal_eval(k)[1][2] not in hs}
print("dropped", len(d["algo"]) - len(keep), "picks of", len(d["algo"]))
d["algo"] = keep
json.dump(d, open(path, "w"), indent=1)
PY
export PLANER_HIP_TUNE_CACHE=$out/$stem.plans PLANER_CONV_TUNE_LOG=1
rm -f $out/$stem.plans $out/$stem.plans.algo.json
: > $out/repick.jsonl
for b in $batches; do python tools/tune_fill.py resnet18 $b 2>>$out/repick.err | tail -1 >> $out/repick.jsonl; done
cat $out/repick.jsonl
grep "w_layout" $out/repick.err | sort | uniq | head -80
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra > $out/check_resnet18.json 2> $out/check_resnet18.err
mv $out/$stem.plans.algo.json $out/$stem.algo.json
python - <<PY
import json
d = json.load(open("$out/check_resnet18.json"))
print("check run:", d["value"], d["config"]["tune_source"], d["config"]["streams"])
PY
wc -l $out/$stem.plans
