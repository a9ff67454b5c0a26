#!/bin/bash
# A/B of environment switches under bench.py on the GPU box: for every "tag|ENV=.. ENV=.." argument, the pipelined rate (driver's
# form) and the one-stream rate, plus the per-layer HIP-event rows whose name matches $ROWS (default: layer3's steps).
#   tools/ab_bench.sh "base|" "xcd|PLANER_HIP_EXPERIMENT=xcd=1" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
out=gpurun_out/ab; mkdir -p $out
ROWS=${ROWS:-l3}
for spec in "$@"; do
  tag=${spec%%|*}; envs=${spec#*|}
  for streams in auto 1x1; do
    env $envs PLANER_HIP_STREAMS=$streams python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra --no-sclk 2>$out/$tag.$streams.err > $out/$tag.$streams.json || tail -3 $out/$tag.$streams.err
    python - $out/$tag.$streams.json "$tag" $streams "$ROWS" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print(sys.argv[2], sys.argv[3], "FAILED", e); sys.exit(0)
rows = [r for r in d["per_layer"] if sys.argv[4] in r["layer"]]
tot = sum(r["us"] for r in d["per_layer"])
print("%-10s %-5s value %8.1f  ms/step %.4f  parity %.1e  sum(per-layer) %.1f us  frac %.4f  %s" % (
    sys.argv[2], sys.argv[3], d["value"], d["ms_per_step"], d["parity_rel_err"], tot, d["roofline"]["frac"], d["config"]["tune_source"][:40]))
if sys.argv[3] == "1x1":
    print("   " + "  ".join("%s %.1f" % (r["layer"].replace("_conv+", ""), r["us"]) for r in rows))
PY
  done
done
