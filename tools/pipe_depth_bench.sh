#!/bin/bash
# bench.py's own timed regions per pipeline depth (PLANER_HIP_STREAMS=pipeN): the driver's --steps 20 --warmup 5 and the
# default 50 / 10 for ResNet-18, the defaults for the other workloads.
#   tools/pipe_depth_bench.sh [resnet18|yolov3|conv2] pipe3 pipe7 pipe15 ...      -> gpurun_out/pipe_depth_<workload>.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
w=$1; shift
out=gpurun_out/pipe_depth_$w.txt
: > $out
for s in "$@"; do
  for a in "--steps 20 --warmup 5" "--steps 50 --warmup 10"; do
    [ $w != resnet18 ] && [ "$a" = "--steps 20 --warmup 5" ] && continue
    PLANER_HIP_STREAMS=$s python bench.py --workload $w $a --no-cpu-baseline --no-e2e 2>/dev/null | tail -1 > /tmp/line.json
    python - "$s" "$a" >> $out <<'PY'
import json, sys
d = json.load(open("/tmp/line.json"))
c = d["config"]
print(sys.argv[1], sys.argv[2], "value", d["value"], d["unit"], "repeats", c.get("repeat_values", {}).get("all"), "streams", c.get("streams"),
      "parity", d.get("parity_rel_err"))
PY
  done
done
cat $out
