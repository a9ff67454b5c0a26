#!/bin/bash
# A/B of two builds of the library on the GPU box: isolated fused-conv times and bench.py pairs.
#   tools/ab_lib.sh <alt .so> [pairs = 3]      (the shipped library against <alt>, selected through PLANER_HIP_LIB)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
alt=$PWD/$1; pairs=${2:-3}
out=gpurun_out/ab_lib.txt; mkdir -p gpurun_out; : > $out
for lib in "" $alt; do
  echo "== ${lib:-shipped build}" >> $out
  env ${lib:+PLANER_HIP_LIB=$lib} python tools/wf4_bench.py --shapes 64x56,128x28,256x14 --algos 9 >> $out 2>&1
done
for rep in $(seq $pairs); do
  for lib in "" $alt; do
    env ${lib:+PLANER_HIP_LIB=$lib} python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-extra --no-sclk 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s value %8.1f  %s  parity %.1e' % ('${lib:+alt}' or 'shipped', d['value'], d['config']['repeat_values']['all'], d['parity_rel_err']))" >> $out
  done
done
cat $out
