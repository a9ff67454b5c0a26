#!/bin/bash
# Probe builds of the fused F(4x4,3x3) kernel: one library per (tag, compiler flags) pair under planer_amd/build/knock/, e.g.
# knock-out masks (-DWF4_KNOCK=<mask>: bit 0 no filter loads, 1 no patch LDS-DMA, 2 no patch transform, 3 no MFMAs, 4 no V
# fragment reads, 5 no output rows); "run" times each with the K sweep (tools/wf4_ksweep.py --tail).
#   tools/wf4_knock.sh build base "" k4 "-DWF4_KNOCK=4" ...     (here)
#   tools/wf4_knock.sh run                                       (on the GPU box; writes gpurun_out/wf4_knock.txt)
set -e
cd "$(dirname "$0")/.."
D=planer_amd/build/knock
mkdir -p $D
if [ "$1" = build ]; then
  shift
  while [ $# -ge 2 ]; do
    tag=$1; flags=$2; shift 2
    ( /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -ffp-contract=off \
        $flags -c planer_amd/csrc/conv_winograd.hip -o $D/cw_$tag.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libk_$tag.so planer_amd/build/runtime.o planer_amd/build/pointwise.o \
        planer_amd/build/head_ops.o planer_amd/build/conv_direct.o $D/cw_$tag.o -ldl && echo built $tag &&
      tools/kernel_resources.sh "wf4_kernelI.*4E" $D/cw_$tag.o | sed 's/^_ZN12_GLOBAL__N_1//' ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait
else
  mkdir -p gpurun_out
  : > gpurun_out/wf4_knock.txt
  for f in $D/libk_*.so; do
    echo "== $f" >> gpurun_out/wf4_knock.txt
    PLANER_HIP_LIB=$PWD/$f timeout 300 python tools/wf4_ksweep.py --tail 2>&1 | tail -4 >> gpurun_out/wf4_knock.txt
  done
  cat gpurun_out/wf4_knock.txt
fi
