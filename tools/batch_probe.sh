mkdir -p gpurun_out/r3h
for cfg in "32 1x1" "64 1x1" "96 1x1" "128 1x1" "32 pipe3" "64 pipe3" "96 pipe3" "64 pipe2"; do
  set -- $cfg
  BATCH=$1 STREAMS=$2 STEPS=60 TAG="b$1-$2" python tools/throughput_probe.py 2>&1 | tail -1
done > gpurun_out/r3h/batch_probe.log 2>&1
cat gpurun_out/r3h/batch_probe.log
