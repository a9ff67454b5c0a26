mkdir -p gpurun_out/r6c
( python -m pytest tests/test_gpu_host_path.py -x -q 2>&1 | tail -3
for spec in "direct_w6|" "direct_w12|WINDOW=12" "plain_w12|PLANER_HIP_HOST_ROUTE=plain WINDOW=12" "staged_w12|PLANER_HIP_HOST_ROUTE=staged WINDOW=12" "direct_w12_q8|WINDOW=12 GPU_MAX_HW_QUEUES=8" "direct_prof|PROFILE=1 WINDOW=12"; do
  tag=${spec%%|*}; envs=${spec#*|}
  env $envs TAG=$tag python tools/host_submit_probe.py 2>&1 | grep -v "^$" | tail -22
done ) | tee gpurun_out/r6c/host_submit_probe4.txt
