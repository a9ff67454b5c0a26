bash tools/profile_bench.sh r06 2>&1 | tail -2
bash tools/other_workloads.sh r06 2>&1 | tail -7
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/prof/r06_bench_driver_form.json 2> gpurun_out/prof/r06_driver_form.err; tail -2 gpurun_out/prof/r06_driver_form.err
python bench.py --steps 5000 --warmup 50 --repeats 1 --no-cpu-baseline --no-e2e --no-extra > gpurun_out/prof/r06_bench_sustained.json 2>/dev/null
bash tools/wf4_stalls.sh > /dev/null 2>&1; grep -c . gpurun_out/wf4_stalls/digest.md
PLANER_HIP_STREAMS=1x1 python bench.py --batch 256 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-e2e --no-extra --no-sclk > gpurun_out/prof/r06_bench_b256_1s.json 2>/dev/null
