mkdir -p gpurun_out/r6q
STREAMS=pipe7 timeout 1300 python tools/pipeline_plan_search.py gpurun_out/r6q/plan_search_cache.txt > gpurun_out/r6q/plan_search_pipe7.txt 2>&1
grep -c "img/s" gpurun_out/r6q/plan_search_pipe7.txt; grep "kept\|after search\|isolated plans" gpurun_out/r6q/plan_search_pipe7.txt | tail -12
