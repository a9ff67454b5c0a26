#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 evidence for the default bench.py command, trial-free.
#   0. untraced run of the default command (kernel choices: the shipped tuning database, planer_amd/tuned/).
#   1. --kernel-trace --stats of the DEFAULT command (pipelined: three streams)      -> <tag>_bench_kernel_stats.csv
#   2. the same with ONE stream (every launch is one full-batch layer)               -> <tag>_bench_1stream_*  + per-layer table
#   3. --pmc passes, one stream, each in its own run (guide: FETCH_SIZE and WRITE_SIZE cannot share a pass):
#      FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 ... GRBM_GUI_ACTIVE
#   4. tools/profile_digest.py: per-kernel MFMA utilisation, HBM bytes per launch, per-layer CSV.
# Copy what should be judged from gpurun_out/prof into profiles/ afterwards (tools/profile_digest.py --install).
# usage: tools/profile_bench.sh <tag> [bench.py args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r03}; shift
out=$R/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# Kernel choices come from the database shipped in planer_amd/tuned/ (loaded by default): every pass below, and the
# driver's own run of bench.py, launch the same kernels -- each line's config.tune_source says so ("shipped").  Only
# when that database does not cover this device does a run fall back to timing, and then a private cache keeps the
# passes consistent with one another.
if ! ls $R/planer_amd/tuned/*.plans > /dev/null 2>&1; then export PLANER_HIP_TUNE_CACHE=$out/${tag}_tune_cache.txt; fi
python $R/bench.py --steps 50 --warmup 10 "$@" > $out/${tag}_bench_line.json 2> $out/${tag}_fill.err      # fills the caches; the untraced line
# (the traced passes skip the secondary workloads bench.py appends -- extra.yolov3_b1 keeps fifteen graphs in flight, which the
#  tool's dispatch interception does not survive: HSA_STATUS_ERROR_INVALID_PACKET_FORMAT -- and the host legs)
# (every traced pass runs under `timeout`: a pass that aborts inside the tool can otherwise sit on its dispatches for minutes)
# (rocprofv3's dispatch interception segfaults now and then when three graphs are in flight on three
#  streams -- never without the tool -- so the pass is retried until its summary exists)
for attempt in 1 2 3 4; do
  PLANER_HIP_STREAMS=1x1 timeout 150 rocprofv3 --kernel-trace --stats -d $out -o ${tag}_bench_1stream --output-format csv -- \
      python $R/bench.py --steps 50 --warmup 10 --repeats 1 --settle-ms 50 --no-cpu-baseline --no-e2e --no-extra "$@" > $out/${tag}_bench_1stream_line.json 2> $out/${tag}_1stream.err
  [ -s $out/${tag}_bench_1stream_line.json ] && [ -s $out/${tag}_bench_1stream_kernel_trace.csv ] && break
  echo "one-stream pass: attempt $attempt failed, retrying"; tail -3 $out/${tag}_1stream.err
done
i=0
for pm in "FETCH_SIZE" "WRITE_SIZE" \
          "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  PLANER_HIP_STREAMS=1x1 timeout 150 rocprofv3 --kernel-trace --pmc $pm -d $out -o ${tag}_pmc$i --output-format csv -- \
      python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-e2e --no-extra "$@" > /dev/null 2> $out/${tag}_pmc$i.err
done
# the pipelined command last (seven graphs in flight: the pass most likely to die inside the tool)
for attempt in 1 2; do
  timeout 150 rocprofv3 --kernel-trace --stats -d $out -o ${tag}_bench --output-format csv -- \
      python $R/bench.py --steps 50 --warmup 10 --no-extra --no-e2e --no-cpu-baseline "$@" > $out/${tag}_bench_line_under_rocprof.json 2> $out/${tag}_stats.err
  [ -s $out/${tag}_bench_kernel_stats.csv ] && break
  echo "stats pass: attempt $attempt failed, retrying"
done
python3 $R/tools/profile_digest.py $out $tag
