#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 summaries of the default bench.py command.
#   1. --kernel-trace --stats            -> per-kernel time (gpurun_out/prof/<tag>_kernel_stats.csv)
#   2. --kernel-trace --pmc FETCH_SIZE   -> HBM read  KB per launch   (separate pass, guide section HBM)
#   3. --kernel-trace --pmc WRITE_SIZE   -> HBM write KB per launch   (separate pass)
# and a markdown/JSON digest of 2+3.  Copy what should be judged into profiles/ afterwards.
# usage: tools/profile_bench.sh <tag> [bench.py args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r01}; shift
out=$R/gpurun_out/prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
export PLANER_HIP_TUNE_CACHE=$out/${tag}_tune_cache.txt        # all passes run the same tile plans
python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /dev/null 2>&1     # fills the cache
# (rocprofv3's dispatch interception segfaults now and then when three graphs are in flight on three
#  streams -- never without the tool -- so the pass is retried until its summary exists)
for attempt in 1 2 3 4 5 6; do
  rocprofv3 --kernel-trace --stats -d $out -o ${tag}_bench --output-format csv -- \
      python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline "$@" > $out/${tag}_bench_line_under_rocprof.json 2> $out/${tag}_stats.err
  [ -s $out/${tag}_bench_kernel_stats.csv ] && break
  echo "stats pass: attempt $attempt failed, retrying"
done
# the same with ONE stream: every launch is a full-batch layer, so a kernel's average duration here is
# directly comparable with bench.py's per-layer HIP-event times (roofline.avg_launch_ms)
PLANER_HIP_STREAMS=1x1 rocprofv3 --kernel-trace --stats -d $out -o ${tag}_bench_1stream --output-format csv -- \
    python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline "$@" > $out/${tag}_bench_1stream_line.json 2> /dev/null
# PMC passes: one stream, so every kernel runs at the full batch and bytes/launch can be checked
# against the layer's algorithmic bytes
for pm in FETCH_SIZE WRITE_SIZE; do
  PLANER_HIP_STREAMS=1x1 rocprofv3 --kernel-trace --pmc $pm -d $out -o ${tag}_pmc_$pm --output-format csv -- \
      python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > /dev/null 2> $out/${tag}_pmc_$pm.err
done
python3 - <<PY
import csv, collections, json, glob, os
out, tag = "$out", "$tag"
def load(pm):
    f = glob.glob(os.path.join(out, "%s_pmc_%s_counter_collection.csv" % (tag, pm)))
    agg, cnt = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f[0])) if f else []:
        if r["Counter_Name"] == pm:
            agg[r["Kernel_Name"]] += float(r["Counter_Value"]); cnt[r["Kernel_Name"]].add(r["Dispatch_Id"])
    return {k: (agg[k] / len(cnt[k]), len(cnt[k])) for k in agg}
def short(k):
    for t in ("void ", "(anonymous namespace)::", "HIP_vector_type<float, 4u>"):
        k = k.replace(t, "f4" if t.startswith("HIP") else "")
    return k.split("(")[0]
rd, wr = load("FETCH_SIZE"), load("WRITE_SIZE")
rows = []
for k in sorted(rd, key=lambda k: -rd[k][0] * rd[k][1]):
    rows.append({"kernel": short(k), "launches": rd[k][1], "fetch_kb_per_launch": round(rd[k][0], 1),
                 "read_mb_per_launch_corrected": round(2 * rd[k][0] / 1e3, 2),
                 "write_kb_per_launch": round(wr.get(k, (0, 0))[0], 1)})
json.dump(rows, open(os.path.join(out, tag + "_hbm_traffic.json"), "w"), indent=1)
with open(os.path.join(out, tag + "_hbm_traffic_table.md"), "w") as f:
    f.write("| kernel | launches | FETCH_SIZE KB/launch | corrected read MB/launch | WRITE_SIZE KB/launch |\n|---|---|---|---|---|\n")
    for r in rows:
        f.write("| \`%s\` | %d | %.0f | %.1f | %.0f |\n" % (r["kernel"], r["launches"], r["fetch_kb_per_launch"],
                                                        r["read_mb_per_launch_corrected"], r["write_kb_per_launch"]))
print(open(os.path.join(out, tag + "_hbm_traffic_table.md")).read())
PY
head -30 $out/${tag}_bench_kernel_stats.csv
cat $out/${tag}_bench_line_under_rocprof.json | tail -1 | cut -c1-300
