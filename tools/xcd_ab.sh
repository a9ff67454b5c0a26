#!/bin/bash
# Run ON THE GPU BOX: evidence for VERDICT round 4 item 1 (tie GEMM -> chain -> next GEMM of a tile block to one XCD).
#   1. tools/ubench/xcd_handoff.hip: what a same-XCD / other-XCD hand-off costs for tensors laid out like V / M
#   2. bench.py on one stream with PLANER_HIP_EXPERIMENT=xcd=0 / 1 (the layer2 / layer4 F(4x4) convs: GEMM column tiles and
#      chain workgroups of N / 8 images per XCD): HIP-event us per step, and FETCH_SIZE per launch of the chain / GEMM kernels
#      from a --pmc pass each (corrected x 2 per the MI355X guide)
# -> gpurun_out/xcd/summary.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/xcd; mkdir -p $out
cd $R/tools/ubench && mkdir -p bin && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o bin/xcd_handoff xcd_handoff.hip 2> $out/build.err
{ echo "== ubench: plain"; timeout 120 bin/xcd_handoff; echo "== ubench: --dirty 40 --both"; timeout 120 bin/xcd_handoff --dirty 40 --both; } > $out/ubench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for x in 0 1; do
  PLANER_HIP_EXPERIMENT=xcd=$x PLANER_HIP_STREAMS=1x1 python $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-e2e --no-extra --no-sclk > $out/line_xcd$x.json 2> $out/line_xcd$x.err
  for attempt in 1 2 3; do
    PLANER_HIP_EXPERIMENT=xcd=$x PLANER_HIP_STREAMS=1x1 timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out -o pmc_xcd$x --output-format csv -- \
        python $R/bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-e2e --no-extra --no-sclk > /dev/null 2> $out/pmc_xcd$x.err
    [ -s $out/pmc_xcd${x}_counter_collection.csv ] && break
  done
done
python3 - $out <<'PY' > $out/summary.txt
import csv, json, re, sys, collections
out = sys.argv[1]
print(open(out + "/ubench.txt").read())
for x in (0, 1):
    d = json.loads(open("%s/line_xcd%d.json" % (out, x)).read().strip().splitlines()[-1])
    rows = [r for r in d["per_layer"] if r["layer"].startswith(("l2", "l4")) and "&" not in r["layer"]]
    print("xcd=%d one-stream %.1f img/s; layer2 + layer4 staged steps, us (HIP events): %s" % (
        x, d["value"], "  ".join("%s %.1f" % (r["layer"].replace("_conv+", ""), r["us"]) for r in rows)))
    acc = collections.defaultdict(lambda: [0, 0.0])
    try:
        for r in csv.DictReader(open("%s/pmc_xcd%d_counter_collection.csv" % (out, x))):
            if r["Counter_Name"] == "FETCH_SIZE":
                k = re.sub(r"^void |\(anonymous namespace\)::|\(.*$", "", r["Kernel_Name"]) + " grid " + r["Grid_Size"]
                acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    except OSError as e:
        print("  no counters:", e)
    for k, (n, v) in sorted(acc.items()):
        if "wino4_" in k or "conv_q4_kernel" in k or "gemm_as" in k:
            print("  xcd=%d  %-70s launches %4d  FETCH_SIZE %8.0f KB/launch = %6.2f MB read (corrected x2)" % (x, k[:70], n, v / n, 2 * v / n / 1e3))
PY
cat $out/summary.txt
rm -f $out/*kernel_trace.csv $out/*counter_collection.csv $out/*agent_info.csv
