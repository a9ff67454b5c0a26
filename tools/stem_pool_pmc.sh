#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 counters of conv_stem_pool_kernel (tools/stem_pool_probe.py --only fused)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/sppmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for pm in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
          "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
          "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pm -d $out -o p$i --output-format csv -- python $R/tools/stem_pool_probe.py --reps 5 --only fused > $out/p$i.log 2>&1
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob("$out/p*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "stem_pool" in r["Kernel_Name"]:
            agg[r["Dispatch_Id"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg: continue
    last = sorted(agg, key=int)[-1]
    print(f.split("/")[-1], "dispatch", last)
    for k, v in sorted(agg[last].items()):
        print("   %-28s %.4g" % (k, sum(v)))
PY
