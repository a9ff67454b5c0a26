#!/bin/bash
# usage: tools/pmc_conv.sh <outdir-tag> N Cin H Cout k stride pad cfgname split
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; tag=$1; shift
i=0
for pm in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
          "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" \
          "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pm -d $R/gpurun_out/pmc_$tag -o pass$i --output-format csv -- python $R/tools/prof_one.py "$@" 3 > /dev/null 2>&1
done
python3 - <<PY
import csv, collections, glob
for f in sorted(glob.glob('$R/gpurun_out/pmc_$tag/*counter_collection.csv')):
    rows=list(csv.DictReader(open(f)))
    agg=collections.defaultdict(float)
    ks=[r for r in rows if 'conv_' in r['Kernel_Name'] and 'permute' not in r['Kernel_Name']]
    disp=len({r['Dispatch_Id'] for r in ks})
    for r in ks: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    if ks: print('  VGPR', ks[0]['VGPR_Count'], 'AGPR', ks[0]['Accum_VGPR_Count'], 'LDS', ks[0]['LDS_Block_Size'], 'grid', ks[0]['Grid_Size'])
    for c,v in agg.items(): print('   %-28s %14.0f' % (c, v/max(disp,1)))
PY
