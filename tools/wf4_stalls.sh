#!/bin/bash
# Run ON THE GPU BOX: the cycle budget of the fused F(4x4,3x3) kernel's two shipped instantiations (ResNet-18 layer1:
# conv_wf4_kernel<..., 4, true>, 196 workgroups; layer2: <..., 3, false>, 112 workgroups; batch 32, bn + residual + relu tail).
# Four SQ counter passes (8 SQ counters each, --kernel-trace only: no other trace domain beside --pmc) over
# tools/wf4_bench.py, digested per instantiation into gpurun_out/wf4_stalls/digest.md (-> profiles/r06_wf4_stalls.md).
#   tools/wf4_stalls.sh [extra env assignments for the runs, e.g. PLANER_HIP_EXPERIMENT=...]
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/wf4_stalls
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $out/counters_list.txt 2>&1
i=0
for pm in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
          "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" \
          "SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
          "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SENDMSG"; do
  i=$((i+1))
  env "$@" rocprofv3 --kernel-trace --pmc $pm -d $out -o p$i --output-format csv -- python $R/tools/wf4_bench.py --shapes 64x56,128x28 --algos 9 > $out/p$i.log 2>&1
  tail -2 $out/p$i.log
done
# the same command untraced (the clock under the counter tool is lower: never compare the two arms)
env "$@" python $R/tools/wf4_bench.py --shapes 64x56,128x28 --algos 9 > $out/untraced.log 2>&1
python3 $R/tools/wf4_stalls_digest.py $out > $out/digest.md
cat $out/digest.md
