#!/bin/bash
# SQ counters of the decode kernels at 64 slots with the absorbed cross-attention (two passes, the counter sets of tools/profile_gpu.sh)
TAG=${1:-r04f}; cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd /tmp
export WH_XABS=${2:-1}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 500 rocprofv3 --pmc $C --kernel-trace -d /tmp/${TAG}_sq$i -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_run.py large-v3 ${3:-64} 2 > $R/${TAG}_pmc_sq$i.log 2>&1; echo sq$i rc=$?
  DB=$(ls /tmp/${TAG}_sq$i/*.db /tmp/${TAG}_sq$i/*/*.db 2>/dev/null | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB > $R/${TAG}_pmc_sq$i.csv 2>> $R/${TAG}_pmc_sq$i.log
  grep -i "xabs\|cross\|name" $R/${TAG}_pmc_sq$i.csv | head
done
