#!/bin/bash
# SQ / TCP counters of the decode kernels at the headline device batch (128 slots, 1 key split): what xabs_attn waits for, and the latency /
# request counts of its L1 -> L2 read path (the per-CU stream cap of DESIGN 3.4).   gpurun -- 'bash tools/xabs_counters.sh r05k'
TAG=${1:-xc}; SLOTS=${SLOTS:-128}; export WH_XABS_SPLITS=${XS:-1}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "(TCP|TCC|SQ|TA)_[A-Za-z0-9_]+" | sort -u > $R/${TAG}_counter_names.txt; wc -l $R/${TAG}_counter_names.txt
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
         "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
         "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/${TAG}_c$i -o c -- python $GRAFT_REPO_ROOT/tools/pmc_run.py large-v3 $SLOTS 4 > $R/${TAG}_c$i.log 2>&1; echo "set $i rc=$?"
  DB=$(ls /tmp/${TAG}_c$i/*.db /tmp/${TAG}_c$i/*/*.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB 2>/dev/null | grep -E "Kernel,|xabs_attn|dec32_proj_kernel<2|xabs_vup|xabs_qk" > $R/${TAG}_counters_set$i.csv
  head -12 $R/${TAG}_counters_set$i.csv
done
