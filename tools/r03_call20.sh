#!/bin/bash
# round 3, GPU call 20: projection kernels with fewer registers (smaller weight chunks) x cross-attention residency caps, three sessions in flight
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
: > $R/r03s_proj_regs_x_xattn_residency.jsonl
for cfg in "5 0" "2 0" "2 24576" "4 32768" "2 32768" "3 32768"; do set -- $cfg; echo "# tc=$1 xatt_lds=$2 inflight=3" | tee -a $R/r03s_proj_regs_x_xattn_residency.jsonl
  WH_D32_TC=$1 WH_XATT_LDS=$2 timeout 300 python tools/time_decode.py large-v3 64 3 2>/dev/null | grep -v "^#" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d.pop('kernels_us', {}); d.update(xattn_us=k.get('dec_cross_attn'), qkv_us=k.get('dec_proj_qkv'), fc1_us=k.get('dec_proj_fc1')); print(json.dumps(d))" | tee -a $R/r03s_proj_regs_x_xattn_residency.jsonl; done
