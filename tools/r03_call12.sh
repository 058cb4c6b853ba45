#!/bin/bash
# round 3, GPU call 12: persistent cross-attention with static-stride / per-XCD-sharded claims (call 11: ONE claim counter = 153 us per launch,
# the same at 2 and 3 workgroups per CU: ~88 claims per us on one word)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
: > $R/r03m_xatt_persist_claims_ab.jsonl
for cfg in "0 0" "2 0" "2 1" "3 0" "3 1" "4 0"; do set -- $cfg; WH_XATT_PERSIST=$1 WH_XATT_CLAIM=$2 timeout 300 python tools/fuse_ab.py large-v3 64 2>&1 | tail -1 | tee -a $R/r03m_xatt_persist_claims_ab.jsonl; done
: > $R/r03m_xatt_persist_inflight.jsonl
for cfg in "0 0 3" "2 0 3" "2 1 3" "3 0 3" "2 0 2"; do set -- $cfg; echo "# persist=$1 claim=$2 inflight=$3" | tee -a $R/r03m_xatt_persist_inflight.jsonl
  WH_XATT_PERSIST=$1 WH_XATT_CLAIM=$2 timeout 300 python tools/time_decode.py large-v3 64 $3 2>/dev/null | grep -v "^#" | tee -a $R/r03m_xatt_persist_inflight.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d.pop('kernels_us', {}); print(json.dumps(d), 'xattn_us', k.get('dec_cross_attn'), 'fc1', k.get('dec_proj_fc1'))"; done
