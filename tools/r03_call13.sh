#!/bin/bash
# round 3, GPU call 13: persistent cross-attention with the publish / ticket / combine of an item deferred by one / two iterations
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
: > $R/r03n_xatt_persist_deferred_ab.jsonl
for cfg in "0 0 64" "2 0 64" "2 1 64" "2 0 32"; do set -- $cfg; WH_XATT_PERSIST=$1 WH_XATT_CLAIM=$2 timeout 300 python tools/fuse_ab.py large-v3 $3 2>&1 | tail -1 | tee -a $R/r03n_xatt_persist_deferred_ab.jsonl; done
: > $R/r03n_xatt_persist_deferred_inflight.jsonl
for cfg in "0 0 3" "2 0 3" "2 1 3" "2 0 2"; do set -- $cfg; echo "# persist=$1 claim=$2 inflight=$3" | tee -a $R/r03n_xatt_persist_deferred_inflight.jsonl
  WH_XATT_PERSIST=$1 WH_XATT_CLAIM=$2 timeout 300 python tools/time_decode.py large-v3 64 $3 2>/dev/null | grep -v "^#" | tee -a $R/r03n_xatt_persist_deferred_inflight.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d.pop('kernels_us', {}); print(json.dumps(d), 'xattn_us', k.get('dec_cross_attn'), 'fc1', k.get('dec_proj_fc1'))"; done
( WH_XATT_PERSIST=2 timeout 900 python -m pytest tests/test_gpu_fulldepth.py tests/test_gpu_dims.py -x -q -k "large or dims" 2>&1 | tail -3 ) | tee $R/r03n_pytest_persist2.log
