#!/bin/bash
# round 3, GPU call 11: the persistent, software-pipelined cross-attention kernel (WH_XATT_PERSIST = workgroups per CU) against the
# one-item-per-workgroup kernel: same bits (MD5), single-stream and three-sessions-in-flight decode rates, parity tests with it on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
: > $R/r03l_xatt_persist_ab.jsonl
for k in 0 2 3; do WH_XATT_PERSIST=$k timeout 300 python tools/fuse_ab.py large-v3 64 2>&1 | tail -1 | tee -a $R/r03l_xatt_persist_ab.jsonl; done
for k in 0 2; do WH_XATT_PERSIST=$k timeout 300 python tools/fuse_ab.py large-v3 32 2>&1 | tail -1 | tee -a $R/r03l_xatt_persist_ab.jsonl; done
: > $R/r03l_xatt_persist_inflight.jsonl
for cfg in "0 3" "2 3" "3 3" "2 2" "0 2" "2 4"; do set -- $cfg; echo "# persist=$1 inflight=$2" | tee -a $R/r03l_xatt_persist_inflight.jsonl
  WH_XATT_PERSIST=$1 timeout 300 python tools/time_decode.py large-v3 64 $2 2>/dev/null | grep -v "^#" | tee -a $R/r03l_xatt_persist_inflight.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d.pop('kernels_us', {}); print(json.dumps(d), 'xattn_us', k.get('dec_cross_attn'))"; done
( WH_XATT_PERSIST=2 timeout 900 python -m pytest tests/test_gpu_fulldepth.py tests/test_gpu_dims.py -x -q -k "large or dims" 2>&1 | tail -5 ) | tee $R/r03l_pytest_persist2.log
( timeout 600 python -m pytest tests/test_gpu_dims.py -x -q 2>&1 | tail -3 ) | tee $R/r03l_pytest_default_dims.log
