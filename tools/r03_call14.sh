#!/bin/bash
# round 3, GPU call 14: 8-wave projection workgroups (a launch's whole weight + plane stream requested in ONE round trip): single-stream
# decode at 64 / 8 / 1 slots, three sessions in flight, parity tests with it on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
: > $R/r03o_proj_8wave_ab.jsonl
for cfg in "4 large-v3 64" "8 large-v3 64" "4 large-v3 8" "8 large-v3 8" "4 tiny.en 1" "8 tiny.en 1" "4 small 8" "8 small 8"; do set -- $cfg; WH_D32_NW=$1 timeout 300 python tools/fuse_ab.py $2 $3 2>&1 | tail -1 | tee -a $R/r03o_proj_8wave_ab.jsonl; done
: > $R/r03o_proj_8wave_inflight.jsonl
for cfg in "4 64 3" "8 64 3" "4 32 3" "8 32 3" "8 8 3" "4 8 3"; do set -- $cfg; echo "# nw=$1 slots=$2 inflight=$3" | tee -a $R/r03o_proj_8wave_inflight.jsonl
  WH_D32_NW=$1 timeout 300 python tools/time_decode.py large-v3 $2 $3 2>/dev/null | grep -v "^#" | tee -a $R/r03o_proj_8wave_inflight.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d.pop('kernels_us', {}); print(json.dumps(d), {x: k.get(x) for x in ('dec_cross_attn', 'dec_proj_qkv', 'dec_proj_oproj', 'dec_proj_fc1', 'dec_proj_fc2')})"; done
( WH_D32_NW=8 timeout 900 python -m pytest tests/test_gpu_dims.py tests/test_gpu_fulldepth.py tests/test_gpu_round2.py -x -q 2>&1 | tail -3 ) | tee $R/r03o_pytest_nw8.log
