#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
timeout 300 python -m pytest tests/test_gpu_round4.py -q -k key_split > $R/r04af_pytest_key_splits.log 2>&1; tail -3 $R/r04af_pytest_key_splits.log
OUT=$R/r04af_more_ab.jsonl; : > $OUT
B="timeout 300 python bench.py --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --no-serial-reference"
run() { # label, env, args
  env $2 $B $3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'case': '$1', 'value': j['value'], 'ms_per_step': j['ms_per_step'], 'median_ms_per_step': j.get('median_ms_per_step'), 'cross_attention': j['config'].get('cross_attention')}))" >> $OUT
}
run "128 slots x3, splits 1" "X=1" "--batch 128 --inflight 3 --steps 6 --cross-attention-splits 1"
run "128 slots x3, splits 2" "X=1" "--batch 128 --inflight 3 --steps 6 --cross-attention-splits 2"
run "64 x4, splits 2, 8 hw queues" "GPU_MAX_HW_QUEUES=8" "--inflight 4 --steps 12"
run "64 x6, splits 2, 8 hw queues" "GPU_MAX_HW_QUEUES=8" "--inflight 6 --steps 12"
run "64 x3, splits 2, 8 hw queues" "GPU_MAX_HW_QUEUES=8" "--inflight 3 --steps 12"
cat $OUT
SKIP_SQ=1 XS=2 bash tools/profile_gpu.sh r04af > $R/r04af_profile.log 2>&1; tail -25 $R/r04af_profile.log
