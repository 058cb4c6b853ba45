#!/bin/bash
# decode timing of ONE cross-attention mode (WH_XABS=$2, default 1): ms per step + kernel table at 64 / 32 / 8 slots, then 64 x 3 in flight
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
OUT=$R/${1:-r04}_xabs_time.jsonl; : > $OUT; M=${2:-1}
WH_XABS=$M timeout 400 python tools/time_decode.py large-v3 ${3:-64,32,8} 1 2>&1 | grep '^{' | sed "s/^{/{\"xabs\": $M, /" >> $OUT
WH_XABS=$M timeout 400 python tools/time_decode.py large-v3 64 3 2>&1 | grep '^{' | sed "s/^{/{\"xabs\": $M, /" >> $OUT
cat $OUT
