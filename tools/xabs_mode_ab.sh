#!/bin/bash
# A/B of the cross-attention modes on one box: per-layer K / V stream (WH_XABS=0) vs absorbed (WH_XABS=1); decode ms per step + kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
OUT=$R/${1:-r04c}_xabs_ab.jsonl; : > $OUT
for mode in 0 1; do
  WH_XABS=$mode timeout 400 python tools/time_decode.py large-v3 ${2:-64,32,8} 1 2>&1 | grep '^{' | sed "s/^{/{\"xabs\": $mode, /" >> $OUT
done
for mode in 0 1; do
  WH_XABS=$mode timeout 400 python tools/time_decode.py large-v3 64 3 2>&1 | grep '^{' | sed "s/^{/{\"xabs\": $mode, /" >> $OUT
done
cat $OUT
