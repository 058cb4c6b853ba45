#!/bin/bash
# round 3, GPU call 2: whole GPU suite with the fused decoder launches, then fused / unfused A-B (bits + time)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $R/r03b_pytest_gpu.log 2>&1
tail -4 $R/r03b_pytest_gpu.log
: > $R/r03b_fuse_ab.jsonl
for cfg in "large-v3 64" "large-v3 8" "tiny.en 1" "small 8"; do
  for f in 0 1 2 3; do
    WH_DEC_FUSE=$f timeout 300 python tools/fuse_ab.py $cfg >> $R/r03b_fuse_ab.jsonl 2>> $R/r03b_fuse_ab.err
  done
done
WH_DEC_FUSE=0 WH_XATT_PASSES=8 timeout 300 python tools/fuse_ab.py large-v3 64 >> $R/r03b_fuse_ab.jsonl 2>> $R/r03b_fuse_ab.err
cat $R/r03b_fuse_ab.jsonl
B="--steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline"
for f in 0 3; do
  ( WH_DEC_FUSE=$f timeout 300 python bench.py $B > $R/r03b_bench_fuse$f.json ) 2> $R/r03b_bench_fuse$f.err
  python -c "
import json; d=json.load(open('$R/r03b_bench_fuse$f.json')); print('fuse$f', d['value'], d['ms_per_step'], d['value_single_stream'], d['stages']['us_per_decoder_step'])"
done
