#!/bin/bash
# Decode-path tuning knobs re-swept in the headline regime (three 128-slot device batches in flight): one bench_sweep line per knob setting
#   gpurun -- 'bash tools/decode_knobs_ab.sh r05n'
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; OUT=$R/${1:-knobs}_decode_knobs_128x3.jsonl
for K in "X=0" "WH_D32_KS_FC2=2" "WH_D32_KS_FC2=8" "WH_D32_TC=3" "WH_D32_TC=2" "WH_D32_KS_Q=2" "WH_D32_KS_RESID=2" "X=1"; do
  env $K timeout 200 python tools/bench_sweep.py large-v3 64:3:128:-1 2>> $R/${1:-knobs}_knobs.err | sed "s/^{/{\"knob\": \"$K\", /" >> $OUT
done
cut -c1-170 $OUT
