#!/bin/bash
# Decode-path tuning knobs re-swept in the headline regime (three 256-slot device batches in flight, 2 slots per cross-attention workgroup): one line per knob setting
#   gpurun -- 'bash tools/decode_knobs_ab.sh r06q'
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; OUT=$R/${1:-knobs}_decode_knobs_256x3.jsonl; : > $OUT
for K in "X=0" "WH_D32_TC=3" "WH_D32_TC=2" "WH_D32_KS_FC2=2" "WH_D32_KS_FC2=8" "WH_D32_KS_Q=2" "WH_D32_KS_RESID=2" "WH_D32_NTW=1" "WH_XABS_NT=0" "X=1"; do
  env $K timeout 400 python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-other-configs --no-roofline --no-serial-reference 2>> $R/${1:-knobs}_knobs.err | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'knob': '$K', 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $OUT
  tail -1 $OUT
done
