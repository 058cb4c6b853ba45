#!/bin/bash
# gpurun -- "bash tools/gate_ab.sh TAG": the cross-attention gate (dec_shared.h) on / off in the headline regime (three 256-slot device batches in flight); it was tuned at 64-slot batches (round 3)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; OUT=gpurun_out/${1:-r06al}_xattn_gate_ab_256x3.jsonl; : > $OUT
for K in "X=0" "WH_XATT_GATE=0" "X=1" "WH_XATT_GATE=0" "X=2"; do
  env $K timeout 400 python bench.py --steps 24 --warmup 12 --no-cpu-baseline --no-other-configs --no-roofline --no-serial-reference 2>/dev/null | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'knob': '$K', 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $OUT
  tail -1 $OUT
done
