#!/bin/bash
# re-tune of the existing decode knobs in the new in-flight regime (2 key splits per slot: a cross-attention launch takes half of the CUs):
# headline workload, 3 sessions in flight, 12 steps per case
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
OUT=$R/${1:-r04ai}_knobs_ab.jsonl; : > $OUT
B="timeout 300 python bench.py --warmup 2 --steps 12 --no-cpu-baseline --no-roofline --no-other-configs --no-serial-reference"
for K in "X=0" "WH_D32_TC=3" "WH_D32_TC=2" "WH_D32_KS_FC2=8" "WH_D32_KS_FC2=2" "WH_D32_KS_Q=2" "WH_D32_KS_RESID=2" "WH_XABS_NT=0"; do
  env $K $B 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'knob': '$K', 'value': j['value'], 'ms_per_step': j['ms_per_step'], 'median_ms_per_step': j.get('median_ms_per_step')}))" >> $OUT
done
cat $OUT
