#!/bin/bash
# weight-load cache policy of the decoder projections at two batch tiles (64 slots): WH_D32_NTW=1 non-temporal always (before), unset = plain loads
# when a slab has more than one reader; headline in flight + single-stream kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
OUT=$R/${1:-r04am}_d32_weight_nt_ab.jsonl; : > $OUT
B="timeout 100 python bench.py --warmup 2 --steps 12 --no-cpu-baseline --no-roofline --no-other-configs --no-serial-reference"
for K in "WH_D32_NTW=1" "X=0"; do
  env $K $B 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'knob': '$K', 'value': j['value'], 'ms_per_step': j['ms_per_step']}))" >> $OUT
  env $K WH_XABS_SPLITS=2 timeout 60 python tools/time_decode.py large-v3 64 1 2>&1 | grep '^{' | sed "s/^{/{\"knob\": \"$K\", /" >> $OUT
done
cut -c1-700 $OUT
