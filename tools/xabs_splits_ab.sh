#!/bin/bash
# key splits per slot of the absorbed cross-attention (bench.py --cross-attention-splits = workgroups per slot = CUs taken) x sessions in flight:
# headline (64 chunks per step) for the pairs in $2.., then the single-stream kernel table for the split counts seen
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
TAG=${1:-r04ae}; shift
OUT=$R/${TAG}_splits_ab.jsonl; : > $OUT
B="timeout 300 python bench.py --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --no-serial-reference"
for cfg in "$@"; do
  set -- $cfg
  $B --cross-attention-splits $1 --inflight $2 --steps $3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'splits': $1, 'inflight': $2, 'steps': $3, 'value': j['value'], 'ms_per_step': j['ms_per_step'], 'median_ms_per_step': j.get('median_ms_per_step')}))" >> $OUT
done
for S in ${SINGLE:-}; do
  WH_XABS_SPLITS=$S timeout 300 python tools/time_decode.py large-v3 64 1 2>&1 | grep '^{' | sed "s/^{/{\"splits\": $S, /" >> $OUT
done
cat $OUT | cut -c1-700
