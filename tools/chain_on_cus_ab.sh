#!/bin/bash
# gpurun -- "bash tools/chain_on_cus_ab.sh TAG": the per-kernel table of one 256-slot session alone (bench.py roofline leg: HIP events around every launch),
# on the whole chip and on a 128- / 64-CU partition (WH_CU_PARTS), with one and with two weight-row tiles per projection workgroup (WH_D32_RT_BT).
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06t}; OUT=gpurun_out/${T}_chain_on_cus_ab.jsonl; : > $OUT
run() {
  env "$@" timeout 500 python bench.py --steps 4 --warmup 4 --inflight 1 --device-batch 256 --cross-attention-splits 1 --cross-attention-slots-per-workgroup 2 --no-cpu-baseline --no-other-configs --no-serial-reference 2>gpurun_out/${T}_last.err | tail -1 | \
    python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['roofline']['kernels']
print(json.dumps({'env': '$*', 'audio_s_per_s': round(d['value'], 1), 'us_per_decoder_step': round(d['stages']['us_per_decoder_step'], 1), 'encoder_ms_per_chunk': round(d['stages']['encoder_ms_per_chunk'], 3),
                  'avg_us': {n: v['avg_us'] for n, v in k.items() if n.startswith('dec_') or n == 'sampler'}}))" >> $OUT
  tail -1 $OUT
}
if [ -n "$AB_POINTS" ]; then
  while read -r line; do [ -n "$line" ] && run $line; done <<< "$AB_POINTS"
  exit 0
fi
run A=1
run WH_D32_RT_BT=5
run WH_CU_PARTS=2
run WH_CU_PARTS=2 WH_D32_RT_BT=5
run WH_CU_PARTS=4
run WH_CU_PARTS=4 WH_D32_RT_BT=5
