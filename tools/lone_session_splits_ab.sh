#!/bin/bash
# gpurun -- "bash tools/lone_session_splits_ab.sh TAG": (1) BASELINE configs[4] with beam = 5 at 1 .. 4 key splits of the absorbed cross-attention (100 slots: 4 splits = 400 workgroups = two rounds of the chip);
# (2) ONE large-v3 session alone at 32 .. 256 slots, ms per decoder step by key splits (the library's automatic choice is 4 whatever the batch: a lone session = wh_transcribe* on a long audio)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06ah}
timeout 500 python tools/beam_ab.py "1:4" "1:3" "1:2" "1:1" "1:4" 2>/dev/null | grep '^{' > gpurun_out/${T}_beam5_key_splits.jsonl
cat gpurun_out/${T}_beam5_key_splits.jsonl
OUT=gpurun_out/${T}_lone_session_key_splits.jsonl; : > $OUT
for S in 4 3 2 1; do
  WH_XABS_SPLITS=$S timeout 400 python tools/time_decode.py large-v3 32,48,96,128,192,256 1 2>/dev/null | grep '^{' | sed "s/^{/{\"splits\": $S, /" | \
    python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d.pop('kernels_us'); d['cross_attn_us'] = k.get('dec_cross_attn'); d['xabs_vup_us'] = k.get('dec_xabs_vup'); print(json.dumps(d))" >> $OUT
done
cat $OUT
