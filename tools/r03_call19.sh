#!/bin/bash
# round 3, GPU call 19: log-mel MFMA loop with the basis loads of 2 / 5 / 10 k-steps unrolled (A/B)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
: > $R/r03r_mel_unroll_ab.jsonl
for u in 2 5 10; do WH_MEL_UNROLL=$u timeout 300 python tools/time_encoder.py large-v3 64 2>/dev/null | grep -v "^#" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels_us']; print(json.dumps({'unroll': $u, 'B': d['B'], 'mel_power_us': k.get('mel_power'), 'mel_finalize_us': k.get('mel_finalize'), 'md5': d['encoder_output_md5']}))" | tee -a $R/r03r_mel_unroll_ab.jsonl; done
