#!/bin/bash
# round 3, GPU call 3: encoder attention v2 - parity tests that cover the encoder, then v1 / v2 timing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
( timeout 900 python -m pytest tests/test_gpu_dims.py tests/test_gpu_fulldepth.py tests/test_gpu_parity.py -k "encoder or fulldepth or dims" -q 2>&1 | tail -30 ) > $R/r03c_pytest_encoder.log 2>&1
tail -4 $R/r03c_pytest_encoder.log
: > $R/r03c_encoder_attention_v1_v2.jsonl
for v in 1 0; do
  WH_ENC_ATTN_V1=$v timeout 300 python tools/time_encoder.py large-v3 64 >> $R/r03c_encoder_attention_v1_v2.jsonl 2>> $R/r03c_enc.err
  WH_ENC_ATTN_V1=$v timeout 300 python tools/time_encoder.py large-v3 8 >> $R/r03c_encoder_attention_v1_v2.jsonl 2>> $R/r03c_enc.err
done
cat $R/r03c_encoder_attention_v1_v2.jsonl
