#!/bin/bash
# round 3, GPU call 4: whole GPU suite (hidden hi|lo planes, encoder attention v2, shared cross K/V of beams), the 256x128 GEMM A/B,
# decoder step time, then the default bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > $R/r03d_pytest_gpu.log 2>&1
tail -4 $R/r03d_pytest_gpu.log
( WH_GEMM2=1 timeout 600 python -m pytest tests/test_gpu_dims.py tests/test_gpu_parity.py -k "encoder" -q 2>&1 | tail -30 ) > $R/r03d_pytest_gemm2.log 2>&1
tail -3 $R/r03d_pytest_gemm2.log
: > $R/r03d_encoder_gemm2_ab.jsonl
for v in 0 1; do
  WH_GEMM2=$v timeout 300 python tools/time_encoder.py large-v3 64,8 >> $R/r03d_encoder_gemm2_ab.jsonl 2>> $R/r03d_enc.err
done
cat $R/r03d_encoder_gemm2_ab.jsonl
: > $R/r03d_decode_step.jsonl
for cfg in "large-v3 64" "large-v3 8" "tiny.en 1" "small 8"; do timeout 300 python tools/fuse_ab.py $cfg >> $R/r03d_decode_step.jsonl 2>> $R/r03d_dec.err; done
cat $R/r03d_decode_step.jsonl
( timeout 900 python bench.py > $R/r03d_bench.json ) 2> $R/r03d_bench.err; tail -3 $R/r03d_bench.err; cut -c1-400 $R/r03d_bench.json
