#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 900 python tools/time_decode.py large-v3 8,32 1 ) > gpurun_out/r02c_time_mfma.log 2>&1
( WH_DEC_PATH=gemv timeout 900 python tools/time_decode.py large-v3 8,32 1 ) > gpurun_out/r02c_time_gemv.log 2>&1
( timeout 300 python tools/time_decode.py tiny.en 1,8 1 ) > gpurun_out/r02c_time_tiny_mfma.log 2>&1
( WH_DEC_PATH=gemv timeout 300 python tools/time_decode.py tiny.en 1,8 1 ) > gpurun_out/r02c_time_tiny_gemv.log 2>&1
( timeout 600 python tools/time_decode.py large-v3 32 2 ) > gpurun_out/r02c_time_mfma_if2.log 2>&1
cat gpurun_out/r02c_time_*.log | cut -c1-1500
