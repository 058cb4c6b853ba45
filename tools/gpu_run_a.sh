#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_gpu_dims.py -x -q 2>&1 | tail -40 ) > gpurun_out/r02a_dims_oldpath.log 2>&1
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "multi_window or chunked_vad or temperature_sampling or greedy_vs_oracle or batch_above" 2>&1 | tail -30 ) > gpurun_out/r02a_parity_subset.log 2>&1
tail -5 gpurun_out/r02a_dims_oldpath.log gpurun_out/r02a_parity_subset.log
