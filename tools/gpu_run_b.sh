#!/bin/bash
# round-2 dev run: MFMA decode path first contact + old-path diagnosis at large-v3 width
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "teacher_forced or batched_equals_single or hf_golden" 2>&1 | tail -40 ) > gpurun_out/r02b_micro_mfma.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_dims.py -q 2>&1 | tail -80 ) > gpurun_out/r02b_dims_mfma.log 2>&1
( WH_DEC_PATH=gemv timeout 900 python -m pytest tests/test_gpu_dims.py -q -k "large" 2>&1 | tail -80 ) > gpurun_out/r02b_dims_gemv_large.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_parity.py -q 2>&1 | tail -60 ) > gpurun_out/r02b_parity_mfma.log 2>&1
for f in r02b_micro_mfma r02b_dims_mfma r02b_dims_gemv_large r02b_parity_mfma; do echo "== $f"; tail -n 3 gpurun_out/$f.log; done
