#!/bin/bash
# tuning sweep of the MFMA decode path (each configuration is its own process: the knobs are read once)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02d_sweep.log
: > $O
run() { echo "## $*" >> $O; ( env "$@" ) 2>&1 | grep -v "^#" >> $O; }
T="timeout 300 python tools/time_decode.py"
run $T large-v3 8,32 1
run WH_D32_KS_RESID=2 WH_D32_KS_Q=2 $T large-v3 8,32 1
run WH_D32_KS_RESID=4 WH_D32_KS_Q=4 $T large-v3 8,32 1
run WH_D32_KS_FC2=2 $T large-v3 8,32 1
run WH_D32_KS_FC2=8 $T large-v3 8,32 1
run WH_D32_KS_WIDE=2 $T large-v3 8,32 1
run WH_XATT_NOFENCE=1 $T large-v3 8,32 1
run WH_XATT_NOFENCE=1 WH_XATT_PASSES=16 $T large-v3 8,32 1
run WH_XATT_NOFENCE=1 WH_XATT_PASSES=8 $T large-v3 8,32 1
run WH_XATT_NOFENCE=1 $T large-v3 32 2
run WH_XATT_NOFENCE=1 $T large-v3 32 3
run WH_XATT_NOFENCE=1 $T large-v3 16 4
run WH_XATT_NOFENCE=1 $T large-v3 8 3
run WH_XATT_NOFENCE=1 $T tiny.en 1,8 1
run WH_XATT_NOFENCE=1 $T small 8 1
( timeout 600 python -m pytest tests/test_gpu_dims.py -q -x 2>&1 | tail -5 ) >> $O 2>&1
( WH_XATT_NOFENCE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "reproducible or batch_above or greedy_vs_oracle" 2>&1 | tail -5 ) >> $O 2>&1
cut -c1-400 $O
