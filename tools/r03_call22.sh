#!/bin/bash
# round 3, GPU call 22: beam-search cross-attention with one wave per beam (WH_XATT_BEAM_WAVES=1): beam tests vs the oracle, configs[4]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
export WH_XATT_BEAM_WAVES=1
( timeout 200 python -m pytest tests/test_gpu_beam.py -x -q 2>&1 | tail -4 ) | tee $R/r03t_pytest_beam_waves.log
timeout 100 python tools/time_beam.py 2>/dev/null | tail -1 | tee $R/r03t_beam_waves_configs4.jsonl
