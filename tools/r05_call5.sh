#!/bin/bash
# Round 5, GPU call 5: 24-bit cross K / V rows (Float16 + 8-bit residual) in place of fp32 rows: parity in K / V-row mode, mode crossover
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; T=r05h
( timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_realistic.py tests/test_gpu_dims.py -q -x -k "not large-v3-absorbed and not small-absorbed" 2>&1 | tail -30 ) > $R/${T}_pytest_rows24.log 2>&1; tail -n 3 $R/${T}_pytest_rows24.log
WH_XABS=0 timeout 300 python tools/time_decode.py large-v3 8,16,20,24,28,32 1 2>&1 | grep '^{' | sed "s/^{/{\"xabs\": 0, /" >> $R/${T}_mode_crossover_rows24.jsonl
timeout 120 python tools/time_decode.py small 8 1 2>&1 | grep '^{' >> $R/${T}_mode_crossover_rows24.jsonl
timeout 120 python tools/time_decode.py tiny.en 1 1 2>&1 | grep '^{' >> $R/${T}_mode_crossover_rows24.jsonl
cut -c1-330 $R/${T}_mode_crossover_rows24.jsonl
