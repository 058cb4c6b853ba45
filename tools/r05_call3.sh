#!/bin/bash
# Round 5, GPU call 3: two batch tiles per projection workgroup (WH_D32_NB=2, the new default) against one (WH_D32_NB=1): parity / bit-identity tests, kernel table, bench
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; T=r05c
( timeout 900 python -m pytest tests/test_gpu_round5.py "tests/test_gpu_fulldepth.py" -q -x -k "not small and not tiny" 2>&1 | tail -30 ) > $R/${T}_pytest_nb2.log 2>&1; tail -n 3 $R/${T}_pytest_nb2.log
for NB in 1 2; do
  WH_D32_NB=$NB WH_XABS_SPLITS=2 timeout 200 python tools/time_decode.py large-v3 64 1 2>&1 | grep '^{' | sed "s/^{/{\"nb\": $NB, \"splits\": 2, /" >> $R/${T}_proj_nb_ab.jsonl
  WH_D32_NB=$NB WH_XABS_SPLITS=1 timeout 200 python tools/time_decode.py large-v3 128 1 2>&1 | grep '^{' | sed "s/^{/{\"nb\": $NB, \"splits\": 1, /" >> $R/${T}_proj_nb_ab.jsonl
done; cut -c1-420 $R/${T}_proj_nb_ab.jsonl
for NB in 1 2; do
  WH_D32_NB=$NB timeout 300 python tools/bench_sweep.py large-v3 64:3:128:-1 64:3:64:-1 2>> $R/${T}_bench.err | sed "s/^{/{\"nb\": $NB, /" >> $R/${T}_bench_nb_ab.jsonl
done; cat $R/${T}_bench_nb_ab.jsonl
GPU_MAX_HW_QUEUES=8 timeout 300 python tools/bench_sweep.py large-v3 64:4:128:-1 2>> $R/${T}_bench.err | sed "s/^{/{\"GPU_MAX_HW_QUEUES\": 8, /" >> $R/${T}_bench_knobs.jsonl
WH_XATT_GATE=1 timeout 300 python tools/bench_sweep.py large-v3 64:3:128:-1 2>> $R/${T}_bench.err | sed "s/^{/{\"WH_XATT_GATE\": 1, /" >> $R/${T}_bench_knobs.jsonl
cat $R/${T}_bench_knobs.jsonl
