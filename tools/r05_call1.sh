#!/bin/bash
# Round 5, GPU call 1: the whole GPU suite on the fp32-row K / V path, configuration sweeps of the headline, cross-attention mode crossover
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; T=r05a
( timeout 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -40 ) > $R/${T}_pytest_gpu.log 2>&1; tail -n 3 $R/${T}_pytest_gpu.log
timeout 420 python tools/bench_sweep.py large-v3 64:3:64:-1 64:3:128:-1 64:2:128:-1 64:2:128:2 32:2:32:4 32:3:32:4 > $R/${T}_bench_sweep.jsonl 2> $R/${T}_bench_sweep.err; cat $R/${T}_bench_sweep.jsonl
for KA in 0 1; do
  HIP_FORCE_DEV_KERNARG=$KA timeout 200 python tools/bench_sweep.py large-v3 64:3:128:-1 2>> $R/${T}_kernarg.err | sed "s/^{/{\"HIP_FORCE_DEV_KERNARG\": $KA, /" >> $R/${T}_kernarg_ab.jsonl
done; cat $R/${T}_kernarg_ab.jsonl
for M in 0 1; do
  WH_XABS=$M timeout 300 python tools/time_decode.py large-v3 8,16,24,32,48 1 2>&1 | grep '^{' | sed "s/^{/{\"xabs\": $M, /" >> $R/${T}_mode_crossover.jsonl
done
timeout 120 python tools/time_decode.py small 8 1 2>&1 | grep '^{' >> $R/${T}_mode_crossover.jsonl
timeout 120 python tools/time_decode.py tiny.en 1 1 2>&1 | grep '^{' >> $R/${T}_mode_crossover.jsonl
cut -c1-400 $R/${T}_mode_crossover.jsonl
