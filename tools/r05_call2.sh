#!/bin/bash
# Round 5, GPU call 2: L2-prefetch A/B of xabs_attn (kernel time, bench), the GPU tests call 1 did not reach, beam = 5 in both modes, MALL-sized session sets
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; T=r05b
for PF in 0 2 3; do
  WH_XABS_PF=$PF WH_XABS_SPLITS=2 timeout 200 python tools/time_decode.py large-v3 64 1 2>&1 | grep '^{' | sed "s/^{/{\"pf\": $PF, \"splits\": 2, /" >> $R/${T}_xabs_prefetch_ab.jsonl
done
WH_XABS_PF=2 WH_XABS_SPLITS=1 timeout 200 python tools/time_decode.py large-v3 128 1 2>&1 | grep '^{' | sed "s/^{/{\"pf\": 2, \"splits\": 1, /" >> $R/${T}_xabs_prefetch_ab.jsonl
WH_XABS_PF=0 WH_XABS_SPLITS=1 timeout 200 python tools/time_decode.py large-v3 128 1 2>&1 | grep '^{' | sed "s/^{/{\"pf\": 0, \"splits\": 1, /" >> $R/${T}_xabs_prefetch_ab.jsonl
cut -c1-330 $R/${T}_xabs_prefetch_ab.jsonl
for PF in 0 2 3; do
  WH_XABS_PF=$PF timeout 300 python tools/bench_sweep.py large-v3 64:3:128:-1 64:3:64:-1 2>> $R/${T}_bench_pf.err | sed "s/^{/{\"pf\": $PF, /" >> $R/${T}_bench_prefetch_ab.jsonl
done; cat $R/${T}_bench_prefetch_ab.jsonl
( WH_XABS_PF=2 timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_dims.py -q -x -k "splits or absorbed or xabs or invarian" 2>&1 | tail -5 ) > $R/${T}_pytest_pf2.log 2>&1; tail -n 2 $R/${T}_pytest_pf2.log
( timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py tests/test_gpu_round5.py -q 2>&1 | tail -30 ) > $R/${T}_pytest_rest.log 2>&1; tail -n 3 $R/${T}_pytest_rest.log
timeout 400 python tools/beam_ab.py 0:0 1:4 1:2 > $R/${T}_beam_ab.jsonl 2> $R/${T}_beam_ab.err; cat $R/${T}_beam_ab.jsonl
WH_XABS=1 timeout 300 python tools/bench_sweep.py large-v3 32:2:32:4 32:3:32:4 64:1:64:4 > $R/${T}_mall_sweep.jsonl 2> $R/${T}_mall_sweep.err; cat $R/${T}_mall_sweep.jsonl
