#!/bin/bash
# gpurun -- "bash tools/xabs_spw_ab.sh [TAG]": slots per xabs_attn workgroup (WH_XABS_SPW) x device batch x sessions in flight, headline workload.
# workgroups per cross-attention launch = ceil(slots / spw) x splits (bench.py picks splits = 128 / slots, at least 1)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06i}; OUT=gpurun_out/${T}_xabs_slots_per_workgroup_ab.jsonl; : > $OUT
( WH_XABS_SPW=2 timeout 600 python -m pytest tests/test_gpu_round5.py -q -k "lone" 2>&1 | tail -3 ) > gpurun_out/${T}_pytest_spw2.log; tail -2 gpurun_out/${T}_pytest_spw2.log
run() {  # device batch, spw, inflight, extra env
  DB=$1; SPW=$2; F=$3; shift 3
  env WH_XABS_SPW=$SPW "$@" timeout 500 python bench.py --steps 24 --warmup 12 --inflight $F --device-batch $DB --no-cpu-baseline --no-other-configs --no-roofline --no-serial-reference 2>/dev/null | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'device_batch': $DB, 'slots_per_workgroup': $SPW, 'inflight': $F, 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'cross_attention': d['config']['cross_attention']}))" >> $OUT
  tail -1 $OUT
}
run 128 1 3
run 256 2 3
run 256 2 2
run 256 1 3
run 128 2 3
run 128 2 4 GPU_MAX_HW_QUEUES=8
run 128 2 5 GPU_MAX_HW_QUEUES=8
run 256 4 3
run 256 4 4 GPU_MAX_HW_QUEUES=8
run 192 2 3
run 256 2 4 GPU_MAX_HW_QUEUES=8
