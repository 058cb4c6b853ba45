#!/bin/bash
# gpurun -- "bash tools/inflight_sweep.sh TAG": a few (device batch, slots per workgroup, sessions in flight) points of the headline workload
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06n}; OUT=gpurun_out/${T}_inflight_sweep.jsonl; : > $OUT
run() {  # device batch, spw, inflight, steps, extra env
  DB=$1; SPW=$2; F=$3; ST=$4; shift 4
  env "$@" timeout 500 python bench.py --steps $ST --warmup 12 --inflight $F --device-batch $DB --cross-attention-slots-per-workgroup $SPW --no-cpu-baseline --no-other-configs --no-roofline --no-serial-reference 2>/dev/null | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'device_batch': $DB, 'slots_per_workgroup': $SPW, 'inflight': $F, 'steps': $ST, 'env': '$*', 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'cross_attention': d['config']['cross_attention']}))" >> $OUT
  tail -1 $OUT
}
run 256 2 3 24 A=1
run 256 3 3 24 A=1
run 256 3 4 24 GPU_MAX_HW_QUEUES=8
run 192 3 3 24 A=1
run 192 3 4 24 GPU_MAX_HW_QUEUES=8
run 256 2 3 24 A=1
