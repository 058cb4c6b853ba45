#!/bin/bash
# gpurun -- "bash tools/cu_partition_sweep.sh TAG": sessions in flight on DISJOINT compute-unit partitions (WH_CU_PARTS, csrc/capi.hip create_session_stream:
# a session's HIP stream carries a CU mask, hipExtStreamCreateWithCUMask) against the free-for-all of the default configuration, on the headline workload.
# First the probe that says which CUs a mask's bits select (tools/cu_mask_probe.hip).
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06s}; OUT=gpurun_out/${T}_cu_partition_sweep.jsonl; : > $OUT
timeout 120 tools/build/cu_mask_probe > gpurun_out/${T}_cu_mask_probe.jsonl 2>&1; cat gpurun_out/${T}_cu_mask_probe.jsonl
run() {  # device batch, spw, inflight, steps, extra env
  DB=$1; SPW=$2; F=$3; ST=$4; shift 4
  env "$@" timeout 500 python bench.py --steps $ST --warmup 12 --inflight $F --device-batch $DB --cross-attention-slots-per-workgroup $SPW --no-cpu-baseline --no-other-configs --no-roofline --no-serial-reference 2>gpurun_out/${T}_last.err | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'device_batch': $DB, 'slots_per_workgroup': $SPW, 'inflight': $F, 'steps': $ST, 'env': '$*', 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'cross_attention': d['config']['cross_attention']}))" >> $OUT
  tail -1 $OUT
}
if [ -n "$SWEEP_POINTS" ]; then
  while read -r line; do [ -n "$line" ] && run $line; done <<< "$SWEEP_POINTS"
  exit 0
fi
run 256 2 3 24 A=1
run 256 2 2 24 WH_CU_PARTS=2
run 256 4 4 32 WH_CU_PARTS=4
run 256 3 3 24 WH_CU_PARTS=3 WH_CU_PART_EXTRA=1
run 256 4 4 32 WH_CU_PARTS=2
run 256 2 3 24 A=1
