#!/bin/bash
# gpurun -- "bash tools/stream_priority_sweep.sh TAG": the headline workload with the sessions' HIP streams on hardware queues of different priority
# (WH_STREAM_PRIORITIES, capi.hip create_session_stream: -1 high, 0 normal, 1 low, dealt to the sessions in creation order).
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06ag}; OUT=gpurun_out/${T}_stream_priority_sweep.jsonl; : > $OUT
run() {  # inflight, steps, label, extra env
  F=$1; ST=$2; L=$3; shift 3
  env "$@" timeout 400 python bench.py --steps $ST --warmup 12 --inflight $F --no-cpu-baseline --no-other-configs --no-roofline --no-serial-reference 2>/dev/null | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'priorities': '$L', 'inflight': $F, 'steps': $ST, 'env': '$*', 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $OUT
  tail -1 $OUT
}
run 3 24 "none (default)" A=1
run 3 24 "-1,0,1" WH_STREAM_PRIORITIES=-1,0,1
run 3 24 "-1,0,0" WH_STREAM_PRIORITIES=-1,0,0
run 3 24 "0,0,1" WH_STREAM_PRIORITIES=0,0,1
run 3 24 "-1,-1,1" WH_STREAM_PRIORITIES=-1,-1,1
run 4 24 "-1,0,0,1 (4 sessions, 8 hardware queues)" WH_STREAM_PRIORITIES=-1,0,0,1 GPU_MAX_HW_QUEUES=8
run 3 24 "none (default), again" A=1
