#!/bin/bash
# gpurun -- "bash tools/enc_confine_ab.sh [TAG]": does confining the encoder's GEMMs to a share of the CUs (persistent tile loop with a capped grid,
# WH_GEMM_PERSIST=1 WH_GEMM_PERSIST_WGS=n) help the mix of one session encoding while the others decode?  Headline workload, short runs.
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06g}; OUT=gpurun_out/${T}_encoder_confinement_ab.jsonl; : > $OUT
run() {  # label, inflight, env...
  L=$1; F=$2; shift 2
  env "$@" timeout 400 python bench.py --steps 12 --warmup 3 --inflight $F --no-cpu-baseline --no-other-configs --no-roofline --no-serial-reference 2>/dev/null | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'config': '$L', 'inflight': $F, 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step'], 'encoder_ms_per_chunk': d['encoder_ms_per_chunk']}))" >> $OUT
  tail -1 $OUT
}
run "one workgroup per tile (default)" 3 WH_GEMM_PERSIST=0
run "persistent, 256 workgroups" 3 WH_GEMM_PERSIST=1
for n in 192 128 96 64; do run "persistent, $n workgroups" 3 WH_GEMM_PERSIST=1 WH_GEMM_PERSIST_WGS=$n; done
for n in 128 96 64; do run "persistent, $n workgroups" 4 WH_GEMM_PERSIST=1 WH_GEMM_PERSIST_WGS=$n GPU_MAX_HW_QUEUES=8; done
run "one workgroup per tile (default)" 4 WH_GEMM_PERSIST=0 GPU_MAX_HW_QUEUES=8
