#!/bin/bash
# round 3, GPU call 5: cross-attention residency cap (extra LDS per workgroup) so that a projection workgroup of another session fits on the CU
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
B="--steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline"
: > $R/r03e_xatt_residency_sweep.jsonl
run() { tag=$1; shift; ( env "$@" timeout 300 python bench.py $B > $R/r03e_bench_$tag.json ) 2> $R/r03e_bench_$tag.err; python - <<PY
import json
try:
    d=json.load(open("$R/r03e_bench_$tag.json")); r=dict(variant="$tag", value=d["value"], ms_per_step=d["ms_per_step"], value_single_stream=d["value_single_stream"], us_per_decoder_step=d["stages"]["us_per_decoder_step"])
except Exception as e: r=dict(variant="$tag", error=str(e))
open("$R/r03e_xatt_residency_sweep.jsonl","a").write(json.dumps(r)+"\n"); print(r)
PY
}
run p6_lds0 WH_XATT_LDS=0
run p6_lds16k WH_XATT_LDS=16384
run p6_lds24k WH_XATT_LDS=24576
run p6_lds32k WH_XATT_LDS=32768
run p6_lds45k WH_XATT_LDS=45000
run p4_lds0 WH_XATT_PASSES=4
run p4_lds24k WH_XATT_PASSES=4 WH_XATT_LDS=24576
run p8_lds32k WH_XATT_PASSES=8 WH_XATT_LDS=32768
B="--steps 8 --warmup 4 --inflight 4 --no-cpu-baseline --no-other-configs --no-roofline"
run p6_lds32k_inflight4 WH_XATT_LDS=32768
run p6_lds0_inflight4 WH_XATT_LDS=0
