#!/bin/bash
# round 3, GPU call 1: the new parity tests + a sweep of two co-residency knobs on the headline workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
( timeout 900 python -m pytest tests/test_gpu_fulldepth.py tests/test_gpu_round3.py "tests/test_gpu_round2.py::test_two_ranks_on_one_gpu_rehearsal" -x -q 2>&1 | tail -40 ) > $R/r03a_pytest_new.log 2>&1
tail -5 $R/r03a_pytest_new.log
B="--steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline"
run() { tag=$1; shift; ( env "$@" timeout 300 python bench.py $B > $R/r03a_bench_$tag.json ) 2> $R/r03a_bench_$tag.err; python - <<PY
import json
try:
    d=json.load(open("$R/r03a_bench_$tag.json")); print("$tag", d["value"], d["ms_per_step"], d["value_single_stream"], d["encoder_ms_per_chunk"], d["stages"]["us_per_decoder_step"])
except Exception as e: print("$tag failed", e)
PY
}
run base WH_X=0
run encstream WH_ENC_STREAM=1
run xatt6 WH_XATT_PASSES=6
run both WH_ENC_STREAM=1 WH_XATT_PASSES=6
