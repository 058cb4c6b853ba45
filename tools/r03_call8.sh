#!/bin/bash
# round 3, GPU call 8: (1) beam-search cross-attention with one K/V fetch per audio (tests + configs[4] A/B); (2) the cross-attention gate:
# sessions in flight take turns at the HBM-bound kernel (decode-only probe + bench A/B, outputs must not change)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
timeout 600 python -m pytest tests/test_gpu_beam.py -x -q 2>&1 | tail -4 | tee $R/r03h_pytest_beam.log
( WH_XATT_GATE=1 timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_dims.py -x -q 2>&1 | tail -4 ) | tee $R/r03h_pytest_gate_forced_on.log
: > $R/r03h_beam_configs4_ab.jsonl
for f in 1 0; do WH_XATT_BEAM_SHARED=$f timeout 400 python tools/time_beam.py 2>/dev/null | tail -1 | tee -a $R/r03h_beam_configs4_ab.jsonl; done
: > $R/r03h_gate_decode_probe.jsonl
for g in 0 1; do for f in 3 4; do echo "# gate=$g inflight=$f" | tee -a $R/r03h_gate_decode_probe.jsonl
  WH_XATT_GATE=$g timeout 300 python tools/time_decode.py large-v3 64 $f 2>/dev/null | grep -v "^#" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); d.pop('kernels_us', None); print(json.dumps(d))" | tee -a $R/r03h_gate_decode_probe.jsonl; done; done
B="--steps 6 --warmup 3 --no-serial-reference --no-cpu-baseline --no-other-configs --no-roofline"
: > $R/r03h_gate_bench_ab.jsonl
run() { tag=$1; shift; fl=$1; shift; ( env "$@" timeout 300 python bench.py $B $fl > $R/r03h_bench_$tag.json ) 2> $R/r03h_bench_$tag.err; python - <<PY
import json
try:
    d=json.load(open("$R/r03h_bench_$tag.json")); r=dict(variant="$tag", value=d["value"], ms_per_step=d["ms_per_step"], median_ms_per_step=d["median_ms_per_step"])
except Exception as e: r=dict(variant="$tag", error=str(e))
open("$R/r03h_gate_bench_ab.jsonl","a").write(json.dumps(r)+"\n"); print(r)
PY
}
run gate0_if3 "--inflight 3" WH_XATT_GATE=0
run gate1_if3 "--inflight 3" WH_XATT_GATE=1
run gate1_if3_lead1792 "--inflight 3" WH_XATT_GATE=1 WH_XATT_GATE_LEAD=1792
run gate1_if4 "--inflight 4" WH_XATT_GATE=1
run gate0_if4 "--inflight 4" WH_XATT_GATE=0
