#!/bin/bash
# round 3, GPU call 16: log-mel with the span staged once (tests + kernel time), default workload sweep (chunks per step x sessions in flight)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "mel or stage or transcribe or smoke" 2>&1 | tail -3 ) | tee $R/r03q_pytest_mel.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $R/r03q_pytest_mel.log
timeout 300 python tools/time_encoder.py large-v3 64,8 2>/dev/null | grep -v "^#" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels_us']; print(json.dumps({'B': d['B'], 'encoder_ms_per_chunk': d['encoder_ms_per_chunk'], 'mel_power_us': k.get('mel_power'), 'mel_finalize_us': k.get('mel_finalize'), 'enc_attention_us': k.get('enc_attention'), 'md5': d['encoder_output_md5']}))" | tee $R/r03q_mel_power_time.jsonl
B="--steps 6 --warmup 3 --no-serial-reference --no-cpu-baseline --no-other-configs --no-roofline"
: > $R/r03q_default_workload_sweep.jsonl
run() { tag=$1; shift; ( timeout 400 python bench.py $B "$@" > $R/r03q_bench_$tag.json ) 2> $R/r03q_bench_$tag.err; python - <<PY
import json
try:
    d=json.load(open("$R/r03q_bench_$tag.json")); r=dict(variant="$tag", value=d["value"], ms_per_step=d["ms_per_step"], median_ms_per_step=d["median_ms_per_step"])
except Exception as e: r=dict(variant="$tag", error=str(e))
open("$R/r03q_default_workload_sweep.jsonl","a").write(json.dumps(r)+"\n"); print(r)
PY
}
run b64x3 --batch 64 --inflight 3
run b128x2 --batch 128 --inflight 2
run b96x2 --batch 96 --inflight 2
run b128x3 --batch 128 --inflight 3
run b96x3 --batch 96 --inflight 3
