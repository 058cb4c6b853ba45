export TMPDIR=/tmp
for v in 1 2 1 2; do WH_LN_V4=$v timeout 200 python tools/time_encoder.py large-v3 256 2>/dev/null | grep "^{" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d['kernels_us']; print(json.dumps({'ln_v4': $v, 'B': d['B'], 'encoder_ms_per_chunk': d['encoder_ms_per_chunk'], 'layernorm_us': k.get('layernorm'), 'md5': d['encoder_output_md5']}))" >> gpurun_out/r06an_layernorm_nt_ab.jsonl; done
cat gpurun_out/r06an_layernorm_nt_ab.jsonl
( timeout 1300 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/r06an_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/r06an_pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06an_smoke.log 2>&1; tail -1 gpurun_out/r06an_smoke.log
timeout 560 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06an_bench.json 2> gpurun_out/r06an_bench.err; cut -c1-300 gpurun_out/r06an_bench.json
