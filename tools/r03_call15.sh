#!/bin/bash
# round 3, GPU call 15: beam search without cache copies (row -> owner table) + fused filter / top-k, self-attention fetch bound per
# 8 positions, cross-attention at 8 waves per SIMD (A/B); the whole GPU suite on this state
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 ) > $R/r03p_pytest_gpu.log 2>&1; tail -4 $R/r03p_pytest_gpu.log
timeout 400 python tools/time_beam.py 2>/dev/null | tail -1 | tee $R/r03p_beam_configs4.jsonl
: > $R/r03p_xatt_w8_ab.jsonl
for w in 0 1; do WH_XATT_W8=$w timeout 300 python tools/fuse_ab.py large-v3 64 2>&1 | tail -1 | tee -a $R/r03p_xatt_w8_ab.jsonl; done
for w in 0 1; do echo "# w8=$w inflight=3" | tee -a $R/r03p_xatt_w8_ab.jsonl; WH_XATT_W8=$w timeout 300 python tools/time_decode.py large-v3 64 3 2>/dev/null | grep -v "^#" | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); k = d.pop('kernels_us', {}); d['xattn_us'] = k.get('dec_cross_attn'); d['self_attn_us'] = k.get('dec_self_attn'); print(json.dumps(d))" | tee -a $R/r03p_xatt_w8_ab.jsonl; done
