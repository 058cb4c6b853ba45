cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; OUT=gpurun_out/r06r_projection_chunk_threshold.jsonl; : > $OUT
for K in "WH_D32_TC_BT=99" "WH_D32_TC_BT=8" "WH_D32_TC_BT=6" "WH_D32_TC_BT=5" "WH_D32_TC_BT=99" "WH_D32_TC_BT=8" "WH_D32_TC_BT=6"; do
  env $K timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-roofline --no-serial-reference 2>/dev/null | tail -1 | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print(json.dumps({'knob': '$K', 'steps': 20, 'audio_s_per_s': d['value'], 'ms_per_step': d['ms_per_step']}))" >> $OUT
  tail -1 $OUT
done
