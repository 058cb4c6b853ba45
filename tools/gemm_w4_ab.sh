#!/bin/bash
# gpurun -- "bash tools/gemm_w4_ab.sh [TAG]": gemm256_kernel (8 waves, 128 x 64 wave tiles, two waves per SIMD in ping-pong) vs gemm256w_kernel (WH_GEMM_W4=1: 4 waves,
# 128 x 128 wave tiles, one wave per SIMD): identical encoder-output MD5s over widths / slot counts / epilogue modes, then the encoder's time per chunk and per kernel.
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06ae}
for w in 0 1; do
  for m in 0 1; do
    WH_GEMM_W4=$w WH_GEMM_EPI_MODE=$m timeout 300 python tools/enc_epi_ab.py > gpurun_out/${T}_md5_w4${w}_mode$m.json 2> gpurun_out/${T}_md5_w4${w}_mode$m.err || echo "md5 w4 $w mode $m FAILED rc=$?"
  done
done
python - <<PY
import json
docs = {(w, m): json.load(open(f"gpurun_out/${T}_md5_w4{w}_mode{m}.json")) for w in (0, 1) for m in (0, 1)}
ref = {k: v for k, v in docs[(0, 1)].items() if k != "mode"}
for key, d in docs.items():
    same = {k: v for k, v in d.items() if k != "mode"} == ref
    print("one wave per SIMD, epilogue mode", key, "identical to the 8-wave kernel / staged:", same)
PY
for w in 0 1; do
  WH_GEMM_W4=$w timeout 400 python tools/time_encoder.py large-v3 256,64,8 > gpurun_out/${T}_encoder_time_w4$w.jsonl 2> gpurun_out/${T}_encoder_time_w4$w.err || echo "time w4 $w FAILED"
  cut -c1-900 gpurun_out/${T}_encoder_time_w4$w.jsonl
done
