#!/bin/bash
# gpurun -- "bash tools/gemm_persist_ab.sh [TAG]": gemm256_kernel one workgroup per tile (WH_GEMM_PERSIST=0, rounds 2 - 5) vs the persistent tile
# loop (=1, round 6): identical encoder-output MD5s over widths / slot counts / epilogue modes, then the encoder's time per chunk and per kernel.
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-r06d}
for p in 0 1; do
  for m in 0 1; do
    WH_GEMM_PERSIST=$p WH_GEMM_EPI_MODE=$m timeout 300 python tools/enc_epi_ab.py > gpurun_out/${T}_md5_persist${p}_mode$m.json 2> gpurun_out/${T}_md5_persist${p}_mode$m.err || echo "md5 persist $p mode $m FAILED rc=$?"
  done
done
python - <<PY
import json
docs = {(p, m): json.load(open(f"gpurun_out/${T}_md5_persist{p}_mode{m}.json")) for p in (0, 1) for m in (0, 1)}
ref = {k: v for k, v in docs[(0, 1)].items() if k != "mode"}
for key, d in docs.items():
    same = {k: v for k, v in d.items() if k != "mode"} == ref
    print("persist, epilogue mode", key, "identical to one-workgroup-per-tile / staged:", same)
PY
for p in 0 1; do
  WH_GEMM_PERSIST=$p timeout 400 python tools/time_encoder.py large-v3 128,64,8 > gpurun_out/${T}_encoder_time_persist$p.jsonl 2> gpurun_out/${T}_encoder_time_persist$p.err || echo "time persist $p FAILED"
  cut -c1-700 gpurun_out/${T}_encoder_time_persist$p.jsonl
done
