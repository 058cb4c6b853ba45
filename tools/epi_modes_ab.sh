#!/bin/bash
# gpurun -- "bash tools/epi_modes_ab.sh": the three epilogue modes of gemm256_kernel - identical encoder outputs, then the encoder's time per chunk and per kernel
mkdir -p gpurun_out
for m in 0 1 2; do
  WH_GEMM_EPI_MODE=$m timeout 300 python tools/enc_epi_ab.py > gpurun_out/r05r_epi_md5_mode$m.json 2> gpurun_out/r05r_epi_md5_mode$m.err || echo "md5 mode $m FAILED rc=$?"
  tail -c 400 gpurun_out/r05r_epi_md5_mode$m.json
done
python - <<'PY'
import json
r=[json.load(open(f"gpurun_out/r05r_epi_md5_mode{m}.json")) for m in (0,1,2)]
for x in r: x.pop("mode")
print("MD5_EQUAL", r[0]==r[1]==r[2])
PY
for m in 0 1 2; do
  WH_GEMM_EPI_MODE=$m timeout 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-other-configs --no-serial-reference --dump-records gpurun_out/r05r_records_mode$m.json > gpurun_out/r05r_bench_mode$m.json 2> gpurun_out/r05r_bench_mode$m.err || echo "bench mode $m FAILED"
  python - <<PY
import json
d=json.load(open("gpurun_out/r05r_bench_mode$m.json"))
k=d["roofline"]["kernels"]
print("mode $m", d["value"], d.get("encoder_ms_per_chunk"), {n: round(v["avg_us"],1) for n,v in k.items() if n.startswith("gemm_") or n in ("enc_attention","layernorm")})
PY
done
md5sum gpurun_out/r05r_records_mode*.json
