#!/bin/bash
# Round 5, final call B: the bench under the driver's command line (plan with the short batch first, per-step roofline accounting), then the profiles
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; T=r05g
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/${T}_bench_steps20_warmup5.json ) 2> $R/${T}_bench.err; tail -n 2 $R/${T}_bench.err; cut -c1-400 $R/${T}_bench_steps20_warmup5.json
XS=1 SLOTS=128 bash tools/profile_gpu.sh r05g 2>&1 | tail -40
