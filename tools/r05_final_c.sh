#!/bin/bash
# Round 5, final call C (the last binary: 24-bit cross K / V rows, automatic mode from 28 slots): GPU suite, smoke, bench under the driver's command line, profiles
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; T=r05i
( timeout 1300 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $R/${T}_pytest_gpu.log 2>&1; tail -n 3 $R/${T}_pytest_gpu.log
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 ) > $R/${T}_smoke.log 2>&1; tail -n 2 $R/${T}_smoke.log
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/${T}_bench_steps20_warmup5.json ) 2> $R/${T}_bench.err; tail -n 2 $R/${T}_bench.err; cut -c1-300 $R/${T}_bench_steps20_warmup5.json
XS=1 SLOTS=128 bash tools/profile_gpu.sh r05i 2>&1 | tail -25
