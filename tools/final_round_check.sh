#!/bin/bash
# gpurun -- "bash tools/final_round_check.sh TAG": the whole GPU suite, smoke() and the driver's bench line on the current tree
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; T=${1:-final}
( timeout 1300 python -m pytest tests -m gpu -q 2>&1 | tail -30 ) > gpurun_out/${T}_pytest_gpu.log 2>&1; grep -E "passed|failed" gpurun_out/${T}_pytest_gpu.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${T}_smoke.log 2>&1; tail -1 gpurun_out/${T}_smoke.log
timeout 560 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; cut -c1-300 gpurun_out/${T}_bench.json
