#!/bin/bash
# Development run on the MI355X box (gpurun): the whole GPU test suite, then bench.py; everything lands in gpurun_out/.
#   gpurun -- 'bash tools/gpu_round.sh TAG [bench args...]'
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
TAG=${1:-dev}; shift
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -60 ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -n 3 gpurun_out/${TAG}_pytest_gpu.log
( timeout 1200 python bench.py "$@" > gpurun_out/${TAG}_bench.json ) 2> gpurun_out/${TAG}_bench.err
tail -n 5 gpurun_out/${TAG}_bench.err; cut -c1-600 gpurun_out/${TAG}_bench.json
