#!/bin/bash
# Round-end validation on the GPU box: parity tests, smoke, the default bench line and a rocprofv3 kernel-trace summary of a
# short run of the same bench.  Outputs under gpurun_out/<tag>_*; copy what should be kept into profiles/.
#   gpurun --timeout 900 -- 'bash tools/validate_gpu.sh r01d [--skip-tests]'
TAG=${1:-run}; SKIP=$2
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
if [ "$SKIP" != "--skip-tests" ]; then
  timeout 400 python -m pytest tests -m gpu -x -q > $R/${TAG}_pytest.log 2>&1; tail -3 $R/${TAG}_pytest.log
  timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $R/${TAG}_smoke.log 2>&1; echo smoke rc=$?; tail -2 $R/${TAG}_smoke.log
fi
timeout 400 python bench.py > $R/${TAG}_bench.json 2> $R/${TAG}_bench.err; echo bench rc=$?; cut -c1-300 $R/${TAG}_bench.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_prof -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs > $R/${TAG}_prof_bench.json 2> $R/${TAG}_prof.err; echo prof rc=$?
cd $GRAFT_REPO_ROOT
DB=$(ls /tmp/${TAG}_prof/*.db /tmp/${TAG}_prof/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py $DB > $R/${TAG}_kernel_stats.csv 2> $R/${TAG}_summary.err; head -8 $R/${TAG}_kernel_stats.csv
