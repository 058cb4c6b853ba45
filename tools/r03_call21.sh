#!/bin/bash
# round 3, GPU call 21: the final binary once more on another box - mel / stage tests + smoke, then the bench under the driver's flags
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_beam.py -x -q 2>&1 | tail -3 ) | tee $R/r03zz_pytest_parity_beam.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $R/r03zz_pytest_parity_beam.log
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $R/r03zz_bench_driver_flags.json ) 2> $R/r03zz_bench_driver_flags.err; tail -2 $R/r03zz_bench_driver_flags.err | cut -c1-200; cut -c1-400 $R/r03zz_bench_driver_flags.json
