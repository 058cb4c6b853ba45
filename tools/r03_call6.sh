#!/bin/bash
# round 3, GPU call 6: where the in-flight regime loses time - host side of the replay loop, and a kernel trace's concurrency picture
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
B="--steps 6 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline"
( WH_DBG_HOST=1 timeout 300 python bench.py $B > $R/r03f_bench_dbg_host.json ) 2> $R/r03f_bench_dbg_host.err
grep "wh host" $R/r03f_bench_dbg_host.err | tail -12
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/r03f_prof -o r03f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline --no-other-configs --no-roofline > $R/r03f_prof_bench.json 2> $R/r03f_prof.err; echo prof rc=$?
cd $GRAFT_REPO_ROOT
DB=$(ls /tmp/r03f_prof/*.db /tmp/r03f_prof/*/*.db 2>/dev/null | head -1)
python - <<PY
import sqlite3
c=sqlite3.connect("$DB")
print([r[1] for r in c.execute("pragma table_info(kernels)")])
PY
python tools/rocpd_overlap.py $DB 0.55 0.95 > $R/r03f_inflight_overlap.txt 2>&1; cat $R/r03f_inflight_overlap.txt
python tools/rocpd_summary.py $DB > $R/r03f_kernel_stats.csv 2>/dev/null; head -14 $R/r03f_kernel_stats.csv
