#!/bin/bash
# round 3, GPU call 7: concurrency picture of the in-flight regime only (no single-stream leg in the trace)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/r03g_prof -o r03g -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-serial-reference --no-cpu-baseline --no-other-configs --no-roofline > $R/r03g_prof_bench.json 2> $R/r03g_prof.err; echo prof rc=$?
cd $GRAFT_REPO_ROOT
DB=$(ls /tmp/r03g_prof/*.db /tmp/r03g_prof/*/*.db 2>/dev/null | head -1)
for w in "0.0 1.0" "0.45 0.6" "0.6 0.75" "0.75 0.9"; do python tools/rocpd_overlap.py $DB $w; done > $R/r03g_inflight_overlap.txt 2>&1; cat $R/r03g_inflight_overlap.txt
python tools/rocpd_summary.py $DB > $R/r03g_kernel_stats.csv 2>/dev/null; head -8 $R/r03g_kernel_stats.csv
python - <<PY
import sqlite3
c=sqlite3.connect("$DB")
print(c.execute("select count(distinct stream_id), count(distinct queue_id) from kernels").fetchall())
print(c.execute("select stream_id, queue_id, count(*) from kernels group by 1,2").fetchall())
PY
