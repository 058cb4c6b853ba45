#!/bin/bash
# concurrency picture of three sessions in flight (decode only, large-v3, 64 slots) from a rocprofv3 kernel trace, for 4 and 2 key splits per slot
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
cd /tmp
for S in 2 4; do
  WH_XABS_SPLITS=$S timeout 400 rocprofv3 --kernel-trace -d /tmp/ov$S -o ov -- python $GRAFT_REPO_ROOT/tools/time_decode.py large-v3 64 3 > $R/${1:-r04aj}_overlap_splits$S.log 2>&1
  DB=$(ls /tmp/ov$S/*.db /tmp/ov$S/*/*.db 2>/dev/null | head -1)
  { grep '^{' $R/${1:-r04aj}_overlap_splits$S.log | cut -c1-400; python $GRAFT_REPO_ROOT/tools/rocpd_overlap.py $DB 0.5 0.9; } > $R/${1:-r04aj}_inflight3_concurrency_splits$S.txt 2>&1
  cat $R/${1:-r04aj}_inflight3_concurrency_splits$S.txt
done
