#!/bin/bash
# concurrency picture of three sessions in flight (decode only, large-v3) from a rocprofv3 kernel trace, in the headline's configuration:
#   gpurun -- 'bash tools/overlap_trace.sh TAG [slots=256] [splits=1] [slots per workgroup=2]'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
T=${1:-r06l}; B=${2:-256}; S=${3:-1}; W=${4:-2}
cd /tmp
WH_XABS_SPLITS=$S WH_XABS_SPW=$W timeout 500 rocprofv3 --kernel-trace -d /tmp/ov_$T -o ov -- python $GRAFT_REPO_ROOT/tools/time_decode.py large-v3 $B 3 > $R/${T}_overlap.log 2>&1
DB=$(ls /tmp/ov_$T/*.db /tmp/ov_$T/*/*.db 2>/dev/null | head -1)
{ grep '^{' $R/${T}_overlap.log | cut -c1-400; python $GRAFT_REPO_ROOT/tools/rocpd_overlap.py $DB 0.5 0.9; } > $R/${T}_inflight3_concurrency_${B}_slots_spw$W.txt 2>&1
cat $R/${T}_inflight3_concurrency_${B}_slots_spw$W.txt
