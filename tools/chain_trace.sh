#!/bin/bash
# Launch-chain trace of the latency-bound configurations (VERDICT r05 "next round" 6): rocprofv3 kernel trace of the decode loop of tiny.en at 1 slot
# (BASELINE configs[1]) and small at 8 slots (configs[2]); per kernel: calls, duration, and the idle gap in front of it (tools/rocpd_gaps.py).
#   gpurun -- 'bash tools/chain_trace.sh r06'
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
cd /tmp
for CFG in "tiny.en 1" "small 8"; do
  set -- $CFG; M=$1; B=$2
  timeout 400 rocprofv3 --kernel-trace -d /tmp/chain_$M -o chain -- python $GRAFT_REPO_ROOT/tools/time_decode.py $M $B 1 > $R/${TAG:-r06}_chain_${M}_b$B.log 2>&1
  DB=$(ls /tmp/chain_$M/*.db /tmp/chain_$M/*/*.db 2>/dev/null | head -1)
  { grep '^{' $R/${TAG:-r06}_chain_${M}_b$B.log | cut -c1-300; python $GRAFT_REPO_ROOT/tools/rocpd_gaps.py $DB; } > $R/${TAG:-r06}_launch_chain_${M}_b$B.txt 2>&1
  head -40 $R/${TAG:-r06}_launch_chain_${M}_b$B.txt
done
