#!/bin/bash
# round 3, GPU call 10: rocprofv3 --kernel-trace segfaults inside the first decode with the current build (call 9) - bisect: old build
# vs new, graph replay vs eager, small vs large model
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
OLD=$GRAFT_REPO_ROOT/tools/abtest/libwhisperhip_b9c5390.so
t() { tag=$1; shift; ( cd /tmp && env "$@" timeout 200 rocprofv3 --kernel-trace -d /tmp/r03j_$tag -o x -- python $GRAFT_REPO_ROOT/tools/time_decode.py $M > $R/r03j_$tag.out 2> $R/r03j_$tag.err ); echo "$tag rc=$? $(grep -c SIGSEGV $R/r03j_$tag.err) $(grep -v '^#' $R/r03j_$tag.out | cut -c1-120 | head -2)"; rm -rf /tmp/r03j_$tag; }
M="tiny.en 1 1"
t new_tiny_graph A=1
t new_tiny_eager WH_NO_GRAPH=1
t old_tiny_graph WHISPERHIP_LIB=$OLD
M="large-v3 8 1"
t new_large8_graph A=1
t new_large8_eager WH_NO_GRAPH=1
t old_large8_graph WHISPERHIP_LIB=$OLD
M="large-v3 64 1"
t new_large64_graph A=1
t old_large64_graph WHISPERHIP_LIB=$OLD
