#!/bin/bash
# round 3, GPU call 9: (1) where a beam-search position goes (kernel trace of configs[4]); (2) concurrency picture of three sessions in
# flight with the cross-attention residency capped (room for a projection workgroup on every CU) and with the gate on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
OLD=$GRAFT_REPO_ROOT/tools/abtest/libwhisperhip_b9c5390.so
t() { tag=$1; shift; ( cd /tmp && env "$@" timeout 200 rocprofv3 --kernel-trace -d /tmp/r03k_$tag -o x -- python $GRAFT_REPO_ROOT/tools/time_decode.py tiny.en 1 1 > $R/r03k_$tag.out 2> $R/r03k_$tag.err ); echo "$tag rc=$? segv=$(grep -c SIGSEGV $R/r03k_$tag.err)" | tee -a $R/r03k_rocprofv3_capture_segfault_bisect.txt; rm -rf /tmp/r03k_$tag; }
: > $R/r03k_rocprofv3_capture_segfault_bisect.txt
t notorch_mainthread WH_TOOL_NO_TORCH=1 WH_TOOL_MAIN_THREAD=1
t notorch_workerthread WH_TOOL_NO_TORCH=1
t torch_mainthread WH_TOOL_MAIN_THREAD=1
t torch_workerthread A=1
cd /tmp && timeout 400 rocprofv3 --kernel-trace -d /tmp/r03i_beam -o r03i -- python $GRAFT_REPO_ROOT/tools/time_beam.py > $R/r03i_beam_prof.json 2> $R/r03i_beam_prof.err; echo beam prof rc=$?
cd $GRAFT_REPO_ROOT
DB=$(ls /tmp/r03i_beam/*.db /tmp/r03i_beam/*/*.db 2>/dev/null | head -1)
python tools/rocpd_summary.py $DB > $R/r03i_beam_kernel_stats.csv 2>/dev/null; head -24 $R/r03i_beam_kernel_stats.csv
tail -1 $R/r03i_beam_prof.json
trace() { tag=$1; shift; cd /tmp && env "$@" timeout 400 rocprofv3 --kernel-trace -d /tmp/r03i_$tag -o r03i -- python $GRAFT_REPO_ROOT/tools/time_decode.py large-v3 64 3 > $R/r03i_decode_$tag.json 2> $R/r03i_decode_$tag.err
  cd $GRAFT_REPO_ROOT; DB=$(ls /tmp/r03i_$tag/*.db /tmp/r03i_$tag/*/*.db 2>/dev/null | head -1)
  echo "== $tag: $(grep -v '^#' $R/r03i_decode_$tag.json | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['seq_steps_per_s'], d['ms_per_step_wall'])")" | tee -a $R/r03i_decode_overlap.txt
  python tools/rocpd_overlap.py $DB 0.5 0.8 >> $R/r03i_decode_overlap.txt 2>&1
  python tools/rocpd_summary.py $DB 2>/dev/null | head -9 >> $R/r03i_decode_overlap.txt; rm -rf /tmp/r03i_$tag; }
: > $R/r03i_decode_overlap.txt
trace base WH_XATT_GATE=0
trace gate WH_XATT_GATE=1
trace lds45k WH_XATT_GATE=0 WH_XATT_LDS=45000
trace lds45k_gate WH_XATT_GATE=1 WH_XATT_LDS=45000
cat $R/r03i_decode_overlap.txt
