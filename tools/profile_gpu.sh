#!/bin/bash
# Profiles of the headline workload on the MI355X box: rocprofv3 kernel-trace stats of a short bench run, HBM traffic counters
# (FETCH_SIZE / WRITE_SIZE in separate --pmc passes, as MI355X_MICROARCH.md prescribes) and SQ counters for the encoder GEMM.
#   gpurun -- 'bash tools/profile_gpu.sh r02'      outputs: gpurun_out/<tag>_*  (copy what should be kept into profiles/)
#   XS = key splits per slot of the absorbed cross-attention for every run (default 1 = what bench.py uses with 128-slot batches in flight); SLOTS = slots of the
#   counter passes (default 128 = bench.py's device batch); SKIP_SQ=1 skips the SQ passes; the counter passes run 16 decoder steps from position 0
#   (the positions bench.py's algorithmic bytes assume: VERDICT r04 weak 10)
TAG=${1:-prof}
XS=${XS:-1}
SLOTS=${SLOTS:-256}
export WH_PMC_STEPS=16
SPW=${SPW:-2}                                   # slots per xabs_attn workgroup (bench.py: 2 at 256 slots)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
cd /tmp
# (12 steps: three sessions x one full 4-step batch = the 256-slot device batch of the headline; fewer steps would pack smaller batches)
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_prof -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --device-batch $SLOTS --no-cpu-baseline --no-other-configs --cross-attention-splits $XS --cross-attention-slots-per-workgroup $SPW > $R/${TAG}_prof_bench.json 2> $R/${TAG}_prof.err; echo prof rc=$?
DB=$(ls /tmp/${TAG}_prof/*.db /tmp/${TAG}_prof/*/*.db 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB > $R/${TAG}_kernel_stats.csv 2> $R/${TAG}_summary.err; head -12 $R/${TAG}_kernel_stats.csv
# the same command with ONE step in flight: the kernel's own duration (what bench.py's roofline leg times with HIP events on an otherwise
# idle GPU); with three sessions in flight the streams share the HBM and the per-kernel averages above are stretched
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_prof1 -o ${TAG}1 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 4 --inflight 1 --device-batch $SLOTS --cross-attention-splits $XS --cross-attention-slots-per-workgroup $SPW --no-cpu-baseline --no-other-configs --no-roofline > $R/${TAG}_prof_inflight1_bench.json 2> $R/${TAG}_prof_inflight1.err; echo prof1 rc=$?
DB=$(ls /tmp/${TAG}_prof1/*.db /tmp/${TAG}_prof1/*/*.db 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB > $R/${TAG}_kernel_stats_inflight1.csv 2>> $R/${TAG}_summary.err; head -4 $R/${TAG}_kernel_stats_inflight1.csv
[ -n "$SKIP_PMC" ] && exit 0
export WH_XABS_SPLITS=$XS WH_XABS_SPW=$SPW      # the counter passes drive the library through tools/pmc_run.py
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/${TAG}_pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py large-v3 $SLOTS 16 > $R/${TAG}_pmc_$C.log 2>&1; echo pmc $C rc=$?
  DB=$(ls /tmp/${TAG}_pmc_$C/*.db /tmp/${TAG}_pmc_$C/*/*.db 2>/dev/null | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB > $R/${TAG}_pmc_$C.csv 2>> $R/${TAG}_pmc_$C.log
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic.py $R/${TAG}_pmc_FETCH_SIZE.csv $R/${TAG}_pmc_WRITE_SIZE.csv large-v3 $SLOTS > $R/${TAG}_pmc_traffic.json 2>> $R/${TAG}_summary.err
i=0
[ -n "$SKIP_SQ" ] || for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/${TAG}_sq$i -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_run.py large-v3 8 2 > $R/${TAG}_pmc_sq$i.log 2>&1; echo sq$i rc=$?
  DB=$(ls /tmp/${TAG}_sq$i/*.db /tmp/${TAG}_sq$i/*/*.db 2>/dev/null | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB > $R/${TAG}_pmc_sq$i.csv 2>> $R/${TAG}_pmc_sq$i.log
done
ls -la $R | grep ${TAG}_ | head -30
