#!/bin/bash
# round 3, GPU call 17: profiles of the final binary - rocprofv3 kernel stats of the bench command (three sessions in flight, and one),
# HBM traffic counters (FETCH_SIZE / WRITE_SIZE in separate --pmc passes), SQ counters (MFMA busy, LDS bank conflicts) of the encoder kernels
TAG=r03z
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_prof -o ${TAG} -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-serial-reference --no-cpu-baseline --no-other-configs > $R/${TAG}_prof_bench.json 2> $R/${TAG}_prof.err; echo prof rc=$?
DB=$(ls /tmp/${TAG}_prof/*.db /tmp/${TAG}_prof/*/*.db 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB > $R/${TAG}_kernel_stats.csv 2> $R/${TAG}_summary.err; head -8 $R/${TAG}_kernel_stats.csv
python $GRAFT_REPO_ROOT/tools/rocpd_overlap.py $DB 0.55 0.95 > $R/${TAG}_inflight_overlap.txt 2>&1; head -8 $R/${TAG}_inflight_overlap.txt
rm -rf /tmp/${TAG}_prof
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/${TAG}_prof1 -o ${TAG}1 -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline --no-other-configs --no-roofline > $R/${TAG}_prof_inflight1_bench.json 2> $R/${TAG}_prof_inflight1.err; echo prof1 rc=$?
DB=$(ls /tmp/${TAG}_prof1/*.db /tmp/${TAG}_prof1/*/*.db 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB > $R/${TAG}_kernel_stats_inflight1.csv 2>> $R/${TAG}_summary.err; head -4 $R/${TAG}_kernel_stats_inflight1.csv
rm -rf /tmp/${TAG}_prof1
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d /tmp/${TAG}_pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/pmc_run.py large-v3 64 8 > $R/${TAG}_pmc_$C.log 2>&1; echo pmc $C rc=$?
  DB=$(ls /tmp/${TAG}_pmc_$C/*.db /tmp/${TAG}_pmc_$C/*/*.db 2>/dev/null | head -1)
  python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB > $R/${TAG}_pmc_$C.csv 2>> $R/${TAG}_pmc_$C.log; rm -rf /tmp/${TAG}_pmc_$C
done
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace -d /tmp/${TAG}_sq2 -o sq -- python $GRAFT_REPO_ROOT/tools/pmc_run.py large-v3 8 2 > $R/${TAG}_pmc_sq2.log 2>&1; echo sq2 rc=$?
DB=$(ls /tmp/${TAG}_sq2/*.db /tmp/${TAG}_sq2/*/*.db 2>/dev/null | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $DB > $R/${TAG}_pmc_sq2.csv 2>> $R/${TAG}_pmc_sq2.log
ls -la $R | grep ${TAG}_ | head -30
