#!/bin/bash
# Round 5, GPU call 4: device batches beyond 128 slots (the session cap lifted to 256)
cd ${GRAFT_REPO_ROOT:-.}; export TMPDIR=/tmp; R=gpurun_out; mkdir -p $R; T=r05d
timeout 600 python tools/bench_sweep.py large-v3 64:3:128:-1 64:3:192:-1 64:3:256:-1 64:2:256:-1 64:2:192:-1 > $R/${T}_bench_device_batch_beyond_128.jsonl 2> $R/${T}_bench.err; cat $R/${T}_bench_device_batch_beyond_128.jsonl; tail -3 $R/${T}_bench.err
