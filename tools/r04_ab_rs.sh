#!/bin/bash
# A/B of the two xabs_attn forms on one box: register-staged (default) vs LDS-DMA ring (WH_XABS_DMA=1): decode ms per step + kernel table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $R
OUT=$R/${1:-r04y}_xabs_rs_vs_dma.jsonl; : > $OUT
for dma in 0 1; do
  WH_XABS=1 WH_XABS_DMA=$dma timeout 400 python tools/time_decode.py large-v3 ${2:-64,32,8} 1 2>&1 | grep '^{' | sed "s/^{/{\"dma\": $dma, /" >> $OUT
  WH_XABS=1 WH_XABS_DMA=$dma timeout 400 python tools/time_decode.py large-v3 64 3 2>&1 | grep '^{' | sed "s/^{/{\"dma\": $dma, /" >> $OUT
done
cat $OUT
