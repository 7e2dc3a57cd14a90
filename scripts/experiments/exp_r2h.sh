#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h.log
: > $O
export RCV_DMA_MIN_CIN=32
for t in "4,2,40" "4,4,20" "4,1,80"; do
export RCV_CONV_TILE=$t
timeout -k 5 120 python scripts/bench_op.py conv 32 120 160 32 64 --stride 2 --mode affine --stats fwd 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_enc --resid 1 2>/dev/null >> $O
done
cat $O
