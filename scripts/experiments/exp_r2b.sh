#!/bin/bash
# round 2, experiment B: cycle stamps of conv_dma_kernel (diagnostic build, make STAMPS=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b.log
: > $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd --stamps 1 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 17 30 40 128 128 --mode affine --stats fwd --stamps 1 2>/dev/null >> $O
RCV_DMA_LDS_MIN=86016 timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd --stamps 1 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 --stamps 1 2>/dev/null >> $O
cat $O
