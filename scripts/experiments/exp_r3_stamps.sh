#!/bin/bash
# round 3: where do the conv_dma waves wait?  (make STAMPS=1 build: per-segment shader-cycle shares of the pipelined loop)
cd $GRAFT_REPO_ROOT
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_S.so
B="python scripts/bench_op.py"
$B conv 32 120 160 32 64 --stride 2 --mode affine --stats fwd --stamps 1
$B conv 32 60 80 64 128 --stride 2 --mode affine --stats fwd --stamps 1
$B tconv 32 30 40 128 64 --mode affine --stats fwd --stamps 1
$B tconv 32 60 80 64 32 --mode affine --stats fwd --stamps 1
$B conv 32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_dec --stamps 1
