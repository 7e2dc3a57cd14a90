#!/bin/bash
# round 2, experiment G: tile choice of the 128-channel convs at both plane sizes (experiments build: make EXPERIMENTS=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g.log
: > $O
run() { RCV_CONV_TILE=$1 timeout -k 5 120 python scripts/bench_op.py "${@:2}" 2>/dev/null >> $O; }
for t in "0,2,40" "4,2,40" "1,4,40"; do run $t conv 32 30 40 128 128 --mode affine --stats fwd; done
for t in "0,2,40" "4,2,40" "1,4,40"; do run $t conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1; done
for t in "0,4,20" "4,4,20" "1,8,20" "4,3,20" "4,5,16"; do run $t conv 64 15 20 128 128 --mode affine --stats fwd; done
for t in "0,4,20" "4,4,20" "1,8,20"; do run $t conv 64 15 20 128 128 --mode grad_enc --stats bwd_enc --resid 1; done
for t in "1,2,80" "4,1,80" "1,4,40" "4,2,40"; do run $t conv 32 60 80 64 64 --mode affine --stats fwd; done
cat $O
