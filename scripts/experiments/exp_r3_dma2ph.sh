#!/bin/bash
# round 3: two-phase 16-channel staging in conv_dma (librcv.so) against the previous build (librcv_X.so), interleaved on one box
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for L in librcv_X.so librcv.so; do
  export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/$L
  for args in "32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_dec" "32 120 160 32 64 --stride 2 --mode affine --stats fwd" \
              "32 60 80 64 128 --stride 2 --mode grad_dec --stats bwd_dec" "32 60 80 64 128 --stride 2 --mode affine --stats fwd" \
              "64 30 40 32 64 --stride 2 --mode grad_dec --stats bwd_dec" "64 30 40 32 64 --stride 2 --mode affine --stats fwd"; do
    python scripts/bench_op.py conv $args 2>/dev/null | sed "s|^|$L |"
  done
done; done
