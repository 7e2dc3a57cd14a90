#!/bin/bash
# round 3: wave-specialised filter-gradient tile with compile-time gathered-tile geometry (librcv.so) vs the previous build (librcv_X.so)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for L in librcv_X.so librcv.so; do
  export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/$L
  for args in "32 30 40 128 128 --mode affine --mode2 grad_enc" "32 60 80 64 128 --stride 2 --mode affine --mode2 grad_enc" \
              "32 60 80 64 64 --mode affine --mode2 grad_enc" "32 60 80 64 128 --stride 2 --mode grad_dec --mode2 affine"; do
    python scripts/bench_op.py wgrad $args 2>/dev/null | sed "s|^|$L |"
  done
done; done
