#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in librcv_A.so librcv_B1.so librcv.so; do
  RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/$L timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd --wino 1 2>/dev/null | sed "s|^|$L |"
done
