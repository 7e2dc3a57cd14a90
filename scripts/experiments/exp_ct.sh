#!/bin/bash
# filter gradients of the wide layers on SMALL planes (160x120 config): 64x64 channel tiles vs 32x32 (RCV_WGRAD_CT, experiments build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/ct.log
: > $O
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so RCV_DEBUG_PLAN=1
for r in 1 2; do
for ct in 64 32; do
  for sh in "64 8 10 128 128" "64 15 20 64 64" "64 15 20 64 128 --stride 2" "32 15 20 128 128" "32 30 40 64 64" "32 30 40 128 128"; do
    RCV_WGRAD_CT=$ct timeout -k 10 120 python scripts/bench_op.py wgrad $sh --mode affine --mode2 grad_enc 2>&1 | grep -v amdgpu | sed "s/^/ct=$ct /" >> $O
  done
done; done
grep -v "wgrad plan" $O | cut -c1-40,98-190
grep "wgrad plan" $O | sort | uniq -c
