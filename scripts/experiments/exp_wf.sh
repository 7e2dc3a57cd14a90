#!/bin/bash
# first-layer filter gradient: resident workgroups per CU (RCV_WF_OCC, experiments build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/wf.log
: > $O
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so
for occ in 2 3 4 5 2 3 4; do
  echo "== occ $occ" >> $O
  RCV_WF_OCC=$occ timeout -k 10 120 python scripts/bench_op.py wgrad 32 480 640 3 8 --mode nchw --mode2 grad_enc 2>&1 | grep -v amdgpu >> $O
  RCV_WF_OCC=$occ timeout -k 10 120 python scripts/bench_op.py wgrad 64 120 160 3 8 --mode nchw --mode2 grad_enc 2>&1 | grep -v amdgpu >> $O
done
cut -c1-30,100-200 $O
