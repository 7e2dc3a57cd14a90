#!/bin/bash
# Winograd vs direct on stride-1 layers with 32 input channels (U-Net 32->64 at 60x80; 160x120 config analogue)
cd $GRAFT_REPO_ROOT
O=gpurun_out/w32.log
: > $O
for r in 1 2; do
for w in 0 1; do
  timeout -k 10 120 python scripts/bench_op.py conv 32 60 80 32 64 --mode plain --stats fwd --wino $w 2>&1 | grep -v amdgpu >> $O
  timeout -k 10 120 python scripts/bench_op.py conv 32 60 80 32 64 --mode affine --stats fwd --wino $w 2>&1 | grep -v amdgpu >> $O
  timeout -k 10 120 python scripts/bench_op.py conv 32 120 160 32 64 --mode affine --stats fwd --wino $w 2>&1 | grep -v amdgpu >> $O
  timeout -k 10 120 python scripts/bench_op.py conv 32 60 80 48 64 --mode affine --stats fwd --wino $w 2>&1 | grep -v amdgpu >> $O
done; done
cut -c1-30,36-80,118-200 $O
