#!/bin/bash
# 8-channel full-resolution convolutions of the U-Net configuration (convs_mfma<1,5,8>): staging-only / MFMA-only ablations
cd $GRAFT_REPO_ROOT
O=gpurun_out/c8.log
: > $O
run() { for f in 0 1048576 2097152; do timeout -k 10 120 python scripts/bench_op.py "$@" --flags $f >> $O 2>&1 || exit 1; done; }
run conv 32 480 640 8 8 --mode affine --stats fwd
run conv 32 480 640 8 8 --mode grad_enc --stats bwd_enc --resid 0
run conv 32 240 320 16 8 --mode grad_enc --stats bwd_enc --resid 0
run wgrad 32 480 640 8 8
grep -v 'amdgpu.ids' $O | cut -c1-200
