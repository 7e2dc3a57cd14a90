#!/bin/bash
# narrow convolutions (convs_mfma): staging-only / MFMA-only ablations of the headline step's shapes
cd $GRAFT_REPO_ROOT
O=gpurun_out/cv.log
: > $O
run() { for f in 0 1048576 2097152; do timeout -k 10 120 python scripts/bench_op.py "$@" --flags $f >> $O 2>&1 || exit 1; done; }
run conv 32 240 320 16 16 --mode affine --stats fwd
run conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc --resid 1
run conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd
run conv 32 120 160 32 32 --mode affine --stats fwd
run conv 32 120 160 32 32 --mode grad_enc --stats bwd_enc --resid 1
run conv 32 240 320 16 32 --stride 2 --mode affine --stats fwd
run tconv 32 240 320 16 8 --merged 1 --mode affine --stats fwd
run tconv 32 120 160 32 16 --merged 1 --mode affine --stats fwd
grep -v 'amdgpu.ids' $O | cut -c1-28,34-75,118-200
