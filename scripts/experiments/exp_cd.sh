#!/bin/bash
# wide stride-2 / transposed layers on conv_dma: staging-only / MFMA-only ablations
cd $GRAFT_REPO_ROOT
O=gpurun_out/cd.log
: > $O
run() { for f in 0 1048576 2097152; do timeout -k 10 120 python scripts/bench_op.py "$@" --flags $f >> $O 2>&1 || exit 1; done; }
run conv 32 120 160 32 64 --stride 2 --mode affine --stats fwd
run conv 32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_dec --resid 1
run conv 32 60 80 64 128 --stride 2 --mode affine --stats fwd
run tconv 32 60 80 64 32 --mode affine --stats fwd
run tconv 32 60 80 64 32 --mode grad_enc --stats bwd_enc --resid 1
run tconv 32 30 40 128 64 --mode affine --stats fwd
grep -v amdgpu $O | cut -c1-28,34-80,110-200
