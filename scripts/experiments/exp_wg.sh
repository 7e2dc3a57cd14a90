#!/bin/bash
# wgrad ablations: staging-only / MFMA-only timings and the plans of the filter-gradient shapes of the headline step
cd $GRAFT_REPO_ROOT
O=gpurun_out/wg.log
: > $O
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so RCV_DEBUG_PLAN=1
run() { for f in 0 1048576 2097152; do timeout -k 10 120 python scripts/bench_op.py "$@" --flags $f >> $O 2>&1 || exit 1; done; }
run wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
run wgrad 32 60 80 64 64 --mode affine --mode2 grad_enc
run wgrad 32 60 80 64 128 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 120 160 32 32 --mode affine --mode2 grad_enc
run wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc
run wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 240 320 16 32 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 120 160 32 64 --stride 2 --mode affine --mode2 grad_enc
grep -v amdgpu.ids $O
