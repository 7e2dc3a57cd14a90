#!/bin/bash
# narrow filter-gradient tiles: shared-role (2 workgroups/CU) vs producer/consumer waves (RCV_WGRAD_SPEC=1), with ablations
cd $GRAFT_REPO_ROOT
O=gpurun_out/wg3.log
: > $O
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so RCV_DEBUG_PLAN=1
run() { for sp in 0 1; do for f in 0 1048576 2097152; do if [ $sp = 1 ]; then export RCV_WGRAD_SPEC=1; else unset RCV_WGRAD_SPEC; fi; timeout -k 10 120 python scripts/bench_op.py "$@" --flags $f >> $O 2>&1 || exit 1; done; done; }
run wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc
run wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 240 320 16 32 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 480 640 8 16 --stride 2 --mode grad_dec --mode2 affine
run wgrad 32 240 320 16 32 --stride 2 --mode grad_dec --mode2 affine
grep -v 'amdgpu.ids\|wgrad plan' $O | cut -c1-28,90-200
