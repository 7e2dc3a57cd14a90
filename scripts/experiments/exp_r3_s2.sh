#!/bin/bash
# conv2_bf3 (stride-2 wide convs, layout 5): accuracy tests, then op timings against conv_dma
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -s -k "test_conv_kernels_vs_fp64" > gpurun_out/s2_tests.log 2>&1
echo "tests exit=$?"; grep -E "passed|failed|Error|error" gpurun_out/s2_tests.log | tail -5; grep "conv2_bf3" gpurun_out/s2_tests.log | tail -10
run() { timeout -k 10 120 python scripts/bench_op.py "$@" --reps 30 2>&1 | tail -1 | sed -E 's/ N32 / /; s/mode=//; s/stats=//; s/merged=[0-9] tile=- //'; }
for w in 0 5; do
  run conv 32 120 160 32 64 --stride 2 --mode affine --stats fwd --wino $w
  run conv 32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_dec --wino $w
  run conv 32 60 80 64 128 --stride 2 --mode affine --stats fwd --wino $w
  run conv 32 60 80 64 128 --stride 2 --mode grad_dec --stats bwd_enc --wino $w
done
