#!/bin/bash
# wgradn_bf3: accuracy tests, then op timings against the fp32 kernels at the narrow shapes of the 640 x 480 step
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -s -k "filter_gradient_kernels" > gpurun_out/w3_tests.log 2>&1
echo "tests exit=$?"; grep -E "passed|failed|Error|error" gpurun_out/w3_tests.log | tail -5; grep "wgradn_bf3" gpurun_out/w3_tests.log | tail -30
run() { timeout -k 10 120 python scripts/bench_op.py "$@" --reps 30 2>&1 | tail -1 | sed -E 's/ N32 / /; s/stats=none merged=0 tile=- //'; }
for fl in 0 131072; do
  run wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc --flags $fl
  run wgrad 32 120 160 32 32 --mode affine --mode2 grad_enc --flags $fl
  run wgrad 32 240 320 16 32 --stride 2 --mode affine --mode2 grad_enc --flags $fl
  run wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 grad_enc --flags $fl
  run wgrad 32 480 640 8 16 --stride 2 --mode grad_dec --mode2 affine --flags $fl
  run wgrad 32 240 320 16 32 --stride 2 --mode grad_dec --mode2 affine --flags $fl
done
