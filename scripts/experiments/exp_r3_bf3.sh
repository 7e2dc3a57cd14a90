#!/bin/bash
# wgrad_bf3 (fp32 products on the bf16 matrix pipe): accuracy tests of the wide stride-1 filter gradients, then op timings against the fp32 kernel
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -k "filter_gradient_kernels" > gpurun_out/bf3_tests.log 2>&1
echo "tests exit=$?"; tail -15 gpurun_out/bf3_tests.log
for shape in "32 30 40 128 128" "32 60 80 64 64" "32 30 40 64 128" "32 30 40 128 64"; do
  for fl in 0 131072; do
    echo "== wgrad $shape flags=$fl"
    timeout -k 10 120 python scripts/bench_op.py wgrad $shape --mode affine --mode2 grad_enc --flags $fl --reps 50 2>&1 | tail -2
  done
done
