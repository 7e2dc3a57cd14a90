#!/bin/bash
# conv_bf3: kernel accuracy tests, then op timings: split-bf16 direct conv against Winograd (fp32 MFMA) on the wide layers
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -s -k "test_conv_kernels_vs_fp64" > gpurun_out/c3_tests.log 2>&1
echo "tests exit=$?"; grep -E "passed|failed|Error|error" gpurun_out/c3_tests.log | tail -5; grep "conv_bf3" gpurun_out/c3_tests.log | tail -30
