#!/bin/bash
# convn_bf3: kernel accuracy (conv + transposed conv tests)
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -p no:cacheprovider -s -k "test_conv_kernels_vs_fp64 or test_transposed_conv_kernels" > gpurun_out/n3_tests.log 2>&1
echo "tests exit=$?"; grep -E "passed|failed|Error|error" gpurun_out/n3_tests.log | tail -5; grep "n_bf3" gpurun_out/n3_tests.log | tail -40
