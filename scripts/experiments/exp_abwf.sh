#!/bin/bash
# first-layer filter gradient: interleaved A/B of two builds (+ its parity tests with build B)
cd $GRAFT_REPO_ROOT
O=gpurun_out/abwf.log
: > $O
timeout -k 10 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_blocks.py -m gpu -x -q -p no:cacheprovider > gpurun_out/abwf_tests.log 2>&1
echo "tests exit=$?" >> $O; tail -1 gpurun_out/abwf_tests.log >> $O
A=robocupvision_amd/${ALIB:-librcv_A.so}; B=robocupvision_amd/librcv.so
R=${1:-3}
bash scripts/ab.sh $A $B $R -- wgrad 32 480 640 3 8 --mode nchw --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 64 120 160 3 8 --mode nchw --mode2 grad_enc >> $O
bash scripts/ab.sh $A $B $R -- wgrad 32 240 320 3 8 --mode nchw --mode2 grad_enc >> $O
grep -v amdgpu $O | cut -c1-40,100-200
