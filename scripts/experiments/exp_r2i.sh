#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i.log
: > $O
timeout -k 10 500 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_net.py -m gpu -x -q --timeout 300 -p no:cacheprovider > gpurun_out/r2i_tests.log 2>&1
echo "tests exit=$?" >> $O; tail -2 gpurun_out/r2i_tests.log >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 120 160 32 64 --stride 2 --mode affine --stats fwd 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_enc --resid 1 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 60 80 64 128 --stride 2 --mode affine --stats fwd 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 60 80 64 64 --mode affine --stats fwd 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py tconv 32 30 40 128 64 --mode affine --stats fwd 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py tconv 32 60 80 64 32 --mode affine --stats fwd 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py tconv 32 30 40 128 64 --mode grad_enc --stats bwd_enc --resid 1 2>/dev/null >> $O
cat $O
