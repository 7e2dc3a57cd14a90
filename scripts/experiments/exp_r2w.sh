#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2w.log
: > $O
timeout -k 10 300 python -m pytest tests/test_gpu_blocks.py -m gpu -x -q -p no:cacheprovider -k "winograd" -s > gpurun_out/r2w_tests.log 2>&1
echo "tests exit=$?" >> $O; grep -E "winograd \(|passed|failed|^E " gpurun_out/r2w_tests.log | head -30 >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd --wino 1 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 --wino 1 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 60 80 64 64 --mode affine --stats fwd --wino 1 2>/dev/null >> $O
cat $O
