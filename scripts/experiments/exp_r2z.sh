#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_blocks.py -m gpu -x -q -p no:cacheprovider -k "winograd" 2>&1 | tail -1
for w in 0 1; do
timeout -k 5 120 python scripts/bench_op.py conv 64 15 20 128 128 --mode affine --stats fwd --wino $w 2>/dev/null
timeout -k 5 120 python scripts/bench_op.py conv 64 15 20 128 128 --mode grad_enc --stats bwd_enc --resid 1 --wino $w 2>/dev/null
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 64 64 --mode affine --stats fwd --wino $w 2>/dev/null
timeout -k 5 120 python scripts/bench_op.py conv 32 15 20 128 128 --mode affine --stats fwd --wino $w 2>/dev/null
timeout -k 5 120 python scripts/bench_op.py conv 64 30 40 64 64 --mode affine --stats fwd --wino $w 2>/dev/null
done
