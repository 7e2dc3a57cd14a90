#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2x.log
: > $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd --wino 1 --stamps 1 2>/dev/null >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 --wino 1 --stamps 1 2>/dev/null >> $O
cat $O
