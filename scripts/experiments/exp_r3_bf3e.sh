#!/bin/bash
# conv_bf3 op timings against Winograd: forward (affine load, forward statistics) and data gradient (two-tensor load, backward statistics)
cd $GRAFT_REPO_ROOT
for shape in "32 30 40 128 128" "32 60 80 64 64"; do
  for w in 2 3; do
    timeout -k 10 120 python scripts/bench_op.py conv $shape --mode affine --stats fwd --wino $w --reps 50 2>&1 | tail -1
    timeout -k 10 120 python scripts/bench_op.py conv $shape --mode grad_enc --stats bwd_enc --resid 0 --wino $w --reps 50 2>&1 | tail -1
  done
done
