#!/bin/bash
# conv_bf3: what the filter-fragment loads cost (flag 1 << 22 = fragments loaded once), with and without staging
cd $GRAFT_REPO_ROOT
for fl in 0 4194304 1048576 5242880; do
  timeout -k 10 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd --wino 3 --flags $fl --reps 50 2>&1 | tail -1
done
