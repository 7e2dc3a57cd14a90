#!/bin/bash
# conv_bf3 ablations: full / no staging / no MFMA / no epilogue / neither
cd $GRAFT_REPO_ROOT
for fl in 0 1048576 2097152 8388608 3145728 11534336; do
  timeout -k 10 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd --wino 3 --flags $fl --reps 50 2>&1 | tail -1
done
