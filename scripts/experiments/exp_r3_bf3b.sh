#!/bin/bash
# wgrad_bf3 ablations: full / no staging / no MFMA / no partial stores
cd $GRAFT_REPO_ROOT
for shape in "32 30 40 128 128" "32 60 80 64 64"; do
  for fl in 0 1048576 2097152 8388608 3145728; do
    timeout -k 10 120 python scripts/bench_op.py wgrad $shape --mode affine --mode2 grad_enc --flags $fl --reps 50 2>&1 | tail -1
  done
done
