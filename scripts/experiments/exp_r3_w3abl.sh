#!/bin/bash
# wgradn_bf3 ablations at 16 -> 16: full / cached-element loads (same instructions, no HBM traffic) / no MFMA / neither
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 120 python scripts/bench_op.py "$@" --reps 30 2>&1 | tail -1 | sed -E 's/ N32 / /; s/stats=none merged=0 tile=- //'; }
for fl in 0 1048576 2097152 3145728; do
  run wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc --flags $fl
done
