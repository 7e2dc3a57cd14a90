#!/bin/bash
# convn_bf3 ablations on two HBM-bound forms: full / no staging / no MFMA / no epilogue / staging only / nothing
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 120 python scripts/bench_op.py "$@" --reps 30 2>&1 | tail -1 | sed -E 's/ N32 / /; s/mode=//; s/stats=//; s/merged=[0-9] tile=- //'; }
for fl in 0 1048576 2097152 8388608 10485760 11534336; do
  run tconv 32 240 320 16 8 --merged 4 --mode plain --stats fwd --flags $fl
done
for fl in 0 1048576 2097152 8388608 10485760 11534336; do
  run conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd --wino 3 --flags $fl
done
