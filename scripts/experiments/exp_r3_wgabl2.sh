#!/bin/bash
# round 3: ablation of the narrow (shared-role) filter-gradient tiles: full / no staging / no MFMA / no partial-filter stores
cd $GRAFT_REPO_ROOT
for args in "32 240 320 16 16" "32 480 640 8 16 --stride 2" "32 240 320 16 32 --stride 2" "32 120 160 32 32" "32 120 160 32 64 --stride 2"; do
  for f in 0 1048576 2097152 8388608; do python scripts/bench_op.py wgrad $args --mode affine --mode2 grad_enc --flags $f 2>/dev/null | sed -E "s/ (mode|stats|merged|tile)=[^ ]*//g"; done
done
