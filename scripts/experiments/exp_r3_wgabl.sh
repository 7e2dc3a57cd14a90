#!/bin/bash
# round 3: ablation of the wave-specialised filter gradient (128 -> 128 at 32x30x40): no staging / no MFMA / no partial-filter stores
cd $GRAFT_REPO_ROOT
B="python scripts/bench_op.py wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc"
for f in 0 1048576 2097152 8388608 9437184 10485760; do $B --flags $f 2>/dev/null; done
B="python scripts/bench_op.py wgrad 32 60 80 64 128 --stride 2 --mode affine --mode2 grad_enc"
for f in 0 1048576 2097152 8388608 9437184; do $B --flags $f 2>/dev/null; done
