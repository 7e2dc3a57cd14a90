#!/bin/bash
# wgrad ablations: epilogue (partial stores) skipped = flag bit 23
cd $GRAFT_REPO_ROOT
O=gpurun_out/wg2.log
: > $O
run() { for f in 0 1048576 8388608 9437184; do timeout -k 10 120 python scripts/bench_op.py "$@" --flags $f >> $O 2>&1 || exit 1; done; }
run wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
run wgrad 32 60 80 64 64 --mode affine --mode2 grad_enc
run wgrad 32 120 160 32 32 --mode affine --mode2 grad_enc
run wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc
grep -v amdgpu.ids $O
