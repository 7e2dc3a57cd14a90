#!/bin/bash
# ablations of narrow wgrad (GPU box)
B="python scripts/bench_op.py"
NS=$((1<<20)); NM=$((1<<21)); BOTH=$((NS|NM))
run() { for fl in 0 $NS $NM $BOTH; do $B "$@" --flags $fl 2>/dev/null | tail -1; done; }
run wgrad 32 480 640 3 8 --mode nchw --mode2 grad_enc
run wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc
run wgrad 32 480 640 8 8 --mode affine --mode2 grad_enc
