#!/bin/bash
B="python scripts/bench_op.py"
NS=$((1<<20)); NM=$((1<<21)); BOTH=$((NS|NM))
run() { for fl in 0 $NS $NM $BOTH; do $B "$@" --flags $fl 2>/dev/null | tail -1; done; }
run wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
RCV_WGRAD_SPEC_TILES=1 run wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
run wgrad 32 120 160 32 32 --mode affine --mode2 grad_enc
