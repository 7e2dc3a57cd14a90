#!/bin/bash
B="python scripts/bench_op.py"
NS=$((1<<20)); NM=$((1<<21)); BOTH=$((NS|NM))
for fl in 0 $NS $NM $BOTH; do $B conv 32 480 640 3 8 --mode nchw --stats fwd --flags $fl 2>/dev/null | tail -1; done
RCV_CONVS_OCC=1 $B conv 32 480 640 3 8 --mode nchw --stats fwd 2>/dev/null | tail -1
for fl in 0 $NS $NM $BOTH; do $B tconv 32 240 320 16 8 --merged 1 --mode plain --stats fwd --flags $fl 2>/dev/null | tail -1; done
for fl in 0 $NS $NM $BOTH; do $B tconv 32 240 320 16 8 --merged 1 --mode grad_enc --stats bwd_enc --flags $fl 2>/dev/null | tail -1; done
