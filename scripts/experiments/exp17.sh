#!/bin/bash
B="python scripts/bench_op.py"
run() { $B "$@" 2>/dev/null | tail -1; }
for m in 0 1; do
run tconv 32 60 80 64 32 --merged $m --mode plain --stats fwd
run tconv 32 60 80 64 32 --merged $m --mode grad_enc --stats bwd_enc
run tconv 32 30 40 128 64 --merged $m --mode plain --stats fwd
run tconv 32 30 40 128 64 --merged $m --mode grad_enc --stats bwd_enc
done
NS=$((1<<20)); NM=$((1<<21))
for fl in $NS $NM; do
run tconv 32 60 80 64 32 --merged 0 --mode plain --stats fwd --flags $fl
run tconv 32 30 40 128 64 --merged 0 --mode plain --stats fwd --flags $fl
done
