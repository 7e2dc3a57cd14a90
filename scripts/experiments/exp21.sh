#!/bin/bash
# merged transposed conv: skipping the structurally zero (parity, tap) filter blocks on/off (GPU box)
B="python scripts/bench_op.py"
NOSKIP=$((1<<22))
run() { $B "$@" 2>/dev/null | tail -1 | sed 's/^/skip   /'; $B "$@" --flags $NOSKIP 2>/dev/null | tail -1 | sed 's/^/noskip /'; }
run tconv 32 120 160 32 16 --merged 1 --mode plain --stats fwd
run tconv 32 120 160 32 16 --merged 1 --mode grad_enc --stats bwd_enc
run tconv 32 240 320 16 8 --merged 1 --mode plain --stats fwd
run tconv 32 240 320 16 8 --merged 1 --mode grad_enc --stats bwd_enc
