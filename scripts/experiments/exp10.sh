#!/bin/bash
B="python scripts/bench_op.py"
run() { for wn in 5 3; do RCV_CONVS_TILE=0,0,$wn $B "$@" 2>/dev/null | tail -1 | sed "s/^/WN=$wn /"; done; $B "$@" 2>/dev/null | tail -1 | sed "s/^/auto /"; RCV_NO_NARROW4=1 $B "$@" 2>/dev/null | tail -1 | sed "s/^/general /"; }
run tconv 32 120 160 32 16 --merged 1 --mode plain --stats fwd
run tconv 32 120 160 32 16 --merged 1 --mode grad_enc --stats bwd_enc
run tconv 32 120 160 16 16 --merged 1 --mode plain --stats fwd
