#!/bin/bash
# narrow conv kernel: WN variants per layer (GPU box).  RCV_CONVS_TILE=0,0,WN forces WN only.
B="python scripts/bench_op.py"
run() { for wn in 5 3 2; do RCV_CONVS_TILE=0,0,$wn $B "$@" 2>/dev/null | tail -1 | sed "s/^/WN=$wn /"; done; $B "$@" 2>/dev/null | tail -1 | sed "s/^/auto /"; }
run conv 32 120 160 32 32 --mode affine --stats fwd
run conv 32 120 160 32 32 --mode grad_enc --stats bwd_enc --resid 1
run conv 32 240 320 16 32 --stride 2 --mode affine --stats fwd
run conv 32 240 320 16 32 --stride 2 --mode grad_dec --stats bwd_enc
run conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd
run conv 32 480 640 8 16 --stride 2 --mode grad_dec --stats bwd_enc
run tconv 32 120 160 32 16 --merged 1 --mode plain --stats fwd
run tconv 32 120 160 32 16 --merged 1 --mode grad_enc --stats bwd_enc
run conv 32 240 320 16 16 --mode affine --stats fwd
run conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc --resid 1
run tconv 32 240 320 16 8 --merged 1 --mode plain --stats fwd
run tconv 32 240 320 16 8 --merged 1 --mode grad_enc --stats bwd_enc
