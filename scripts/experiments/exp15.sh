#!/bin/bash
B="python scripts/bench_op.py"
run() { for t in 1 2 4 8; do RCV_WGRAD_SPEC_TILES=$t $B "$@" 2>/dev/null | tail -1 | sed "s/^/tiles>=$t /"; done; }
run wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
run wgrad 32 60 80 64 64 --mode affine --mode2 grad_enc
run wgrad 32 120 160 32 32 --mode affine --mode2 grad_enc
run wgrad 32 120 160 32 64 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 60 80 64 128 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 60 80 64 32 --stride 2 --mode grad_dec --mode2 affine
