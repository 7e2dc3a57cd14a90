#!/bin/bash
B="python scripts/bench_op.py"
run() { $B "$@" 2>/dev/null | tail -1; }
run wgrad 32 30 40 128 128 --mode affine --mode2 grad_enc
run wgrad 32 60 80 64 64 --mode affine --mode2 grad_enc
run wgrad 32 120 160 32 32 --mode affine --mode2 grad_enc
run wgrad 32 120 160 32 64 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 480 640 3 8 --mode nchw --mode2 grad_enc
run wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 480 640 8 16 --stride 2 --mode grad_dec --mode2 affine
run wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc
run wgrad 32 240 320 16 32 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 480 640 8 8 --mode affine --mode2 grad_enc
