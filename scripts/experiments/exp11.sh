#!/bin/bash
# narrow wgrad tiles: shared-role (default) vs producer/consumer waves (GPU box)
B="python scripts/bench_op.py"
run() { $B "$@" 2>/dev/null | tail -1 | sed "s/^/base /"; RCV_WGRAD_SPEC_NARROW=1 $B "$@" 2>/dev/null | tail -1 | sed "s/^/spec /"; }
run wgrad 32 480 640 3 8 --mode nchw --mode2 grad_enc
run wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 480 640 8 16 --stride 2 --mode grad_dec --mode2 affine
run wgrad 32 240 320 16 16 --mode affine --mode2 grad_enc
run wgrad 32 240 320 16 32 --stride 2 --mode affine --mode2 grad_enc
run wgrad 32 240 320 16 32 --stride 2 --mode grad_dec --mode2 affine
run wgrad 32 480 640 8 8 --mode affine --mode2 grad_enc
run wgrad 32 240 320 8 16 --mode affine --mode2 grad_enc
