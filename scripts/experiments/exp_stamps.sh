#!/bin/bash
# narrow conv kernel, diagnostic build (make STAMPS=1): where a wave's cycles go, per phase of the tile loop
cd $GRAFT_REPO_ROOT
O=gpurun_out/stamps.log
: > $O
run() { timeout -k 10 120 python scripts/bench_op.py "$@" --stamps 1 >> $O 2>&1 || exit 1; }
run conv 32 240 320 16 16 --mode affine --stats fwd
run conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc
run conv 32 480 640 8 8 --mode affine --stats fwd
run conv 32 480 640 8 8 --mode grad_enc --stats bwd_enc
run conv 32 120 160 32 32 --mode affine --stats fwd
run conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd
grep -v 'amdgpu.ids' $O | cut -c1-150
