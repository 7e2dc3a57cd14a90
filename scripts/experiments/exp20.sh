#!/bin/bash
B="python scripts/bench_op.py"
run() { $B "$@" 2>/dev/null | tail -1; }
for t in "" "0,2,40" "0,4,20" "0,5,16" "0,3,20" "0,6,10" "0,8,10"; do
RCV_CONV_TILE=$t run conv 32 30 40 128 128 --mode affine --stats fwd
done
for t in "" "0,4,20" "0,5,16"; do
RCV_CONV_TILE=$t run conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc
done
run conv 64 15 20 128 64 --mode affine --stats fwd
run conv 64 15 20 64 64 --mode affine --stats fwd
