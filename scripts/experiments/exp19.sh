#!/bin/bash
B="python scripts/bench_op.py"
run() { $B "$@" 2>/dev/null | tail -1; }
for t in "" "4,2,40" "4,4,20" "1,8,20" "0,4,20"; do
RCV_CONV_TILE=$t run conv 64 15 20 128 128 --mode affine --stats fwd
done
for t in "" "4,2,40" "4,4,20" "0,2,40"; do
RCV_CONV_TILE=$t run conv 64 15 20 128 64 --mode affine --stats fwd
done
for t in "" "4,2,40" "4,4,20"; do
RCV_CONV_TILE=$t run conv 32 30 40 128 128 --mode affine --stats fwd
done
