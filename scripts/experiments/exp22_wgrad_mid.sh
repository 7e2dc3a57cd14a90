#!/bin/bash
# filter-gradient layers of the headline config: plans + timings (GPU box)
for args in "32 120 160 32 32" "32 120 160 32 64 --stride 2" "32 240 320 16 16" "32 240 320 16 32 --stride 2" "32 480 640 8 16 --stride 2" "32 60 80 64 64" "32 60 80 64 128 --stride 2" "32 30 40 128 128" "32 480 640 16 8 --stride 2 --mode grad_dec --mode2 affine" "32 240 320 32 16 --stride 2 --mode grad_dec --mode2 affine"; do
  RCV_DEBUG_PLAN=1 python scripts/bench_op.py wgrad $args 2>&1 | grep -E "wgrad plan|wgrad_" | tail -2
done
