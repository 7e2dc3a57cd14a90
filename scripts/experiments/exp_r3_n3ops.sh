#!/bin/bash
# narrow layers op by op: split-bf16 kernel (layout 3 / 4) against the fp32 narrow kernel, forward and data-gradient forms
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 120 python scripts/bench_op.py "$@" --reps 30 2>&1 | tail -1 | sed -E 's/ N32 / /; s/mode=//; s/stats=//; s/merged=[0-9] tile=- //'; }
for w in 0 3; do
  run conv 32 240 320 16 16 --mode affine --stats fwd --wino $w
  run conv 32 240 320 16 16 --mode grad_enc --stats bwd_enc --wino $w
  run conv 32 480 640 8 16 --stride 2 --mode affine --stats fwd --wino $w
  run conv 32 480 640 8 16 --stride 2 --mode grad_dec --stats bwd_dec --wino $w
  run conv 32 240 320 16 32 --stride 2 --mode affine --stats fwd --wino $w
  run conv 32 120 160 32 32 --mode affine --stats fwd --wino $w
  run conv 32 120 160 32 32 --mode grad_enc --stats bwd_enc --wino $w
done
for m in 1 4; do
  run tconv 32 240 320 16 8 --merged $m --mode plain --stats fwd
  run tconv 32 240 320 16 8 --merged $m --mode grad_enc --stats bwd_enc --resid 1
  run tconv 32 120 160 32 16 --merged $m --mode plain --stats fwd
  run tconv 32 120 160 32 16 --merged $m --mode grad_enc --stats bwd_enc --resid 1
done
