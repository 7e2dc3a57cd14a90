#!/bin/bash
# round 3: what would the narrow filter gradients cost if the gradient operand were ONE plain tensor (dz written by the data-gradient kernel)?
cd $GRAFT_REPO_ROOT
B="python scripts/bench_op.py"
for m2 in grad_enc plain; do
  $B wgrad 32 240 320 16 16 --mode affine --mode2 $m2
  $B wgrad 32 480 640 8 16 --stride 2 --mode affine --mode2 $m2
  $B wgrad 32 240 320 16 32 --stride 2 --mode affine --mode2 $m2
  $B wgrad 32 120 160 32 32 --mode affine --mode2 $m2
  $B wgrad 32 120 160 32 64 --stride 2 --mode affine --mode2 $m2
  $B wgrad 32 30 40 128 128 --mode affine --mode2 $m2
  $B wgrad 32 480 640 8 8 --mode affine --mode2 $m2
done
$B wgrad 32 240 320 16 16 --mode plain --mode2 plain
# decoder side: gathered operand = dt (two tensors), pointwise = layer input
for m in grad_dec plain; do
  $B wgrad 32 480 640 8 16 --stride 2 --mode $m --mode2 affine
  $B wgrad 32 240 320 16 32 --stride 2 --mode $m --mode2 affine
  $B wgrad 32 120 160 32 64 --stride 2 --mode $m --mode2 affine
done
# the 32 -> 64 stride-2 data gradient (Up1 backward) and neighbours
$B conv 32 120 160 32 64 --stride 2 --mode grad_dec --stats bwd_dec
$B conv 32 120 160 32 64 --stride 2 --mode plain --stats bwd_dec
$B conv 32 120 160 32 64 --stride 2 --mode affine --stats fwd
$B conv 32 60 80 64 128 --stride 2 --mode grad_dec --stats bwd_dec
$B conv 32 240 320 16 32 --stride 2 --mode grad_dec --stats bwd_dec
