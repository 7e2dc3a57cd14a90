#!/bin/bash
# ablations of the narrow conv kernel on the headline net's narrow layers (GPU box)
B="python scripts/bench_op.py"
NS=$((1<<20)); NM=$((1<<21)); BOTH=$((NS|NM))
for shape in "32 120 160 32 32" "32 240 320 16 16" "32 480 640 8 8"; do
  for fl in 0 $NS $NM $BOTH; do
    $B conv $shape --mode affine --stats fwd --flags $fl 2>/dev/null | tail -1
  done
  RCV_CONVS_OCC=1 $B conv $shape --mode affine --stats fwd 2>/dev/null | tail -1
done
$B conv 32 120 160 32 32 --mode grad_enc --stats bwd_enc --resid 1 2>/dev/null | tail -1
$B conv 32 120 160 32 32 --mode grad_enc --stats bwd_enc --resid 1 --flags $NM 2>/dev/null | tail -1
$B conv 32 120 160 32 32 --mode grad_enc --stats bwd_enc --resid 1 --flags $NS 2>/dev/null | tail -1
