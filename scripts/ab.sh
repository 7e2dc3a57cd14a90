#!/bin/bash
# usage (GPU box): scripts/ab.sh LIB_A LIB_B ROUNDS -- bench_op.py args...   interleaved A/B timing of two library builds on ONE box
cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=$3; shift 4
for r in $(seq 1 $R); do
  for L in $A $B; do
    RCV_LIBRARY=$GRAFT_REPO_ROOT/$L timeout -k 5 120 python scripts/bench_op.py "$@" 2>/dev/null | sed "s|^|$(basename $L) |"
  done
done
