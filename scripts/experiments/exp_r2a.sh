#!/bin/bash
# round 2, experiment A: workgroup placement of the under-filled 128-channel conv grid (480 workgroups on 256 CUs)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a.log
: > $O
for args in "480 51200 100" "480 57344 100" "480 86016 100" "256 51200 100" "512 51200 100" "768 51200 100" "240 98304 100 512"; do
  timeout -k 5 60 scripts/micro/wg_census $args >> $O 2>&1
done
for lds in 0 57344 86016; do
  for n in 17 32 34 51; do
    RCV_DMA_LDS_MIN=$lds timeout -k 5 120 python scripts/bench_op.py conv $n 30 40 128 128 --mode affine --stats fwd 2>/dev/null | sed "s/^/lds_min=$lds /" >> $O
  done
done
RCV_DMA_LDS_MIN=57344 timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 2>/dev/null | sed "s/^/lds_min=57344 /" >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 2>/dev/null | sed "s/^/lds_min=0 /" >> $O
cat $O
