#!/bin/bash
# round 3: Winograd kernel, V image with the tile swap on rows 2,3 (librcv.so) vs the previous build (librcv_X.so): time + LDS counters
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for L in librcv_A.so librcv.so; do
  export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/$L
  for args in "32 30 40 128 128 --mode affine --stats fwd --wino 1" "32 30 40 128 128 --mode grad_enc --stats bwd_enc --resid 1 --wino 1" \
              "32 60 80 64 64 --mode affine --stats fwd --wino 1" "64 15 20 128 128 --mode affine --stats fwd --wino 1"; do
    python scripts/bench_op.py conv $args 2>/dev/null | sed "s|^|$L |"
  done
done; done
for L in librcv_A.so librcv.so; do
  export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/$L
  bash scripts/pmc_op.sh wino_$L conv 32 30 40 128 128 --mode affine --stats fwd --wino 1 > /dev/null 2>&1
  echo "== $L"; grep -E "conv_wino|BANK_CONFLICT|LDS_IDX_ACTIVE|MFMA_BUSY|GRBM_GUI|WAIT_INST_LDS|SQ_INSTS_LDS" gpurun_out/pmc_wino_$L.txt
done
