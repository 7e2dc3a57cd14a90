#!/bin/bash
# round 2, experiment J: LDS pitch of the staged input tile vs bank conflicts (experiments build)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j.log
: > $O
for pad in 1 2 3 4 5; do
export RCV_XPITCH_PAD=$pad
echo "== pad $pad" >> $O
timeout -k 5 120 python scripts/bench_op.py conv 32 30 40 128 128 --mode affine --stats fwd 2>/dev/null >> $O
bash scripts/pmc_op.sh r2j_$pad conv 32 30 40 128 128 --mode affine --stats fwd 2>/dev/null | grep -E "LDS_BANK|LDS_IDX|conv_dma" | head -3 >> $O
done
cat $O
