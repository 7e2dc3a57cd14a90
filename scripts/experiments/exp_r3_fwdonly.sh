#!/bin/bash
# which BACKWARD launches of the split-bf16 conv kernels pay inside the two-stream step (experiments build librcv_X.so):
# default / conv_bf3 forward only / convn_bf3 forward only / both
cd $GRAFT_REPO_ROOT
export RCV_LIBRARY=$GRAFT_REPO_ROOT/robocupvision_amd/librcv_X.so
run() { timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', 'ms', d['ms_per_step'], 'median', d.get('ms_per_step_median'))
"; }
for r in 1 2; do
  run default
  RCV_BF3_FWD=1 run conv_bf3_fwd_only
  RCV_BF3N_FWD=1 run convn_bf3_fwd_only
  RCV_BF3N_ALL=1 run convn_bf3_all
done
